"""Debug of one train_steps_sweep case: trajectories under each switch separately (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import train_steps_sweep as tss
from shape_sweep import random_graph
dev = torch.device('cuda:0')
N, tg, sg, B, F = 80, True, None, 256, 64
S = random_graph(N, seed=7)
def run(env, opt):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return ['%.4f' % x for x in tss.trajectory(S, N, tg, sg, opt, torch.float32, B, F, dev)]
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
for rep in range(2):
    for opt in ('flat', 'graphed'):
        print(rep, opt, 'default      ', run({}, opt), flush=True)
        print(rep, opt, 'NO_PACK_CACHE', run({'GCRNN_NO_PACK_CACHE': '1'}, opt), flush=True)
        print(rep, opt, 'SEQ32=0      ', run({'GCRNN_SEQ32': '0'}, opt), flush=True)
        print(rep, opt, 'NO_INLINE    ', run({'GCRNN_NO_INLINE_PACK': '1'}, opt), flush=True)
        print(rep, opt, 'all three    ', run(dict(tss.SWITCHES), opt), flush=True)
