#!/usr/bin/env python3
"""Debug aid: which elements of xs[t >= 1] written by the inline pack differ from the pack kernel's (repeated runs at full size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd import ops, _lib
from gated_gcrnns_amd.ops import _p, _stream, check, lib
dev = torch.device('cuda:0')
N, F, G, K, B, T = 1000, 64, 64, 5, 256, 32
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
S = bench.sbm_graph(N)
torch.manual_seed(0)
cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True); cell.addGSO(torch.tensor(S)); cell = cell.to(dev).to(torch.bfloat16)
gen = torch.Generator(device=dev); gen.manual_seed(5)
X = torch.randn(B, T, G, N, device=dev, generator=gen).to(torch.bfloat16).contiguous()
h0 = (0.3 * torch.randn(B, F, N, device=dev, generator=gen)).to(torch.bfloat16)
plan = cell.graph.fused_plan(); npad = plan['npad']; st = _stream()
xs_ref, hs_all = ops.fused_pack_inputs(X, h0, cell.graph)
wpack = ops._fused_pack_weights(cell.weight_A.detach(), cell.weight_B.detach(), st)
b32 = cell.bias.detach().float().contiguous().view(-1)
H = torch.empty((B, T, F, N), dtype=torch.bfloat16, device=dev)
bad = 0
for r in range(reps):
    xs, _ = ops.fused_pack_inputs(X, h0, cell.graph, first_only=True)
    xs[1:] = 7.0
    check(lib.gcrnn_fused_forward_bf16(_p(xs), _p(hs_all[:1]), _p(hs_all[1:]), _p(wpack), _p(b32), None, None, *ops._fused_graph_args(plan),
                                       B, T, N, F, G, K, _p(H), 0, None, plan['uniform_w'], _p(X), None, None, st), 'fwd')
    torch.cuda.synchronize()
    d = (xs.view(torch.int16) != xs_ref.view(torch.int16))
    if bool(d.any()):
        bad += 1
        idx = d.nonzero()
        t, b, n, f = idx.t().cpu().numpy()
        print('rep', r, 'mismatching elements', int(d.sum()), 't', np.unique(t), 'b', np.unique(b), 'nodes', n.min(), '..', n.max(), 'count', np.unique(n).size,
              'features', np.unique(f))
        for k in range(min(4, idx.shape[0])):
            tt, bb, nn, ff = idx[k].tolist()
            print('   ', (tt, bb, nn, ff), 'got', float(xs[tt, bb, nn, ff]), 'want', float(xs_ref[tt, bb, nn, ff]))
print('bad', bad, 'of', reps)
