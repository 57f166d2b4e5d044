#!/usr/bin/env python3
"""Reference-driver-sized training step (N=80 SBM, taps 5, T=5, F=20, batch 100, fp64; kStepPredGRNNs.py defaults, BASELINE.md
row R9): eager composed path vs the whole step captured as one hipGraph.  python3 tools/graphed_train_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gated_gcrnns_amd.Modules.architectures as archit
from gated_gcrnns_amd.Modules.train_rnn import train_step, GraphedTrainStep
from gated_gcrnns_amd.Utils import dataTools, miscTools
from gated_gcrnns_amd.optim import FlatAdam

OPT = sys.argv[1] if len(sys.argv) > 1 else 'flat'        # flat: optim.FlatAdam; torch: torch.optim.Adam (round 1)
ONLY = sys.argv[2] if len(sys.argv) > 2 else None          # e.g. GCRNNMLP:eager -> run just that (for launch counting under rocprofv3)


def make_opt(m, mode):
    if OPT == 'flat':
        return FlatAdam(m.parameters(), lr=1e-3)
    return torch.optim.Adam(m.parameters(), lr=1e-3, capturable=(mode == 'hipgraph'))


dev = torch.device('cuda:0')
torch.set_default_dtype(torch.float64)
rng = np.random.default_rng(0)
W = dataTools.sbm_adjacency(80, 5, 0.8, 0.2, rng)
S = dataTools.normalised_gso(W)
data = dataTools.KStepPrediction(W, 5, 400, 10, 10, horizon=10, rng=rng)
xT, yT = data.getSamples('train')
x = xT[:100].view(100, 5, 1, 80).to(dev); y = yT[:100].view(100, 5, 1, 80).to(dev)
for name, tg, sg in (('GCRNNMLP', False, None), ('TimeGCRNNMLP', True, None), ('NodeGCRNNMLP', False, 'node'),
                     ('EdgeGCRNNMLP', False, 'edge')):
    res = {}
    if ONLY and ONLY.split(':')[0] != name:
        continue
    for mode in (('eager', 'hipgraph') if not ONLY else (ONLY.split(':')[1],)):
        torch.manual_seed(0)
        m = archit.GatedGCRNNforRegression(1, 20, 5, 5, torch.tanh, torch.nn.ReLU, [1], S, True, time_gating=tg, spatial_gating=sg,
                                           mlpType='multipMlp').to(dev)
        opt = make_opt(m, mode)
        if mode == 'eager':
            fn = lambda: train_step(m, miscTools.batchTimeL1Loss, opt, x, y, 20)[0]
        else:
            g = GraphedTrainStep(m, miscTools.batchTimeL1Loss, opt, x, y, 20)
            fn = lambda: g(x, y)[0]
        for _ in range(3): l = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): l = fn()
        torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 20, float(l))
    if ONLY:
        print('%s %s %s: %.2f ms/step over 23 steps' % (name, OPT, ONLY, 1e3 * list(res.values())[0][0]))
        sys.exit(0)
    print('%-14s fp64 B=100 (%s Adam): eager %.2f ms/step (%.0f seq/s)   hipGraph %.2f ms/step (%.0f seq/s)   loss %.5f / %.5f' % (
        name, OPT, 1e3 * res['eager'][0], 100 / res['eager'][0], 1e3 * res['hipgraph'][0], 100 / res['hipgraph'][0], res['eager'][1], res['hipgraph'][1]))

# ---- BASELINE configs[3]: seismic graph N=59 (directed), K=3, T=200, G=1, F=20, batch 100, 11-class head (reference R7/R8)
adj = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'adj59.npy'))
S4 = dataTools.normalised_gso(adj)
x4 = torch.randn(100, 200, 1, 59, device=dev)
y4 = torch.randint(0, 11, (100,), device=dev)
ce = torch.nn.CrossEntropyLoss()
for name, tg in (('cfg4 GCRNN cls', False), ('cfg4 TimeGCRNN cls', True)):
    res = {}
    for mode in ('eager', 'hipgraph'):
        torch.manual_seed(0)
        m = archit.GatedGCRNNforClassification(1, 20, 3, 3, torch.tanh, torch.nn.ReLU, [11], S4, True, time_gating=tg).to(dev)
        opt = make_opt(m, mode)
        if mode == 'eager':
            fn = lambda: train_step(m, ce, opt, x4, y4, 20)[0]
        else:
            g = GraphedTrainStep(m, ce, opt, x4, y4, 20)
            fn = lambda: g(x4, y4)[0]
        for _ in range(2): l = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): l = fn()
        torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 5, float(l))
    print('%-18s fp64 B=100 T=200: eager %.1f ms/step (%.0f seq/s)   hipGraph %.1f ms/step (%.0f seq/s)   loss %.5f / %.5f' % (
        name, 1e3 * res['eager'][0], 100 / res['eager'][0], 1e3 * res['hipgraph'][0], 100 / res['hipgraph'][0], res['eager'][1], res['hipgraph'][1]))
