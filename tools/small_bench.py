#!/usr/bin/env python3
"""Inference timing of the small-graph configurations (BASELINE configs[0] and [3]) on the persistent one-launch
kernel vs the composed (multi-launch) path:  python3 tools/small_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gated_gcrnns_amd.Modules.architectures as archit
from gated_gcrnns_amd.Utils import dataTools

dev = torch.device('cuda:0')
adj = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'adj59.npy'))
rng = np.random.default_rng(0)
cases = [('cfg1 SBM N=50 K=2 T=8 G=1 F=20 B=100 regression', dataTools.normalised_gso(dataTools.sbm_adjacency(50, 5, 0.8, 0.2, rng)), 2, 8, 'reg'),
         ('cfg4 seismic N=59 K=3 T=200 G=1 F=20 B=100 classification', dataTools.normalised_gso(adj), 3, 200, 'cls')]
for name, S, K, T, kind in cases:
    N = S.shape[0]
    for dt in (torch.float64, torch.float32):
        for tg in (False, True):
            torch.manual_seed(0)
            if kind == 'reg':
                m = archit.GatedGCRNNforRegression(1, 20, K, K, torch.tanh, torch.nn.ReLU, [1], S, True, time_gating=tg, mlpType='multipMlp')
            else:
                m = archit.GatedGCRNNforClassification(1, 20, K, K, torch.tanh, torch.nn.ReLU, [11], S, True, time_gating=tg)
            m = m.to(dev).to(dt)
            x = torch.randn(100, T, 1, N, device=dev, dtype=dt)
            h0 = torch.zeros(100, 20, N, device=dev, dtype=dt)
            res = {}
            for path in ('small', 'composed'):
                cell = m.stateGCRNN
                saved = cell._use_small
                if path == 'composed':
                    cell._use_small = lambda *a: False
                with torch.no_grad():
                    for _ in range(3):
                        y = m(x, h0)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        y = m(x, h0)
                    torch.cuda.synchronize()
                res[path] = (time.perf_counter() - t0) / 10
                cell._use_small = saved
            print('%-62s %s time_gating=%-5s  persistent %.3f ms (%.0f seq/s)   composed %.2f ms (%.0f seq/s)' % (
                name, str(dt).split('.')[-1], tg, 1e3 * res['small'], 100 / res['small'], 1e3 * res['composed'], 100 / res['composed']))
