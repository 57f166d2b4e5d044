#!/usr/bin/env python3
"""Measured bf16 errors behind the tolerances of tests/test_fused.py (forward vs the fp64 oracle on the same bf16-rounded operands):
prints, per test shape, the first-step max, the overall max and the mean -- the gates in the tests sit at <= 2x the worst of these."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from oracle import gcrnn_oracle as orc
import gated_gcrnns_amd.Utils.graphML as gml
from test_fused import random_graph, bf16_round
dev = torch.device('cuda:0')
for tg in (False, True):
    for (N, F, K, B, T, iso) in [(1000, 64, 5, 3, 4, 0), (200, 32, 3, 9, 5, 11), (1024, 64, 2, 8, 2, 0), (37, 32, 5, 2, 3, 0), (1000, 64, 3, 17, 3, 40),
                                 (500, 64, 4, 5, 3, 0), (300, 32, 2, 6, 3, 5), (1000, 64, 5, 2, 32, 0)]:
        G = F
        S = random_graph(N, min(0.5, 10.0 / N), 21, iso)
        rng = np.random.default_rng(7)
        X = bf16_round(rng.standard_normal((B, T, G, N)))
        h0 = bf16_round(0.5 * rng.standard_normal((B, F, N)))
        torch.manual_seed(3)
        cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
        cell.addGSO(torch.tensor(S))
        cell = cell.to(torch.bfloat16)
        params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
        Href = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X, h0, tg, None)
        cell = cell.to(dev)
        with torch.no_grad():
            H = cell(torch.tensor(X, dtype=torch.bfloat16, device=dev), torch.tensor(h0, dtype=torch.bfloat16, device=dev))
        err = np.abs(H.double().cpu().numpy() - Href)
        print('tg=%d N=%4d F=%2d K=%d T=%2d  first %.2e  max %.2e  mean %.2e' % (tg, N, F, K, T, err[:, 0].max(), err.max(), err.mean()), flush=True)
