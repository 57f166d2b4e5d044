#!/usr/bin/env python3
"""The reference drivers' own cell shape at the bench's graph size: G = 1 input feature, F1 = 20 state features, K1 = 5 taps
(kStepPredGRNNs.py:220-222), N = 1000, T = 32, B = 256 -- on the fused kernels through the zero-padded state channels
(GGCRNNCell._state_padded) and the channel-padding pack. Prints one JSON line per (variant, mode).
    python3 tools/driver_shape_bench.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd.optim import FlatAdam
from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, G, F, K, T = 1000, 1, 20, 5, 32
dev = torch.device('cuda:0')
S = bench.sbm_graph(N)
for tg in (False, True):
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev)
    X = torch.randn(B, T, G, N, device=dev).to(torch.bfloat16)
    h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
    target = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
    ci = cell.to(torch.bfloat16)
    with torch.no_grad():
        assert ci._state_padded(X, h0) is not None
        dt = bench._timed(lambda: ci(X, h0), 10, 3)
    print(json.dumps({'shape': 'N=1000 G=1 F=20 K=5 T=32', 'time_gating': tg, 'mode': 'fwd', 'batch': B, 'ms': 1e3 * dt, 'seq_per_s': B / dt}), flush=True)
    ct = cell.float()
    opt = FlatAdam(ct.parameters(), lr=1e-3)

    def step():
        opt.zero_grad()
        batchTimeL1Loss(ct(X, h0), target).backward()
        opt.step()

    dt = bench._timed(step, 6, 2)
    print(json.dumps({'shape': 'N=1000 G=1 F=20 K=5 T=32', 'time_gating': tg, 'mode': 'train', 'batch': B, 'ms': 1e3 * dt, 'seq_per_s': B / dt}), flush=True)
