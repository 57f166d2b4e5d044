#!/usr/bin/env python3
"""One optimiser step of the time-gated cell at fp32 accuracy (ops.fused_cell_train_x3_gated) at the bench's sizes, for
`rocprofv3 --kernel-trace --stats -- python3 tools/x3_gated_train_profile.py` (per-kernel breakdown of secondary.train_timegated_f32_x3)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd.optim import FlatAdam
from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss

B = int(os.environ.get('B', 256))
steps = int(os.environ.get('STEPS', 2))
CFG = bench.CFG
N, K, T, G, F = CFG['N'], CFG['K'], CFG['T'], CFG['G'], CFG['F']
S = bench.sbm_graph(N)
dev = torch.device('cuda:0')
torch.manual_seed(0)
c = gml.GGCRNNCell(G, F, K, K, torch.tanh, True, None, 1, True)
c.addGSO(torch.tensor(S))
c = c.to(dev).float()
X = torch.randn(B, T, G, N, device=dev)
h0 = torch.zeros(B, F, N, device=dev)
target = torch.randn(B, T, F, N, device=dev)
opt = FlatAdam(c.parameters(), lr=1e-3)
assert c._use_fused_x3_training(X, h0, time_gated=True)


def step():
    opt.zero_grad()
    batchTimeL1Loss(c(X, h0), target).backward()
    opt.step()


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print('time-gated x3 training step: %.2f ms = %.0f sequences/s (B = %d)' % (1e3 * dt, B / dt, B))
