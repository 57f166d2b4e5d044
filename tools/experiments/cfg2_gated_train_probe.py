#!/usr/bin/env python3
"""Training step of a gated cell at BASELINE configs[1]/[2] sizes (N=1000, K=5, T=32, G=F=64):
    python3 tools/cfg2_gated_train_probe.py {none|time} [B] [dtype bf16|f32] [steps]
bf16 = bf16 activations with fp32 master weights (fused kernels where supported), f32 = composed path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss

variant = sys.argv[1] if len(sys.argv) > 1 else 'time'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dts = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda:0')
N, K, T, G, F = 1000, 5, 32, 64, 64
S = bench.sbm_graph(N)
torch.manual_seed(0)
cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, variant == 'time', None, 1, True)
cell.addGSO(torch.tensor(S))
cell = cell.to(dev).float()
adt = torch.bfloat16 if dts == 'bf16' else torch.float32
X = torch.randn(B, T, G, N, device=dev).to(adt)
h0 = torch.zeros(B, F, N, device=dev, dtype=adt)
target = torch.randn(B, T, F, N, device=dev).to(adt)
opt = torch.optim.Adam(cell.parameters(), lr=1e-3)


def step():
    cell.zero_grad()
    loss = batchTimeL1Loss(cell(X, h0), target)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / steps
print('%s %s B=%d: %.2f ms/step = %.0f seq/s, loss %.5f, peak mem %.1f GB' % (
    variant, dts, B, ms, 1e3 * B / ms, float(l), torch.cuda.max_memory_allocated() / 2**30))
