#!/usr/bin/env python3
"""Only the fused weight-gradient kernel at BASELINE configs[1] sizes:  python3 tools/wgrad_probe.py [B] [T] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gated_gcrnns_amd import ops
from gated_gcrnns_amd.graph import GraphOperator
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
N, K, F = 1000, 5, 64
g = GraphOperator(bench.sbm_graph(N), device=dev)
npad = g.fused_plan()['npad']
X = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
H = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
dpre = torch.randn(T, B, npad, F, device=dev).to(torch.bfloat16)
dpre[:, :, N:] = 0
for _ in range(2):
    ops.fused_backward_weight(dpre, X, H, h0, g, F, F, K, want_bias=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    ops.fused_backward_weight(dpre, X, H, h0, g, F, F, K, want_bias=True)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
print('wgrad B=%d T=%d: %.3f ms per launch = %.1f us per (item, chunk) on a CU' % (B, T, ms, 1e3 * ms * 256 / (B * T * 4)))
