#!/usr/bin/env python3
"""Run only the fused step kernel chain (no pack/unpack) for profiling: python3 tools/step_kernel_probe.py [B] [T] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import sbm_graph
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
S = sbm_graph(1000)
torch.manual_seed(0)
cell = gml.GGCRNNCell(64, 64, 5, 5, torch.tanh, False, None, 1, True)
cell.addGSO(torch.tensor(S))
cell = cell.to(dev).to(torch.bfloat16)
plan = cell.graph.fused_plan()
print('entries', plan['entries'], 'nnz', cell.graph.nnz, 'lds bytes', 65536 + 20480 + plan['entries'] * 96)
X = torch.randn(B, T, 64, 1000, device=dev).to(torch.bfloat16)
h0 = torch.zeros(B, 64, 1000, device=dev, dtype=torch.bfloat16)
NATIVE = len(sys.argv) > 4 and sys.argv[4] == 'native'      # sequence-major in and out only (GGCRNNCell.forward_native)
if NATIVE:
    xs = ops.to_sequence_major(X, cell.graph)
    with torch.no_grad():
        for _ in range(reps):
            cell.forward_native(xs, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            cell.forward_native(xs, None)
        torch.cuda.synchronize()
    print('per step (native layout): %.1f us' % (1e6 * (time.perf_counter() - t0) / reps / T))
    sys.exit(0)
with torch.no_grad():
    for _ in range(reps):
        hs, _, _Hu = ops.fused_cell_forward(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, return_states=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        hs, _, _Hu = ops.fused_cell_forward(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, return_states=True)
    torch.cuda.synchronize()
    print('per step (incl. pack of x): %.1f us' % (1e6 * (time.perf_counter() - t0) / reps / T))
