#!/usr/bin/env python3
"""Only the BPTT data chain of the fused cell (profiling / ablation builds): python3 tools/bptt_chain_probe.py [B] [T] [reps] [inline]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import sbm_graph
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
inline = (sys.argv[4] != '0') if len(sys.argv) > 4 else True
dev = torch.device('cuda:0')
S = sbm_graph(1000)
torch.manual_seed(0)
cell = gml.GGCRNNCell(64, 64, 5, 5, torch.tanh, False, None, 1, True)
cell.addGSO(torch.tensor(S))
cell = cell.to(dev).to(torch.bfloat16)
npad = cell.graph.fused_plan(adjoint=True)['npad']
hs = (torch.rand(T, B, npad, 64, device=dev) * 1.6 - 0.8).to(torch.bfloat16)
hs[:, :, 1000:] = 0
dH_user = (torch.randn(B, T, 64, 1000, device=dev) * 1e-3).to(torch.bfloat16)
dHs = torch.zeros(T, B, npad, 64, device=dev, dtype=torch.bfloat16)
dHs[:, :, :1000] = dH_user.permute(1, 0, 3, 2)
kw = dict(dH_user=dH_user) if inline else {}
for _ in range(2):
    ops.fused_backward_data(dHs, hs, cell.weight_B, cell.graph, want_dh0=True, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.fused_backward_data(dHs, hs, cell.weight_B, cell.graph, want_dh0=True, **kw)
e1.record()
torch.cuda.synchronize()
print('BPTT chain, inline=%s: %.1f us per launch (T launches per chain)' % (inline, 1e3 * e0.elapsed_time(e1) / reps / T))
