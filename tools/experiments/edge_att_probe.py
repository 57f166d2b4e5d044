#!/usr/bin/env python3
"""Run only the fused edge-attention kernel for profiling: python3 tools/edge_att_probe.py [items] [reps] [mode]
mode 0 = relu(att(z)) (the x branch), 1 = the step epilogue tanh(gx + relu(att(z))) with the user-layout store, 2 = the backward kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import sbm_graph
from gated_gcrnns_amd import ops
from gated_gcrnns_amd.graph import as_operator

items = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device('cuda:0')
N, F = 1000, 64
graph = as_operator(torch.tensor(sbm_graph(N))).to(dev)
npad = graph.fused_plan()['npad']
torch.manual_seed(0)
z = torch.randn(items, npad, F, device=dev).to(torch.bfloat16)
gx = torch.randn(items, npad, F, device=dev).to(torch.bfloat16) if mode == 1 else None
a12 = (0.2 * torch.randn(2, F, device=dev)).contiguous()
out = torch.empty_like(z)
H = torch.empty((items, F, N), dtype=torch.bfloat16, device=dev) if mode == 1 else None
if mode == 2:
    r = ops.fused_edge_attention(z, a12, graph, N=N)
    dpre = (1e-3 * torch.randn(items, npad, F, device=dev)).to(torch.bfloat16)
    scratch = torch.empty((items, graph.edge_plan()['nnz']), dtype=torch.float32, device=dev)
    dz = torch.empty_like(z); dap = torch.empty((items, 2, F), dtype=torch.float32, device=dev)
    run = lambda: ops.fused_edge_attention_backward(dpre, r, None, z, a12, graph, N, scratch=scratch, dz=dz, da_part=dap)
else:
    run = lambda: ops.fused_edge_attention(z, a12, graph, gx=gx, out=out, Huser=H, huser_item_stride=F * N, N=N)
run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / reps
print('edge attention mode %d: %d items, %.1f us per launch, %.1f us per item-slot (256 CUs)' % (mode, items, us, us / max(1.0, items / 256.0)))
