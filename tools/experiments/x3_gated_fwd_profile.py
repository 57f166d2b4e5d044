#!/usr/bin/env python3
"""Kernel-level profile target: the node- / edge-gated forward at 1e-5 (ops.fused_node_cell_forward_x3 / fused_edge_cell_forward_x3) at the bench's
size.   rocprofv3 --kernel-trace --stats -d out -- python3 tools/experiments/x3_gated_fwd_profile.py node|edge [reps]"""
import os, sys, time
R = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R)
import torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
sg = sys.argv[1] if len(sys.argv) > 1 else 'node'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
N, K, T, F, B = 1000, 5, 32, 64, 256
torch.manual_seed(0)
c = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, sg, 1, True)
c.addGSO(torch.tensor(bench.sbm_graph(N)))
c = c.to(dev).float()
X = torch.randn(B, T, F, N, device=dev)
h0 = torch.zeros(B, F, N, device=dev)
with torch.no_grad():
    c(X, h0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c(X, h0)
    torch.cuda.synchronize()
print('%s-gated x3 forward: %.1f ms' % (sg, 1e3 * (time.perf_counter() - t0) / reps))
