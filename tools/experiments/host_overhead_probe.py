#!/usr/bin/env python3
"""Host-side cost of one training step (cProfile, top cumulative): python3 tools/host_overhead_probe.py {none|time|node|edge} [B] [T]"""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
variant = sys.argv[1] if len(sys.argv) > 1 else 'node'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device('cuda:0')
N, K, F, G = 1000, 5, 64, 1
cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, variant == 'time', variant if variant in ('node', 'edge') else None, 1, True)
cell.addGSO(torch.tensor(bench.sbm_graph(N)))
cell = cell.to(dev)
X = torch.randn(B, T, G, N, device=dev).to(torch.bfloat16)
h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)


def step():
    cell.zero_grad(set_to_none=True)
    cell(X, h0).float().square().mean().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print('wall per step %.2f ms' % (1e3 * (time.perf_counter() - t0) / 5))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
