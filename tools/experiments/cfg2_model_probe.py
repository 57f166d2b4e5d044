#!/usr/bin/env python3
"""GatedGCRNNforRegression (cell + head, reference architectures.py:1405-1645) at BASELINE configs[1] sizes:
    python3 tools/cfg2_model_probe.py {multipMlp|oneMlp} {none|time} [B] [fwd|train]
bf16 activations, fp32 master weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gated_gcrnns_amd.Modules.architectures as archit
from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss

head = sys.argv[1] if len(sys.argv) > 1 else 'multipMlp'
variant = sys.argv[2] if len(sys.argv) > 2 else 'none'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
mode = sys.argv[4] if len(sys.argv) > 4 else 'fwd'
dev = torch.device('cuda:0')
N, K, T, G, F = 1000, 5, 32, 64, 64
S = bench.sbm_graph(N)
torch.manual_seed(0)
m = archit.GatedGCRNNforRegression(G, F, K, K, torch.tanh, torch.nn.ReLU, [1], S[0], True, time_gating=(variant == 'time'),
                                   spatial_gating=None, mlpType=head).to(dev).float()
X = torch.randn(B, T, G, N, device=dev).to(torch.bfloat16)
h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
target = torch.randn(B, T, 1, N, device=dev).to(torch.bfloat16)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)


def step():
    if mode == 'fwd':
        with torch.no_grad():
            return m(X, h0)
    m.zero_grad()
    loss = batchTimeL1Loss(m(X, h0), target)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    out = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    out = step()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / steps
print('%s %s %s B=%d: %.2f ms/step = %.0f seq/s (out %s %s)' % (head, variant, mode, B, ms, 1e3 * B / ms, tuple(out.shape), out.dtype))
