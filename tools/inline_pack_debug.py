#!/usr/bin/env python3
"""Debug aid: which elements of xs[t >= 1] written by the inline pack differ from the pack kernel's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd import ops, _lib
from gated_gcrnns_amd.ops import _p, _stream, check, lib
dev = torch.device('cuda:0')
N, F, G, K, B, T = 1000, 64, 64, 5, 5, 4
rng = np.random.default_rng(41)
W = (rng.random((N, N)) < 10.0 / N).astype(np.float64); W = np.triu(W, 1); W = W + W.T
S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True); cell.addGSO(torch.tensor(S)); cell = cell.to(dev).to(torch.bfloat16)
X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16).contiguous()
h0 = torch.zeros((B, F, N), dtype=torch.bfloat16, device=dev)
plan = cell.graph.fused_plan(); npad = plan['npad']; st = _stream()
xs_ref, hs_all = ops.fused_pack_inputs(X, h0, cell.graph)
xs, _ = ops.fused_pack_inputs(X, h0, cell.graph, first_only=True)
xs[1:] = 7.0
wpack = ops._fused_pack_weights(cell.weight_A.detach(), cell.weight_B.detach(), st)
b32 = cell.bias.detach().float().contiguous().view(-1)
H = torch.empty((B, T, F, N), dtype=torch.bfloat16, device=dev)
check(lib.gcrnn_fused_forward_bf16(_p(xs), _p(hs_all[:1]), _p(hs_all[1:]), _p(wpack), _p(b32), None, None, *ops._fused_graph_args(plan),
                                   B, T, N, F, G, K, _p(H), 0, None, plan['uniform_w'], _p(X), st), 'fwd')
torch.cuda.synchronize()
d = (xs.float() != xs_ref.float())
print('mismatching elements', int(d.sum()), 'of', d.numel())
idx = d.nonzero()
if idx.numel():
    t, b, n, f = idx.t().cpu().numpy()
    print('t', np.unique(t), 'b', np.unique(b))
    print('nodes', np.unique(n)[:40], '... count', np.unique(n).size)
    print('features', np.unique(f))
    for k in range(min(10, idx.shape[0])):
        tt, bb, nn, ff = idx[k].tolist()
        got = float(xs[tt, bb, nn, ff]); want = float(xs_ref[tt, bb, nn, ff])
        # where does the got value live in X?
        hit = (X[bb, tt].float() == got).nonzero()[:4].tolist()
        print((tt, bb, nn, ff), 'got', got, 'want', want, 'X positions (f, n) with the got value', hit)
