#!/usr/bin/env python3
"""Stress: repeated full-size forwards must agree bit for bit (and with the packed path); prints where they do not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
dev = torch.device('cuda:0')
N, K, T, F, B = 1000, 5, 32, 64, 256
S = bench.sbm_graph(N)
torch.manual_seed(0)
cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
cell.addGSO(torch.tensor(S))
cell = cell.to(torch.bfloat16).to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(5)
X = torch.randn(B, T, F, N, device=dev, generator=gen).to(torch.bfloat16)
h0 = (0.3 * torch.randn(B, F, N, device=dev, generator=gen)).to(torch.bfloat16)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
with torch.no_grad():
    os.environ['GCRNN_NO_INLINE_PACK'] = '1'
    Href = cell(X, h0).clone()
    if not (len(sys.argv) > 2 and sys.argv[2] == 'packed'):      # 'packed': stress the packed path against itself
        del os.environ['GCRNN_NO_INLINE_PACK']
    bad = 0
    for r in range(reps):
        junk = torch.randn(int(np.random.randint(1, 64)) * 1024 * 1024, device=dev)      # move the allocator around
        H = cell(X, h0)
        if not torch.equal(H, Href):
            bad += 1
            d = (H != Href)
            idx = d.nonzero()
            b, t, f, n = idx.t().cpu().numpy()
            print('rep', r, 'mismatches', int(d.sum()), 'b', np.unique(b)[:8], 't', np.unique(t)[:8], 'first t', t.min(),
                  'f', np.unique(f)[:16], 'n range', n.min(), n.max(), 'count n', np.unique(n).size)
        Hs = cell(X[100:108].contiguous(), h0[100:108].contiguous())
        if not torch.equal(Hs, Href[100:108]):
            bad += 1
            d = (Hs != Href[100:108]); idx = d.nonzero(); b, t, f, n = idx.t().cpu().numpy()
            print('rep', r, 'B=8 mismatches', int(d.sum()), 'b', np.unique(b), 'first t', t.min(), 'f', np.unique(f)[:16], 'n', n.min(), n.max())
        del junk
print('bad runs', bad, 'of', 2 * reps)

if len(sys.argv) > 3 and sys.argv[3] == 'train':
    # training steps: every gradient must equal the packed path's, bit for bit, in every repetition
    cellt = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
    cellt.addGSO(torch.tensor(S))
    cellt = cellt.to(dev)
    tgt = torch.randn(B, T, F, N, device=dev, generator=gen)

    def step():
        cellt.zero_grad(set_to_none=True)
        H = cellt(X, h0)
        torch.nn.functional.l1_loss(H.float(), tgt).backward()
        return {n: p.grad.clone() for n, p in cellt.named_parameters() if p.grad is not None}

    os.environ['GCRNN_NO_INLINE_PACK'] = '1'
    gref = step()
    del os.environ['GCRNN_NO_INLINE_PACK']
    badt = 0
    for r in range(reps):
        g = step()
        if any(not torch.equal(g[n], gref[n]) for n in gref):
            badt += 1
            print('rep', r, 'gradient mismatch in', [n for n in gref if not torch.equal(g[n], gref[n])])
    print('bad training steps', badt, 'of', reps)
