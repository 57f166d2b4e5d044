#!/usr/bin/env python3
"""One case of tools/shape_sweep.py against the fp64 composed path (same bf16-rounded parameters and inputs):
python3 tools/sweep_case.py uniform|normalized tg(0/1) sg(none|node|edge) F G T B train(0/1) [p32]   (p32: fp32 parameters, bf16 activations)"""
import copy
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml

gname, tg, sg, F, G, T, B, train = sys.argv[1], bool(int(sys.argv[2])), (None if sys.argv[3] == 'none' else sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), bool(int(sys.argv[8]))
dev = torch.device('cuda:0')
N, K = 1000, 5
S = bench.sbm_graph(N, normalized=(gname == 'normalized'))
torch.manual_seed(1)
cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, sg, 1, True)
cell.addGSO(torch.tensor(S))
cell = (cell.float() if 'p32' in sys.argv[9:] else cell.to(torch.bfloat16)).to(dev)
X = torch.randn(B, T, G, N, device=dev).to(torch.bfloat16)
h0 = (0.3 * torch.randn(B, F, N, device=dev)).to(torch.bfloat16) if (B % 2) else torch.zeros(B, F, N, dtype=torch.bfloat16, device=dev)
tgt = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
ref = copy.deepcopy(cell).double()


def run(c, X, h0, tgt):
    if not train:
        with torch.no_grad():
            return c(X, h0).double(), {}
    for q in c.parameters():
        q.grad = None
    H = c(X, h0)
    (H.double() * tgt.double()).sum().backward()
    return H.detach().double(), {k: q.grad.detach().double().clone() for k, q in c.named_parameters() if q.grad is not None}


Hr, gr = run(ref, X.double(), h0.double(), tgt.double())
H1, g1 = run(cell, X, h0, tgt)
os.environ.update({'GCRNN_SEQ32': '0', 'GCRNN_NO_INLINE_PACK': '1'})
for k in ('_fused_plan_rank1', '_fused_plan_rank1_adj'):
    cell.graph.__dict__.pop(k, None)
H0, g0 = run(cell, X, h0, tgt)
for name, H in (('default dispatch', H1), ('round-3 kernels', H0)):
    d = (H - Hr).abs()
    print('%-18s H vs fp64: max %.4g mean %.4g' % (name, float(d.max()), float(d.mean())))
d = (H1 - H0).abs()
i = int(d.argmax())
print('between the two: max %.4g at flat %d: default %.5f round-3 %.5f fp64 %.5f' % (float(d.max()), i, float(H1.view(-1)[i]), float(H0.view(-1)[i]), float(Hr.view(-1)[i])))
for k in gr:
    sc = float(gr[k].abs().max())
    print('%-34s |ref| %.4g  default err %.3g  round-3 err %.3g  (of max)' % (k, sc, float((g1[k] - gr[k]).abs().max()) / max(sc, 1e-30), float((g0[k] - gr[k]).abs().max()) / max(sc, 1e-30)))
