#!/usr/bin/env python3
"""BASELINE configs[4]: large sparse graph N=100k, edge density 1e-3 (nnz ~ 1e7), K=3, T=16, G=F=32, B=8 -- the
HBM-bound CSR SpMM stress on the streaming path (every hop is one pass over the [N][B*C] matrix). The reference cannot
run this at all (a dense 1e5 x 1e5 GSO is 40 GB in fp32).  python3 tools/cfg5_bench.py [N] [density]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd.graph import operator_from_csr
from gated_gcrnns_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
K, T, G, F, B = 3, 16, 32, 32, 8
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
deg = rng.binomial(N, dens, size=N).astype(np.int64)                 # Erdos-Renyi, generated directly in CSR
rowptr = np.concatenate([[0], np.cumsum(deg)])
col = np.concatenate([np.sort(rng.choice(N, size=d, replace=False)) for d in deg]).astype(np.int32)
val = rng.random(col.size)
val /= np.max(np.add.reduceat(val, rowptr[:-1][deg > 0]))            # 1 / max row sum: cheap spectral bound
graph = operator_from_csr(rowptr, col, val, N, device=dev)
nnz = graph.nnz
torch.manual_seed(0)
cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
cell.N, cell.S, cell.graph = N, None, graph                           # addGSO equivalent for a CSR-native graph
cell = cell.to(dev)
X = torch.randn(B, T, G, N, device=dev)
h0 = torch.zeros(B, F, N, device=dev)
with torch.no_grad():
    for _ in range(2):
        H = cell(X, h0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        H = cell(X, h0)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
s = 4
hop_bytes = nnz * (4 + s) + 2 * s * N * B * (G + F)                   # SURVEY 8d hop-streaming bytes of one hop over x and h
step_bytes = (K - 1) * hop_bytes + s * N * B * (G + 2 * F)
print('cfg5 N=%d nnz=%d K=%d T=%d G=F=%d B=%d fp32: %.1f ms per batch = %.1f seq/s ; %.2f ms/step ; algorithmic %.0f MB/step -> %.0f GB/s'
      % (N, nnz, K, T, F, B, 1e3 * dt, B / dt, 1e3 * dt / T, step_bytes / 1e6, step_bytes * T / dt / 1e9))
# what a gather-based SpMM actually has to move through the cache hierarchy: every non-zero pulls one B*F-wide row
gather = (K - 1) * nnz * B * F * s
print('   gathered rows: %.1f GB/step (the [N][B F] state is %.0f MB: it lives in the 256 MB Infinity Cache, not in L2) -> %.2f TB/s'
      % (gather / 1e9, N * B * F * s / 1e6, gather * T / dt / 1e12))
# spot check against a CPU CSR evaluation of h_1 on a few nodes
x0 = X[:, 0].double().cpu().numpy(); A = cell.weight_A.detach().double().cpu().numpy()[:, 0]
b = cell.bias.detach().double().cpu().numpy().reshape(-1)
frp, fcol, fval = [a.cpu().numpy() for a in (graph.fwd[0].rowptr, graph.fwd[0].col, graph.fwd[0].val(torch.float64))]
def shift(z):       # z: B x C x N -> (z S)[.., n] = sum_m S[m, n] z[.., m] = row n of CSR(S^T)
    out = np.zeros_like(z)
    for n in nodes_needed: out[:, :, n] = z[:, :, fcol[frp[n]:frp[n + 1]]] @ fval[frp[n]:frp[n + 1]]
    return out
probe = rng.choice(N, 5, replace=False)
# two hops back: nodes needed = probe, their in-neighbours, and theirs
lvl1 = np.unique(np.concatenate([fcol[frp[n]:frp[n + 1]] for n in probe] + [probe]))
nodes_needed = np.unique(np.concatenate([fcol[frp[n]:frp[n + 1]] for n in lvl1] + [lvl1]))
z0 = x0; z1 = shift(z0); nodes_needed = lvl1; z2 = shift(z1)
pre = sum(np.einsum('fg,bgn->bfn', A[:, k], z)[:, :, probe] for k, z in enumerate((z0, z1, z2))) + 2 * b[None, :, None]
ref = np.tanh(pre)
err = np.abs(H[:, 0][:, :, torch.tensor(probe, device=dev)].double().cpu().numpy() - ref).max()
print('spot check of h_0 on 5 nodes vs CPU CSR evaluation: max |diff| = %.2e' % err)
assert err < 1e-5

# bf16 rows (half the gather bytes): same path on the bf16 accumulate-SpMM
cellb = cell.to(torch.bfloat16)
Xb, h0b = X.to(torch.bfloat16), h0.to(torch.bfloat16)
with torch.no_grad():
    for _ in range(2):
        Hb = cellb(Xb, h0b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        Hb = cellb(Xb, h0b)
    torch.cuda.synchronize()
dtb = (time.perf_counter() - t0) / reps
print('cfg5 bf16: %.1f ms per batch = %.1f seq/s ; %.2f ms/step ; max |bf16 - fp32| = %.3e'
      % (1e3 * dtb, B / dtb, 1e3 * dtb / T, float((Hb.float() - H).abs().max())))
