#!/usr/bin/env python3
"""Probe for source-range blocking of the streaming SpMM at cfg5: the same N = 1e5 destination rows, but every row's neighbours
drawn from the first N/groups nodes only (25 per row for groups = 4) -- the slab an XCD's L2 has to hold shrinks from N x piece to
N/groups x piece. If the gather rate rises well above the fabric's ~9 TB/s, blocking the sources (one launch per source group) pays.
python3 tools/spmm_l2_probe.py [groups]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gated_gcrnns_amd.graph import operator_from_csr
from gated_gcrnns_amd import ops
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, B, F = 100000, 8, 32
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
for gcount in (1, groups):
    Ns = N // gcount
    deg = rng.binomial(N, 1e-3 / gcount, size=N).astype(np.int64)
    rows = np.repeat(np.arange(N, dtype=np.int64), deg)
    cols = rng.integers(0, Ns, size=rows.size, dtype=np.int64)
    key = np.unique(rows * N + cols)
    rows, cols = key // N, (key % N).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=N))]).astype(np.int64)
    val = rng.random(cols.size)
    # operator_from_csr takes CSR(S) and builds CSR(S^T); feed the transpose roles directly: we want fwd = (rowptr, cols)
    g = operator_from_csr(rowptr, cols, val, N, device=dev)
    csr = g.adj[0]                       # adj = the CSR we passed in: destination rows with sources < Ns
    for dt in (torch.bfloat16, torch.float32):
        acc = torch.randn(1, N, B, F, device=dev).to(dt)
        dst = torch.randn(1, N, B, F, device=dev).to(dt)
        gathered = csr.nnz * B * F * acc.element_size()
        for pl, u in ((8, 8), (8, 4), (16, 4)):
            tune = dict(piece_lanes=pl, unroll=u, rows_per_wave=4)
            for _ in range(2):
                ops.spmm_raw(csr, acc, out=dst, accumulate=True, tune=tune)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10):
                ops.spmm_raw(csr, acc, out=dst, accumulate=True, tune=tune)
            e1.record(); torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 10
            print('sources in first 1/%d of the nodes, %s, piece %d B x %d loads: nnz %d, %7.1f us, %6.2f TB/s gathered' % (
                gcount, str(dt)[6:], pl * 16, u, csr.nnz, us, gathered / us / 1e6), flush=True)
