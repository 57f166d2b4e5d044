#!/usr/bin/env python3
"""Sweep of the streaming SpMM's tuning space at BASELINE configs[4] (N = 1e5, nnz = 1e7, B = 8, F = 32): piece width
(column chunk = 16 * piece_lanes bytes, the slab an XCD's L2 has to hold), gather instructions in flight, rows per wave.
python3 tools/spmm_sweep.py [bf16|f32] -> one line per variant: us per hop, gathered TB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gated_gcrnns_amd.graph import erdos_renyi_csr, operator_from_csr
from gated_gcrnns_amd import ops

dtn = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dt = {'bf16': torch.bfloat16, 'f32': torch.float32}[dtn]
N, B, F = 100000, 8, 32
dev = torch.device('cuda:0')
rowptr, col, val = erdos_renyi_csr(N, 1e-3, seed=0)
g = operator_from_csr(rowptr, col, val, N, device=dev)
nnz = g.nnz
acc = torch.randn(1, N, B, F, device=dev).to(dt)
dst = torch.randn(1, N, B, F, device=dev).to(dt)
elt = acc.element_size()
gathered = nnz * B * F * elt


def timeit(tune, reps=10):
    for _ in range(2):
        ops.spmm_raw(g.fwd[0], acc, out=dst, accumulate=True, tune=tune)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.spmm_raw(g.fwd[0], acc, out=dst, accumulate=True, tune=tune)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


print('cfg5 %s: N=%d nnz=%d L=%d (%d B rows), gathered %.2f GB per hop' % (dtn, N, nnz, B * F, B * F * elt, gathered / 1e9), flush=True)
maxpl = min(64, B * F * elt // 16)
for pl in (4, 8, 16, 32, 64):
    if pl > maxpl:
        continue
    for u in (2, 4, 8):
        if u == 2 and pl > 8:
            continue
        for rpw in (2, 4, 8):
            us = timeit(dict(piece_lanes=pl, unroll=u, rows_per_wave=rpw))
            print('piece %4d B (%d chunks) unroll %d rows/wave %d: %8.1f us/hop  %6.2f TB/s gathered' % (
                pl * 16, B * F * elt // (pl * 16), u, rpw, us, gathered / us / 1e6), flush=True)
