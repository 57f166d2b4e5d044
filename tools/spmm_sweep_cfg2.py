#!/usr/bin/env python3
"""Streaming SpMM at the cfg2 composed-path shape: sparse SBM N = 1000 (~10 non-zeros per row), node rows of B*F = 256*64 fp32
(64 KiB). python3 tools/spmm_sweep_cfg2.py -> us per hop per variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import sbm_graph
from gated_gcrnns_amd.graph import GraphOperator
from gated_gcrnns_amd import ops
dev = torch.device('cuda:0')
g = GraphOperator(torch.tensor(sbm_graph(1000))).to(dev)
for dt in (torch.float32, torch.float64):
    acc = torch.randn(1, 1000, 256, 64, device=dev, dtype=dt)
    dst = torch.randn(1, 1000, 256, 64, device=dev, dtype=dt)
    gathered = g.nnz * 256 * 64 * acc.element_size()
    for pl in (8, 16, 32, 64):
        for u in (2, 4, 8):
            if u == 2 and pl > 8:
                continue
            for rpw in (1, 4):
                tune = dict(piece_lanes=pl, unroll=u, rows_per_wave=rpw)
                for _ in range(2):
                    ops.spmm_raw(g.fwd[0], acc, out=dst, accumulate=True, tune=tune)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(10):
                    ops.spmm_raw(g.fwd[0], acc, out=dst, accumulate=True, tune=tune)
                e1.record(); torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / 10
                print('%s piece %4d B unroll %d rows/wave %d: %7.1f us/hop  %5.2f TB/s gathered' % (str(dt)[6:], pl * 16, u, rpw, us, gathered / us / 1e6), flush=True)
    print('auto:', ops.spmm_auto_tune(g.fwd[0], 256 * 64, acc.element_size()))
