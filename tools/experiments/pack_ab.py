#!/usr/bin/env python3
"""A/B of the two bf16 layout kernels (user [B][T][C][N] -> sequence-major [T][B][NPad][C]): gcrnn_pack_seq_major (64 x 64 tiles) against
gcrnn_pack_seq_major_steps (32 x 64 tiles, 4.2 KB of LDS) at the forward's sizes. HIP events, 20 launches each."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gated_gcrnns_amd import _lib, ops
lib = _lib.lib
dev = torch.device('cuda:0')
npad = int(lib.gcrnn_fused_padded_nodes())
for (B, T, C, N) in ((256, 1, 64, 1000), (256, 2, 64, 1000), (256, 32, 64, 1000), (100, 1, 64, 1000), (256, 1, 32, 1000)):
    src = torch.randn(B, T, C, N, device=dev).to(torch.bfloat16)
    d0 = torch.zeros(T, B, npad, C, device=dev, dtype=torch.bfloat16)
    d1 = torch.zeros_like(d0)
    st = ops._stream()
    def a():
        ops.check(lib.gcrnn_pack_seq_major(_lib.BF16, ops._p(src), ops._p(d0), B, T, C, N, npad, None, st), 'a')
    def b():
        ops.check(lib.gcrnn_pack_seq_major_steps(ops._p(src), ops._p(d1), B, T, C, N, npad, 0, T, 0, st), 'b')
    res = []
    for fn in (a, b):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / 20)
    mb = 2 * src.numel() * 2 / 1e6
    print('B=%d T=%d C=%d: pack_seq_major %.1f us (%.2f TB/s)   pack_seq_major_steps %.1f us (%.2f TB/s)   equal %s' % (
        B, T, C, res[0], mb / res[0] / 1e6 * 1e6 / 1e6 * 1e0, res[1], mb / res[1], bool(torch.equal(d0, d1))))
