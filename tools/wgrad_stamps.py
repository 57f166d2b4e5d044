#!/usr/bin/env python3
"""Phase timeline of the bf16 weight-gradient kernel (gcrnn_fused_wgrad.hip) from in-kernel s_memtime stamps: runs two training steps of the
bench's cell on a diagnostic library (GCRNN_STAMP_LIB: built with -DGCRNN_WGRAD_STAMPS, tools/build_variant_lib.sh wgstamps
"-DGCRNN_WGRAD_STAMPS" gcrnn_fused_wgrad) and prints, per phase, the median over workgroups of the stamp differences of each workgroup's
third item (unit: 100 ticks of s_memtime = 100 shader cycles).   GCRNN_STAMP_LIB=... python3 tools/wgrad_stamps.py"""
import ctypes, os, sys
R = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.environ['GCRNN_STAMP_LIB']
os.environ['GCRNN_LIBPATH'] = lib
sys.path.insert(0, R)
import numpy as np, torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
dev = torch.device('cuda:0')
N, K, T, F, B = 1000, 5, 32, 64, 256
torch.manual_seed(0)
cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
cell.addGSO(torch.tensor(bench.sbm_graph(N)))
cell = cell.to(dev)
X = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
for _ in range(2):
    cell.zero_grad(set_to_none=True)
    cell(X, h0).float().sum().backward()
torch.cuda.synchronize()
buf = np.zeros(1024 * 32, dtype=np.uint64)
assert ctypes.CDLL(lib).gcrnn_debug_read_wgrad_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(1024, 32).astype(np.int64)
st = st[st[:, 0] > 0]
print('%d workgroups stamped; third item of each; unit = 100 shader cycles' % len(st))
names = {0: 'item start', 1: 'z requested, dpre chunk in registers, bias sums'}
for k in range(K):
    names[2 + 4 * k] = 'tap %d images written + barrier' % k
    names[3 + 4 * k] = 'tap %d adjoint hop' % k
    names[4 + 4 * k] = 'tap %d GEMM' % k
    names[5 + 4 * k] = 'tap %d end barrier' % k
names[25] = 'the pair\'s second chunk, whole (z stays resident)'
prev, tot = 0, {}
for s in sorted(names):
    d = st[:, s] - st[:, prev]
    print('%-52s +%7.2f   (min %.2f max %.2f)   t = %.2f' % (names[s], np.median(d) / 100.0, d.min() / 100.0, d.max() / 100.0, np.median(st[:, s] - st[:, 0]) / 100.0))
    key = names[s].split(' ', 2)[2] if names[s].startswith('tap') else names[s]
    tot[key] = tot.get(key, 0.0) + np.median(d) / 100.0
    prev = s
print('--- per (item, chunk), by phase kind')
for k, v in tot.items():
    print('%-52s %8.2f' % (k, v))
