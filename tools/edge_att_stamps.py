#!/usr/bin/env python3
"""Phase timeline of the edge-softmax attention kernel (gcrnn_edge_gate.hip) from in-kernel s_memtime stamps: one edge-gated forward of the
bench's cell on a diagnostic library (GCRNN_STAMP_LIB: tools/build_variant_lib.sh eastamps "-DGCRNN_EDGE_STAMPS" gcrnn_edge_gate); prints the
median over the workgroups of the LAST attention launch (the last time step: MODE 1 with the user-layout copy). Unit: 100 shader cycles."""
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
cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, 'edge', 1, True)
cell.addGSO(torch.tensor(bench.sbm_graph(N)))
cell = cell.to(torch.bfloat16).to(dev)
X = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(2):
        cell(X, h0)
torch.cuda.synchronize()
buf = np.zeros(1024 * 16, dtype=np.uint64)
assert ctypes.CDLL(lib).gcrnn_debug_read_edge_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(1024, 16).astype(np.int64)[:B]
names = ['start', 'z -> LDS, scores, barrier', 'phase A: row softmax statistics, barrier', 'phase B: aggregation + epilogue (tanh, sequence-major stores)',
         'barrier (stores drained)', 'read-back + transposed image, barrier', 'user-layout row stores']
for s in range(1, 7):
    d = st[:, s] - st[:, s - 1]
    print('%-64s +%7.2f   (min %.2f max %.2f)   t = %.2f' % (names[s], np.median(d) / 100.0, d.min() / 100.0, d.max() / 100.0, np.median(st[:, s] - st[:, 0]) / 100.0))
