#!/usr/bin/env python3
"""Phase timeline of the sequence-resident step kernel (gcrnn_fused_seq.h) from in-kernel s_memtime stamps.
Builds a diagnostic library with -DGCRNN_SEQ_STAMPS into /tmp, runs forwards at the bench's size and prints, per phase, the median
over workgroups of the stamp differences (100 MHz ticks -> us). Usage on the GPU box: python3 tools/seq_stamps.py [train]"""
import ctypes, glob, os, subprocess, sys
R = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(R, 'gated_gcrnns_amd', 'csrc')
out = '/tmp/seqst'
os.makedirs(out, exist_ok=True)
if any(k.startswith('GCRNN_HOP16_') for k in os.environ):      # timing experiments (wrong results) / generator switches: regenerate the stream into a private include directory
    inc = os.path.join(out, 'inc')
    os.makedirs(inc, exist_ok=True)
    for f in glob.glob(C + '/*'):
        subprocess.check_call(['cp', f, inc])
    with open(os.path.join(inc, 'gcrnn_hop_asm.inc'), 'w') as fh:
        subprocess.check_call([sys.executable, os.path.join(R, 'tools', 'gen_hop_asm.py')], stdout=fh)
    os.makedirs('/tmp/include', exist_ok=True)      # (the sources include "../../include/gcrnn.h")
    subprocess.check_call(['cp', os.path.join(R, 'include', 'gcrnn.h'), '/tmp/include/'])
    C = inc
procs = []
for f in sorted(glob.glob(C + '/*.hip') + glob.glob(C + '/*.cpp')):
    o = os.path.join(out, os.path.basename(f) + '.o')
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-DGCRNN_SEQ_STAMPS'] + os.environ.get('GCRNN_STAMP_FLAGS', '').split() + ['-I' + os.path.join(R, 'include'), '-c', f, '-o', o]))
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(out, 'lib.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + sorted(glob.glob(out + '/*.o')))
os.environ['GCRNN_LIBPATH'] = lib
sys.path.insert(0, R)
import numpy as np, torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd import _lib
dev = torch.device('cuda:0')
N, K, T, F, B = 1000, 5, 32, 64, 256
torch.manual_seed(0)
cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
_dens = float(os.environ.get('GCRNN_STAMP_DENSITY', '1'))      # scales the graph's edge probabilities (stream time vs trips per hop)
cell.addGSO(torch.tensor(bench.sbm_graph(N, p_in=0.04 * _dens, p_out=0.0025 * _dens)))
print('graph: density x%g, ELL entries %d (trips per hop and wave = entries / 32)' % (_dens, cell.graph.fused_plan()['entries']))
cell = cell.to(torch.bfloat16).to(dev)
X = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
TRAIN = len(sys.argv) > 1 and sys.argv[1] == 'train'      # (GCRNN_STAMP_FLAGS='-DGCRNN_SEQ_NO_PK_PREFETCH' etc. for A/B builds)
if TRAIN:        # the stamps then come from the BPTT data chain (MODE 2), the last sequence-resident launch of a training step
    cell = cell.float()
    target = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
    for _ in range(2):
        cell.zero_grad()
        torch.nn.functional.l1_loss(cell(X, h0).float(), target.float()).backward()
else:
    with torch.no_grad():
        for _ in range(3):
            cell(X, h0)
torch.cuda.synchronize()
buf = np.zeros(256 * 64, dtype=np.uint64)
dll = ctypes.CDLL(lib)
assert dll.gcrnn_debug_read_seq_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(256, 64).astype(np.int64)
names = {0: 'start', 1: 'operand requested'}
for c in range(4):
    b0 = 2 + 14 * c
    names[b0] = 'c%d seeded+barrier' % c
    for j in range(1, K):
        names[b0 + 2 * j - 1] = 'c%d taps(hop %d)' % (c, j)
        names[b0 + 2 * j] = 'c%d hop %d (+put)' % (c, j)
    names[b0 + 9] = 'c%d vmcnt(0)' % c
    names[b0 + 10] = 'c%d tanh + state stores' % c
    names[b0 + 11] = 'c%d transposed tile + barriers' % c
    names[b0 + 12] = 'c%d user-layout row stores' % c
    names[b0 + 13] = 'c%d pack + end barrier' % c
prev = 0
print('stamps of the LAST step (T-1) of the persistent launch, median over 256 workgroups; unit = 100 shader cycles (s_memtime)')
for s in sorted(names):
    d = st[:, s] - st[:, prev]
    print('%-34s +%7.2f   (min %.2f max %.2f)   t = %.2f' % (names[s], np.median(d) / 100.0, d.min() / 100.0, d.max() / 100.0, np.median(st[:, s] - st[:, 0]) / 100.0))
    prev = s

