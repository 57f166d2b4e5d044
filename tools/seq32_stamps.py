#!/usr/bin/env python3
"""Phase timeline of the wide sequence-resident kernel (gcrnn_fused_seq32.h) from in-kernel s_memtime stamps: builds a diagnostic
library with -DGCRNN_SEQ_STAMPS into /tmp, runs forwards at the bench's size and prints, per phase, the median over workgroups of the
stamp differences (unit: 100 shader cycles). Usage on the GPU box: python3 tools/seq32_stamps.py [native]"""
import ctypes, glob, os, subprocess, sys
R = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(R, 'gated_gcrnns_amd', 'csrc')
out = '/tmp/seq32st'
os.makedirs(out, exist_ok=True)
lib = os.environ.get('GCRNN_STAMP_LIB')      # a diagnostic library built beforehand (tools/build_stamp_lib.sh: the two wide units with -DGCRNN_SEQ_STAMPS, linked with the in-tree objects)
if not (lib and os.path.exists(lib)):
    procs = []
    for f in sorted(glob.glob(C + '/*.hip') + glob.glob(C + '/*.cpp')):
        o = os.path.join(out, os.path.basename(f) + '.o')
        procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-DGCRNN_SEQ_STAMPS'] + os.environ.get('GCRNN_STAMP_FLAGS', '').split() + ['-c', f, '-o', o]))
    assert all(p.wait() == 0 for p in procs)
    lib = os.path.join(out, 'lib.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + sorted(glob.glob(out + '/*.o')))
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
cell = cell.to(torch.bfloat16).to(dev)
X = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
native = len(sys.argv) > 1 and sys.argv[1] == 'native'
chain = len(sys.argv) > 1 and sys.argv[1] == 'chain'      # the BPTT data chain (MODE 2): the last wide launch of a training step's backward
if chain:
    cell = cell.float()
    for _ in range(2):
        cell.zero_grad(set_to_none=True)
        cell(X, h0).float().sum().backward()
gates = len(sys.argv) > 1 and sys.argv[1] == 'gates'      # the time gates' PAIR pre-pass (MODE 1: four 32-feature chunks per item), first item of every workgroup
if gates:
    from gated_gcrnns_amd import ops
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, True, None, 1, True)
    cell.addGSO(torch.tensor(bench.sbm_graph(N)))
    cell = cell.to(torch.bfloat16).to(dev)
    xs = ops.to_sequence_major(X, cell.graph)
    h0s = ops.to_sequence_major(h0.view(B, 1, F, N), cell.graph)
    g = cell._fused_gates()
    with torch.no_grad():
        for _ in range(2):
            ops.fused_time_gate_pair(xs, h0s, g['in'], g['forget'], cell.graph, N, hzero=ops.fused_h0_zero_flag(h0))
with torch.no_grad():
    if chain or gates:
        pass
    elif native:
        from gated_gcrnns_amd import ops
        xs = ops.to_sequence_major(X, cell.graph)
        for _ in range(3):
            cell.forward_native(xs, None)
    else:
        for _ in range(3):
            cell(X, h0)
torch.cuda.synchronize()
buf = np.zeros(256 * 96, dtype=np.uint64)
dll = ctypes.CDLL(lib)
pinned = os.environ.get('GCRNN_SEQ32P', '1') != '0' and not (chain or gates)      # the un-gated forward runs the hand-allocated-hop kernel (gcrnn_fused_seq32p.h)
assert (dll.gcrnn_debug_read_seq32p_stamps if pinned else dll.gcrnn_debug_read_seq32_stamps)(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(256, 96).astype(np.int64)
names = {0: 'step start (operand in registers)'}
for c in range(4 if gates else 2):
    b0 = 1 + 24 * c
    names[b0] = 'c%d seed + barrier' % c
    for j in range(1, K):
        names[b0 + 4 * (j - 1) + 1] = ('c%d hop %d zero + dma issue' if pinned else 'c%d hop %d stream') % (c, j)
        names[b0 + 4 * (j - 1) + 2] = ('c%d hop %d stream with taps' if pinned else 'c%d hop %d taps') % (c, j)
        names[b0 + 4 * (j - 1) + 3] = 'c%d hop %d dma wait + barrier' % (c, j)
        names[b0 + 4 * (j - 1) + 4] = 'c%d hop %d put + weights dma + pack drain + barrier' % (c, j)
    names[b0 + 17] = 'c%d next operand requests + tanh + state stores' % c
    names[b0 + 18] = 'c%d transposed tile + barrier' % c
    names[b0 + 19] = 'c%d user-layout row stores' % c
    names[b0 + 20] = 'c%d vmcnt(0) + end barrier' % c
prev = 0
print('stamps of step T-3 of the persistent launch (%s), median over 256 workgroups; unit = 100 shader cycles (s_memtime)' % ('time gates, pair pre-pass: first item of a workgroup, X laid out by the caller' if gates else 'BPTT data chain, inline layout of dH' if chain else 'native layout' if native else 'user layout + inline pack'))
tot = {}
for s in sorted(names):
    d = st[:, s] - st[:, prev]
    print('%-52s +%7.2f   (min %.2f max %.2f)   t = %.2f' % (names[s], np.median(d) / 100.0, d.min() / 100.0, d.max() / 100.0, np.median(st[:, s] - st[:, 0]) / 100.0))
    key = names[s].split(' ', 1)[1] if names[s][0] == 'c' else names[s]
    key = ' '.join(w for w in key.split() if not w.isdigit())
    tot[key] = tot.get(key, 0.0) + np.median(d) / 100.0
    prev = s
if gates:
    print('--- gate pre-pass epilogue, chunk 1: last hop done -> epilogue start -> tile loop done -> partial stored (units)')
    b0 = 1 + 24
    print('   %.2f  %.2f  %.2f' % (np.median(st[:, b0 + 19] - st[:, b0 + 16]) / 100.0, np.median(st[:, b0 + 18] - st[:, b0 + 19]) / 100.0, np.median(st[:, b0 + 17] - st[:, b0 + 18]) / 100.0))
print('--- per wave, c0 hop 2: first / second phase (waves 0-3: stream, taps; waves 4-7: taps, stream) done, units after the start of the hop')
ref = st[:, 5]          # end of hop 1 (wave 0)
for w in range(8):
    print('wave %d: first phase done +%.2f   second phase done +%.2f' % (w, np.median(st[:, 56 + w] - ref) / 100.0, np.median(st[:, 64 + w] - ref) / 100.0))
print('--- per step, by phase kind')
for k, v in tot.items():
    print('%-52s %8.2f' % (k, v))
