"""CPU: the C-ABI library loads, exports every symbol include/gcrnn.h declares, and its host logic
(CSR construction, degree order, argument validation) is exact. No GPU compute is called."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, GOLDEN
from gated_gcrnns_amd import _lib
from gated_gcrnns_amd.graph import GraphOperator, csr_from_dense, degree_order
from oracle import gcrnn_oracle as orc


def header_functions():
    txt = open(os.path.join(ROOT, 'include', 'gcrnn.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(gcrnn_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 10
    lib = C.CDLL(_lib._build.LIBPATH)
    for n in names:
        assert hasattr(lib, n), 'libgcrnn_hip.so does not export %s' % n
    assert sorted(_lib.EXPORTS) == names, 'ctypes binding and header disagree'
    assert _lib.lib.gcrnn_version() >= 100
    assert _lib.lib.gcrnn_status_string(0) == b'ok'


@pytest.mark.parametrize('tag', ['dir30', 'adj59', 'sbm50'])
def test_csr_builder_is_index_exact(tag):
    g = np.load(os.path.join(GOLDEN, 'g7_csr.npz'))
    M = g[tag + '/dense']
    for nm, tr in (('S', False), ('ST', True)):
        rp, col, val = csr_from_dense(M, transpose=tr)
        assert np.array_equal(rp, g['%s/%s_rowptr' % (tag, nm)])
        assert np.array_equal(col, g['%s/%s_col' % (tag, nm)])
        assert np.array_equal(val, g['%s/%s_val' % (tag, nm)])
    rp, col, _ = csr_from_dense(M, add_identity=True, tol=1e-9)
    assert np.array_equal(rp, g[tag + '/mask_rowptr']) and np.array_equal(col, g[tag + '/mask_col'])


def test_csr_edge_cases():
    Z = np.zeros((5, 5))
    rp, col, val = csr_from_dense(Z)
    assert rp.tolist() == [0] * 6 and col.size == 0                    # empty graph
    rp, col, val = csr_from_dense(Z, add_identity=True, tol=1e-9)
    assert col.tolist() == [0, 1, 2, 3, 4] and val.tolist() == [1.0] * 5
    M = -np.eye(3)                                                      # S + I cancels: support empty
    rp, col, _ = csr_from_dense(M, add_identity=True, tol=1e-9)
    assert col.size == 0
    D = np.arange(1, 17, dtype=np.float64).reshape(4, 4)               # full matrix, directed
    rp, col, val = csr_from_dense(D, transpose=True)
    assert np.array_equal(val.reshape(4, 4), D.T)
    nnz = C.c_int64()
    assert _lib.lib.gcrnn_csr_count(None, 4, 0, 0, 0.0, C.byref(nnz)) == 3          # null pointer
    assert _lib.lib.gcrnn_csr_count(D.ctypes.data_as(C.c_void_p), 0, 0, 0, 0.0, C.byref(nnz)) == 2   # bad shape


def test_graph_operator_matches_oracle_and_degree_order():
    rng = np.random.default_rng(0)
    S = (rng.random((1, 40, 40)) < 0.1) * rng.standard_normal((1, 40, 40))
    op = GraphOperator(S)
    rp, col, val = orc.csr_from_dense(S[0].T.copy())
    assert np.array_equal(op.fwd[0].rowptr.numpy(), rp) and np.array_equal(op.fwd[0].col.numpy(), col)
    assert np.array_equal(op.fwd[0].val(torch.float64).numpy(), val)
    assert op.fwd[0].val(torch.float32).dtype == torch.float32
    order = degree_order(op.fwd[0].rowptr.numpy())
    deg = np.diff(rp)
    assert sorted(order.tolist()) == list(range(40))
    assert np.all(np.diff(deg[order]) <= 0)
    # stable: equal degrees keep ascending node id
    for d in np.unique(deg):
        ids = order[deg[order] == d]
        assert np.all(np.diff(ids) > 0)


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch (so these calls are safe on a CPU-only box)."""
    lib = _lib.lib
    one = C.c_void_p(16)
    assert lib.gcrnn_spmm(0, 10, None, None, None, one, one, 4, 1, 0, None) == 3
    assert lib.gcrnn_spmm(0, 0, one, one, one, one, C.c_void_p(32), 4, 1, 0, None) == 2
    assert lib.gcrnn_spmm(0, 10, one, one, one, one, one, 4, 1, 0, None) == 4            # in place
    assert lib.gcrnn_spmm(7, 10, one, one, one, one, C.c_void_p(32), 4, 1, 0, None) == 1   # dtype
    assert lib.gcrnn_pack_node_major(0, None, one, 1, 1, 1, 1, None, None) == 3
    assert lib.gcrnn_pack_node_major(0, one, one, 1, 0, 1, 1, None, None) == 2
    assert lib.gcrnn_taps_forward(0, one, None, 0, one, None, 1.0, one, 8, 3, 2, 5, 0, None) == 3   # K>1 needs zrest
    assert lib.gcrnn_taps_forward(0, one, one, 0, one, None, 1.0, one, 0, 3, 2, 5, 0, None) == 2
    # round 5: one step of the node-gated cell at 1e-5 (F in {32, 64}, N % 4 == 0, NPad % 64 == 0, 16-byte aligned H)
    assert lib.gcrnn_x3_node_gate_step(None, one, one, one, None, one, None, 0, 4, 1000, 1024, 64, None) == 3
    assert lib.gcrnn_x3_node_gate_step(one, one, one, one, None, one, None, 0, 4, 1000, 1024, 48, None) == 2
    assert lib.gcrnn_x3_node_gate_step(one, one, one, one, None, one, None, 0, 4, 1002, 1024, 64, None) == 2
    assert lib.gcrnn_x3_node_gate_step(one, one, one, one, None, one, C.c_void_p(8), 0, 4, 1000, 1024, 64, None) == 2
    with pytest.raises(_lib.GcrnnError):
        _lib.check(2, 'x')


def test_weight_gradient_slot_counts():
    """Partial-sum slots of the bf16 weight-gradient launch (host arithmetic only): with the bf16-image plans a workgroup visits an item for TWO
    16-feature chunks, so an item has half the workgroups and twice the slots fill the 256 CUs; without them (or when the parked accumulators do
    not fit in LDS beside a large graph image) one chunk per visit, the fp32-accurate kernel's count."""
    lib = _lib.lib
    assert lib.gcrnn_fused_wgrad_slots(8192, 64) == 64 and lib.gcrnn_fused_wgrad_slots(8192, 32) == 128
    assert lib.gcrnn_fused_wgrad_bf16_slots(8192, 64, 5, 736, 1) == 128          # pairs: 2 workgroups per item
    assert lib.gcrnn_fused_wgrad_bf16_slots(8192, 64, 5, 736, 0) == 64           # fp32 image: 4 workgroups per item
    assert lib.gcrnn_fused_wgrad_bf16_slots(8192, 32, 3, 736, 1) == 256          # F = 32: one workgroup per item
    assert lib.gcrnn_fused_wgrad_bf16_slots(8192, 64, 5, 2400, 1) == 64          # a graph image that leaves no room for the parked accumulators
    assert lib.gcrnn_fused_wgrad_bf16_slots(5, 64, 5, 736, 1) == 8               # fewer items than slots: rounded up to 8


def test_module_rejects_cpu_tensors_loudly():
    import gated_gcrnns_amd.Utils.graphML as gml
    cell = gml.GGCRNNCell(2, 3, 2, 2, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.eye(6).reshape(1, 6, 6))
    with pytest.raises(_lib.GcrnnError, match='no CPU path'):
        cell(torch.zeros(1, 2, 2, 6), torch.zeros(1, 3, 6))
    with pytest.raises(AssertionError):                                  # reference graphML.py:2339
        cell(torch.zeros(2, 2, 2, 6), torch.zeros(1, 3, 6))
    with pytest.raises(AssertionError):                                  # reference graphML.py:2239-2243
        cell.addGSO(torch.eye(6))


def test_gso_cache_is_not_fooled_by_recycled_storage():
    """A freed dense GSO's address may be handed to a new tensor of the same shape: the cache must not return the old graph."""
    from gated_gcrnns_amd.graph import as_operator
    import gc
    seen = []
    for seed in range(6):
        rng = np.random.default_rng(seed)
        S = torch.tensor(((rng.random((1, 64, 64)) < 0.1) * rng.random((1, 64, 64))))
        op = as_operator(S)
        assert op.nnz == int(np.count_nonzero(S.numpy()))
        ref = csr_from_dense(S[0].numpy(), transpose=True)
        assert np.array_equal(op.fwd[0].col.numpy(), ref[1])
        seen.append(S.data_ptr())
        del S, op
        gc.collect()


def test_fused_kernels_are_spill_free():
    """The hop gather stream is one asm block (gcrnn_hop_asm.inc): no asm LDS read is in flight across compiler-scheduled code any
    more, so a spill can no longer capture a register before its data has landed -- it only costs time. (The compiler-scheduled
    macro streams of round 1 -- GCRNN_HOP_ASM=0 or GCRNN_STEP_WAVES != 8 -- are diagnostic builds only: gcrnn_fused_step.h refuses to
    compile them without -DGCRNN_DIAGNOSTIC_STREAMS, see tools/experiments/hop_asm_ab.sh.) The fused kernels sit at
    the 256-register budget of two waves per SIMD: enforce that no instantiation spills more than a few registers (two K = 5
    gate pre-pass instantiations spill 8 / 24 bytes per lane outside the stream), at build time (hipcc cross-compiles)."""
    import subprocess
    from gated_gcrnns_amd import build as b
    if not os.path.exists(b.HIPCC):
        pytest.skip('hipcc not available')
    import glob
    cur, bad, seen = None, [], 0
    procs = [subprocess.Popen([b.HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-c', src, '-o', os.devnull,
                               '-Rpass-analysis=kernel-resource-usage'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for src in sorted(glob.glob(os.path.join(b.CSRC, 'gcrnn_fused*.hip')))]      # the step kernel's instantiations span several files
    for p in procs:
        _, err = p.communicate()
        assert p.returncode == 0, err[-2000:]
        for line in err.splitlines():
            m = re.search(r'Function Name: (\S+)', line)
            if m:
                cur = m.group(1)
                continue
            m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
            if m and cur and ('fused_step_kernel' in cur or 'fused_wgrad_kernel' in cur or 'fused_seq_kernel' in cur or 'fused_seq32_kernel' in cur or 'fused_seq32p_kernel' in cur):
                seen += 1
                # the sequence-resident kernel (128 operand registers resident across every asm block) must not spill at all -- but for the
                # rank-1 variants of the time-gated recurrence and of the gate pre-pass (template arguments ..., GATED or MODE 1, R1 = true): the
                # per-node factor of their accumulators costs them 8-16 registers (32-64 bytes per lane); measured +70 % over the weighted path
                # (fused_seq32p_kernel, round 5: operand and accumulators are PINNED tuples, v[0:191]; a spill there would be a scratch reload + vmcnt(0)
                #  in front of the hop block)
                seq = 'fused_seq_kernel' in cur or 'fused_seq32_kernel' in cur or 'fused_seq32p_kernel' in cur
                r1_gated = 'fused_seq32_kernel' in cur and ('Lb1ELb1ELb0E' in cur or re.search(r'Li1ELb0ELb1ELb0E', cur) is not None)
                # (the weight-gradient kernel's fp32-image variant of uniform graphs, UNI = 1 -- only reached with GCRNN_NO_IMG16 -- re-fetches half of z
                #  per tap AND keeps the round-5 node-order image: 40-52 bytes per lane outside the asm stream)
                wgrad_uni1 = re.search(r'fused_wgrad_kernelILi\dELi\dELi\dELi1E', cur) is not None
                limit = 64 if (r1_gated or wgrad_uni1) else (0 if seq else 32)
                if int(m.group(1)) > limit:
                    bad.append((cur[:70], int(m.group(1))))
    assert seen >= 168 + 48 + 48 and not bad, bad      # (+ 48 instantiations of the wide sequence-resident kernel, + 48 of its hand-allocated-hop form)
