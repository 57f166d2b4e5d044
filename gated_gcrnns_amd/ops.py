"""torch.autograd bindings over the C ABI (include/gcrnn.h).

Everything here moves raw device pointers into libgcrnn_hip.so on the current
HIP stream; torch only owns the memory. CPU tensors are rejected: the product
has no CPU path (the CPU restatement lives in oracle/ and is test-only).

Node-major layout used by every op: X[T][N][B][C] (see include/gcrnn.h).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import lib, check, dtype_code, GcrnnError


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GcrnnError('gated_gcrnns_amd has no CPU path: move the module and its inputs to a ROCm '
                             'device (model.to("cuda"), x.cuda()); got a tensor on %s' % t.device)


# ------------------------------------------------------------------------------------------ layout
def _pack_raw(x, perm=None):
    B, T, Cc, N = x.shape
    out = torch.empty((T, N, B, Cc), dtype=x.dtype, device=x.device)
    check(lib.gcrnn_pack_node_major(dtype_code(x.dtype), _p(x), _p(out), B, T, Cc, N, _p(perm), _stream()), 'pack')
    return out


def _unpack_raw(X, perm=None):
    T, N, B, Cc = X.shape
    out = torch.empty((B, T, Cc, N), dtype=X.dtype, device=X.device)
    check(lib.gcrnn_unpack_node_major(dtype_code(X.dtype), _p(X), _p(out), B, T, Cc, N, _p(perm), _stream()), 'unpack')
    return out


class _Pack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _pack_raw(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return _unpack_raw(g.contiguous())


class _Unpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        return _unpack_raw(X.contiguous())

    @staticmethod
    def backward(ctx, g):
        return _pack_raw(g.contiguous())


def pack_node_major(x):
    """user [B][T][C][N] -> node-major [T][N][B][C]"""
    require_device(x)
    assert x.dim() == 4
    return _Pack.apply(x)


def unpack_node_major(X):
    """node-major [T][N][B][C] -> user [B][T][C][N]"""
    require_device(X)
    assert X.dim() == 4
    return _Unpack.apply(X)


# ------------------------------------------------------------------------------------------ shift
_SPMM_TUNE = {}      # {'piece_lanes': .., 'unroll': .., 'rows_per_wave': ..}: overrides for tuning sweeps (tools/spmm_sweep.py)


def spmm_auto_tune(csr, L, elt):
    """Piece width / loads in flight of the streaming SpMM from what the host knows (the kernel's own default knows neither the
    mean degree nor where the operand lives). Dense neighbourhoods on an operand far beyond L2 (cfg5: ~100 non-zeros per row,
    51 MB): 128-byte pieces, 8 slots x 8 loads (measured best, tools/spmm_sweep.py). Sparse rows (cfg2: ~10 per row) on wide
    node rows: whole-wave pieces and as many loads in flight as a row has neighbours, otherwise most slots idle."""
    deg = csr.nnz / max(csr.N, 1)
    row_bytes = L * elt
    lanes = max(4, min(64, 1 << max(0, (row_bytes // 16 - 1)).bit_length()))        # lanes that cover one row (power of two)
    if csr.N * row_bytes > (8 << 20) and deg >= 32:
        return dict(piece_lanes=min(lanes, 8), unroll=8, rows_per_wave=4)
    slots = 64 // lanes
    unroll = 8 if deg > 16 * slots else (4 if deg > 2 * slots else 2)                # measured at cfg2 (deg 10, 1 KiB pieces): 4 beats 8
    if unroll == 2 and lanes > 8:
        unroll = 4                                                                   # (2-deep variants are built for narrow pieces only)
    return dict(piece_lanes=lanes, unroll=unroll, rows_per_wave=1 if csr.N <= 4096 else 4)


def spmm_raw(csr, X, out=None, accumulate=False, bias=None, bias_scale=0.0, tanh=False, tune=None):
    """Y[i][n][:] (+)= sum_j val[j] X[i][col[j]][:] on a contiguous [nbatch][N][L...] tensor (bf16 / fp32 / fp64 rows; the
    CSR weights are fp32 for bf16 rows). tanh: Y = tanh(. + bias_scale * bias[l % F]) -- the last hop of a Horner-form step
    (bias: [F] in the weights' dtype or None). Rows that are whole 16-byte vectors run on the streaming kernel
    (gcrnn_spmm_ex); other shapes fall back to the scalar kernel (no epilogue there)."""
    N = csr.N
    assert X.is_contiguous() and X.dim() >= 3
    nbatch = X.shape[0]
    assert X.shape[1] == N, 'node-major tensor has %d rows, graph has %d nodes' % (X.shape[1], N)
    L = X[0, 0].numel()
    if out is None:
        assert not accumulate
        out = torch.empty_like(X)
    ve = 16 // X.element_size()
    if L % ve == 0 and X.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0:
        t = tune if tune is not None else (_SPMM_TUNE or spmm_auto_tune(csr, L, X.element_size()))
        F = bias.numel() if bias is not None else 0
        check(lib.gcrnn_spmm_ex(dtype_code(X.dtype), N, _p(csr.rowptr), _p(csr.col), _p(csr.val(X.dtype)), _p(X), _p(out),
                                L, nbatch, int(accumulate), _p(bias), float(bias_scale), F, int(tanh),
                                int(t.get('piece_lanes', 0)), int(t.get('unroll', 0)), int(t.get('rows_per_wave', 0)), _stream()), 'spmm_ex')
        return out
    check(lib.gcrnn_spmm(dtype_code(X.dtype), N, _p(csr.rowptr), _p(csr.col), _p(csr.val(X.dtype)), _p(X), _p(out),
                         L, nbatch, int(accumulate), _stream()), 'spmm')
    if tanh:
        if bias is not None:
            out.add_(bias_scale * bias.view(-1).repeat(L // bias.numel()).view(out.shape[2:]))
        torch.tanh(out, out=out)
    return out


def taps_bf16_supported(F, G, K):
    return bool(lib.gcrnn_taps_bf16_supported(int(F), int(G), int(K)))


def taps_bf16(zh, zx, wA, wB, out0=None):
    """All K = max(Kin, Kst) taps of a Horner-form step in one pass on the matrix cores: u_k = zh B_k^T + zx A_k^T.
    zh [..][F], zx [..][G] bf16 node-major rows (same leading shape; zx None for a state-only operand), wA F x 1 x Kin x G,
    wB F x 1 x Kst x F (bf16 or fp32). Returns (u_0 [..][F], u_rest [K-1][..][F]) bf16; out0: where tap 0 is written."""
    F = wB.shape[0]
    G = wA.shape[3] if zx is not None else 0
    K = max(wA.shape[2], wB.shape[2])
    st = _stream()
    if zx is not None:
        wpack = _fused_pack_weights(wA.detach(), wB.detach(), st)
    else:
        wpack = _fused_pack_state_taps(wB, K, st)
    zh = zh.contiguous()
    zx = zx.contiguous() if zx is not None else None
    R = zh.numel() // F
    if out0 is None:
        out0 = torch.empty_like(zh)
    rest = torch.empty((max(K - 1, 1),) + tuple(zh.shape), dtype=torch.bfloat16, device=zh.device)
    check(lib.gcrnn_taps_bf16_forward(_p(zh), _p(zx), _p(wpack), _p(out0), _p(rest), R, F, G, K, st), 'taps_bf16')
    return out0, rest


def taps_mfma_supported(dtype, F, Ch, Cx):
    return dtype in (torch.float32, torch.float64) and bool(lib.gcrnn_taps_mfma_supported(dtype_code(dtype), int(F), int(Ch), int(Cx)))


def taps_mfma(zh, zx, wA, wB, out0=None):
    """fp32 / fp64 counterpart of taps_bf16 on the fp32 / fp64 matrix cores (gcrnn_taps_mfma_forward): all
    K = max(Kin, Kst) taps u_k = zh B_k^T + zx A_k^T of a Horner-form step. zh [..][F], zx [..][G] node-major rows, wA
    F x 1 x Kin x G, wB F x 1 x Kst x F in the rows' dtype. Returns (u_0, u_rest [K-1][..][F])."""
    F, Ch = wB.shape[0], wB.shape[3]
    Cx = wA.shape[3] if zx is not None else 0
    Kst, Kin = wB.shape[2], (wA.shape[2] if zx is not None else 0)
    K = max(Kin, Kst)
    zh = zh.contiguous()
    zx = zx.contiguous() if zx is not None else None
    wBc = wB.detach().contiguous()
    wAc = wA.detach().contiguous() if zx is not None else None
    R = zh.numel() // Ch
    if out0 is None:
        out0 = torch.empty(tuple(zh.shape[:-1]) + (F,), dtype=zh.dtype, device=zh.device)
    rest = torch.empty((max(K - 1, 1),) + tuple(out0.shape), dtype=zh.dtype, device=zh.device)
    check(lib.gcrnn_taps_mfma_forward(dtype_code(zh.dtype), _p(zh), _p(zx), _p(wBc), _p(wAc), _p(out0), _p(rest), R, F, Ch, Cx,
                                      Kst, Kin, _stream()), 'taps_mfma')
    return out0, rest


def taps_rows(z, w, out=None, accumulate=False):
    """y[r][:] (+)= z[r][:] w^T on node-major rows with the LDS-tiled tap kernel (fp32 / fp64): z [..][C], w [F][C]."""
    Cin = z.shape[-1]
    F = w.shape[0]
    z, w = z.contiguous(), w.contiguous()
    rows = z.numel() // Cin
    if out is None:
        assert not accumulate
        out = torch.empty(tuple(z.shape[:-1]) + (F,), dtype=z.dtype, device=z.device)
    check(lib.gcrnn_taps_forward(dtype_code(z.dtype), _p(z), None, 0, _p(w), None, 0.0, _p(out), rows, 1, Cin, F,
                                 int(accumulate), _stream()), 'taps_forward')
    return out


class _Shift(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, fwd, adj):
        ctx.adj = adj
        return spmm_raw(fwd, X.contiguous())

    @staticmethod
    def backward(ctx, g):
        return spmm_raw(ctx.adj, g.contiguous()), None, None


def graph_shift(X, graph, e=0):
    """One application of the GSO on node-major data: the reference's x @ S_e (graphML.py:123)."""
    require_device(X)
    return _Shift.apply(X, graph.fwd[e], graph.adj[e])


# ------------------------------------------------------------------------------------------ LSIGF
class _LSIGF(torch.autograd.Function):
    """Node-major linear shift-invariant graph filter (graphML.py:47-140) for all E, K.

    forward : hops z_{e,k} = P_e z_{e,k-1} (CSR(S_e^T) SpMM), then one tap GEMM per edge feature.
    backward: dz = dy W (tap GEMM), reverse hop chain with CSR(S_e) in Horner form, dW by a
              row-reduction GEMM. The hop matrices are saved, not recomputed.
    """

    @staticmethod
    def forward(ctx, X, w, bias, graph, bias_scale):
        T, N, B, G = X.shape
        F, E, K, Gw = w.shape
        assert Gw == G and E == graph.E and N == graph.N
        dt = dtype_code(X.dtype)
        X = X.contiguous()
        rows = T * N * B
        zstride = rows * G
        zrest = torch.empty((E, max(K - 1, 1), T, N, B, G), dtype=X.dtype, device=X.device) if K > 1 else None
        y = torch.empty((T, N, B, F), dtype=X.dtype, device=X.device)
        wc = w.contiguous()
        bvec = bias.contiguous().view(-1) if bias is not None else None
        st = _stream()
        for e in range(E):
            src = X
            for k in range(1, K):
                spmm_raw(graph.fwd[e], src, out=zrest[e, k - 1])
                src = zrest[e, k - 1]
            we = wc[:, e].contiguous() if E > 1 else wc
            check(lib.gcrnn_taps_forward(dt, _p(X), _p(zrest[e]) if K > 1 else None, zstride, _p(we),
                                         _p(bvec) if e == 0 else None, float(bias_scale), _p(y), rows, K, G, F,
                                         int(e > 0), st), 'taps_forward')
        ctx.save_for_backward(X, zrest, wc)
        ctx.graph = graph
        ctx.has_bias = bias is not None
        ctx.bias_shape = tuple(bias.shape) if bias is not None else None
        ctx.bias_scale = float(bias_scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        X, zrest, wc = ctx.saved_tensors
        graph = ctx.graph
        T, N, B, G = X.shape
        F, E, K, _ = wc.shape
        dt = dtype_code(X.dtype)
        dy = dy.contiguous()
        rows = T * N * B
        zstride = rows * G
        st = _stream()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2] and ctx.has_bias
        dX = None
        dW = torch.zeros_like(wc) if (need_w and E > 1) else None
        db = None
        for e in range(E):
            we = wc[:, e].contiguous() if E > 1 else wc
            if need_x:
                dz0 = torch.empty_like(X)
                dzr = torch.empty((max(K - 1, 1), T, N, B, G), dtype=X.dtype, device=X.device) if K > 1 else None
                check(lib.gcrnn_taps_backward_data(dt, _p(dy), _p(we), _p(dz0), _p(dzr), zstride, rows, K, G, F, st),
                      'taps_backward_data')
                # dX += dz0 + A(dz1 + A(dz2 + ... A dz_{K-1}))   with A = adjoint shift = CSR(S_e) SpMM
                for k in range(K - 1, 0, -1):
                    dst = dzr[k - 2] if k >= 2 else dz0
                    spmm_raw(graph.adj[e], dzr[k - 1], out=dst, accumulate=True)
                dX = dz0 if dX is None else dX.add_(dz0)
            if need_w or need_b:
                # per-split partial sums, added in a fixed order: the weight gradient is bit-reproducible
                nsp, nbb = C.c_int64(0), C.c_int64(0)
                check(lib.gcrnn_taps_backward_weight_parts(rows, K, G, F, C.byref(nsp), C.byref(nbb)), 'taps_backward_weight_parts')
                dwp = torch.empty((nsp.value, F, K, G), dtype=X.dtype, device=X.device)
                dbp = torch.empty((nbb.value, F), dtype=X.dtype, device=X.device) if (need_b and e == 0) else None
                check(lib.gcrnn_taps_backward_weight(dt, _p(dy), _p(X), _p(zrest[e]) if K > 1 else None, zstride,
                                                     _p(dwp), _p(dbp), ctx.bias_scale, rows, K, G, F, st), 'taps_backward_weight')
                if need_w:
                    if E == 1:
                        dW = dwp.sum(dim=0).view(F, 1, K, G)
                    else:
                        dW[:, e] = dwp.sum(dim=0)
                if dbp is not None:
                    db = dbp.sum(dim=0)
        if db is not None:
            db = db.view(ctx.bias_shape)
        return dX, dW, db, None, None


def lsigf_node_major(X, w, bias, graph, bias_scale=1.0):
    """X: [T][N][B][G] node-major -> [T][N][B][F]; w: F x E x K x G; bias: F x 1 or None."""
    require_device(X, w, bias)
    if w.dtype != X.dtype:
        raise GcrnnError('filter taps are %s but the signal is %s' % (w.dtype, X.dtype))
    if X.dtype == torch.bfloat16:
        # the any-shape filter kernels are fp32 / fp64 (bf16 lives in the cell's own kernels): a stand-alone bf16 filter is evaluated in fp32 and
        # rounded once at the end -- differentiable, the casts' backward returns bf16 gradients
        y = _LSIGF.apply(X.float(), w.float(), bias.float() if bias is not None else None, graph, bias_scale)
        return y.to(torch.bfloat16)
    return _LSIGF.apply(X, w, bias, graph, bias_scale)


# ------------------------------------------------------------------------------------------ fused flagship path
def fused_padded_inputs(F, G):
    """Input-feature count the fused kernels run with for a cell with G input and F state features: the x operand is
    consumed in 32-feature MFMA steps, so G is zero-padded to 32 (or 64) channels -- the reference drivers' G = 1
    (kStepPredGRNNs.py:220, epicenterEstimation.py:170) runs as 32. None: no kernel for this pair."""
    if F in (32, 64) and 0 < G <= 32:
        return 32
    if F == 64 and 32 < G <= 64:
        return 64
    return None


def fused_pad_operands(X, wA):
    """Zero-pad the input sequence X [B][T][G][N] and the input taps wA [F][1][K][G] to the kernels' channel count
    (differentiable w.r.t. wA: the padding's gradient is dropped by autograd)."""
    F, G = wA.shape[0], wA.shape[3]
    Gp = fused_padded_inputs(F, G)
    if Gp == G:
        return X, wA
    B, T, _, N = X.shape
    Xp = X.new_zeros((B, T, Gp, N))
    Xp[:, :, :G] = X
    return Xp, torch.nn.functional.pad(wA, (0, Gp - G))


def fused_supported(N, F, G, Kin, Kst, dtype, E=1):
    Gp = fused_padded_inputs(F, G)
    return (E == 1 and dtype == torch.bfloat16 and Gp is not None and
            bool(lib.gcrnn_fused_supported(int(N), int(F), int(Gp), int(max(Kin, Kst)))))


def time_fused_step_kernel(X, h0, wA, wB, bias, graph, reps=3, inline=None, user_layout=True, warm=1):
    """Average duration of ONE fused step launch, measured with HIP events on the launch stream
    (inputs pre-packed, only the T step launches sit between the events; on uniform-weight graphs each launch also lays out
    x_{t+1}, exactly as in fused_cell_forward). warm: untimed launches queued in front of the timed ones without a gap (behind an idle
    period the chip runs its first ~30 ms of load 10-25 % slower than its steady state, profiles/r04_clock_transient.txt)."""
    X, wA = fused_pad_operands(X, wA.detach())
    B, T, G, N = X.shape
    F = wA.shape[0]
    Kin, Kst = wA.shape[2], wB.shape[2]
    K = max(Kin, Kst)
    plan = graph.fused_plan()
    npad = plan['npad']
    st = _stream()
    dev = X.device
    xs = torch.empty((T, B, npad, G), dtype=torch.bfloat16, device=dev)
    h0s = torch.empty((1, B, npad, F), dtype=torch.bfloat16, device=dev)
    hs = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=dev)
    Xc, h0c = X.contiguous(), h0.contiguous()      # named: a temporary would be freed (and reusable) before the launch
    check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(Xc), _p(xs), B, T, G, N, npad, None, st), 'pack_seq')
    check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(h0c), _p(h0s), B, 1, F, N, npad, None, st), 'pack_seq')
    wpack = torch.empty(((F // 16) * K * ((F + G) // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
    wAc, wBc = wA.contiguous(), wB.contiguous()
    check(lib.gcrnn_fused_pack_weights(dtype_code(wA.dtype), _p(wAc), _p(wBc), _p(wpack),
                                       F, G, Kin, Kst, st), 'pack_weights')
    b32 = bias.detach().float().contiguous().view(-1) if bias is not None else None
    # same launch configuration as fused_cell_forward: the kernel also writes the user-layout output when it can
    # (user_layout=False: sequence-major in and out only -- what GGCRNNCell.forward_native issues)
    H = torch.empty((B, T, F, N), dtype=torch.bfloat16, device=dev) if (N % 8 == 0 and user_layout) else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ok = fused_inline_pack_ok(plan, N, F, G, K)          # the launches as the forward issues them: with the inline pack of x_{t+1} where it
    inline_arg = inline
    inline = (ok if inline is None else (ok and inline)) and X.data_ptr() % 16 == 0     # applies; inline=False times the bare step for comparison
    plan16 = fused_img16_plan(graph, False, None)
    wide = fused_wide_plan(graph, B, T, N, F, G, K, inline)
    if wide is None and plan16 is None:                  # rank-1-weighted graph (normalised adjacency): the wide kernel on the pattern plan, as fused_cell_forward issues it
        inl1 = inline_arg is not False and not os.environ.get('GCRNN_NO_INLINE_PACK') and X.data_ptr() % 16 == 0
        wide = fused_wide_plan(graph, B, T, N, F, G, K, inl1, rank1=True)
        if wide is None and inl1:
            inl1 = False
            wide = fused_wide_plan(graph, B, T, N, F, G, K, False, rank1=True)
        if wide is not None:
            inline = inl1
    if wide is not None:
        # the wide sequence-resident kernel (what fused_cell_forward issues for this problem): ONE launch per forward
        wpw = _fused_pack_weights_wide(wAc, wBc, wide['uniform_w'], st)
        for rep in range(reps + warm):
            if rep == warm:
                e0.record()                              # (no synchronisation here: the timed launches follow the warm ones without an idle gap)
            _fused_forward_wide(wide, xs, h0s, hs, wAc, wBc, b32, B, T, N, F, G, K, H, False, Xc if inline else None, st, wpw=wpw)
        e1.record()
        torch.cuda.synchronize()
        per_step = 1e3 * e0.elapsed_time(e1) / (reps * T)
        return {'avg_us': per_step, 'launches': reps, 'inline_pack': bool(inline), 'steps_per_launch': T,
                'launch_avg_us': per_step * T, 'kernel': 'fused_seq32_kernel'}
    for rep in range(reps + 1):
        if rep == 1:                                     # (launch 0 is a warm-up: first touch of the fresh output buffers, code and plan not yet in cache)
            torch.cuda.synchronize()
            e0.record()
        check(lib.gcrnn_fused_forward_bf16(_p(xs), _p(h0s), _p(hs), _p(wpack), _p(b32), None, None, *_fused_graph_args(plan16 or plan),
                                           B, T, N, F, G, K, _p(H) if H is not None else None, 2 if plan16 else 0, None, plan.get('uniform_w', 0.0),
                                           _p(Xc) if inline else None, None, None, st),
              'fused_forward')
    e1.record()
    torch.cuda.synchronize()
    # time steps one launch covers: T on the sequence-resident persistent kernel (gcrnn_fused_seq.h), else 1 (one launch per step)
    spl = int(lib.gcrnn_fused_seq_steps_per_launch(B, T, N, F, G, K, int((plan16 or plan)['entries']), float(plan.get('uniform_w', 0.0)),
                                                   1 if plan16 else 0, 1 if inline else 0, 0))
    per_step = 1e3 * e0.elapsed_time(e1) / (reps * T)
    return {'avg_us': per_step, 'launches': reps * T // max(spl, 1), 'inline_pack': bool(inline), 'steps_per_launch': max(spl, 1),
            'launch_avg_us': per_step * max(spl, 1), 'kernel': 'fused_seq_kernel' if spl else 'fused_step_kernel'}


_PACK_CACHE = {}
_PARAM_EPOCH = [0]
_PACK_CACHE_ON = [bool(os.environ.get('GCRNN_PACK_CACHE'))]


def parameters_changed():
    """Tell the (opt-in) pack cache that parameters were written through a path PyTorch does not see: a kernel that updates them through raw
    pointers (optim.FlatAdam's flat Adam step, a replayed hipGraph of a whole training step) leaves their autograd version counters where they
    were. Every such writer calls this; the epoch is part of every key. Without `freeze_parameters()` nothing is cached and this is a no-op."""
    _PARAM_EPOCH[0] += 1


def freeze_parameters(on=True):
    """Opt in to (or out of) keeping the PACKED forms of parameters -- tap fragments, the fp32 bias, concatenated gate weights -- between
    forwards. The default is OFF: every forward packs from the live parameters (three ~5 us kernels, ~1 % of a forward at the bench size), so
    ANY write PyTorch permits is seen by the next forward -- including `p.data.mul_(2)`, `p.data.copy_(t)` and `dist.broadcast(p.data, 0)`,
    which move no version counter the host could key on (the reference's own `reset_parameters`, Utils/graphML.py:2229-2235, writes that way).
    With the cache on the caller promises not to write parameters through `.data` or raw pointers without calling `parameters_changed()`;
    in-place updates under `torch.no_grad()`, `copy_`, `load_state_dict` and tensor swaps are still detected (storage pointer + version
    counter in the key). GCRNN_PACK_CACHE=1 in the environment switches it on from the start. Returns the previous setting."""
    prev = _PACK_CACHE_ON[0]
    _PACK_CACHE_ON[0] = bool(on)
    if not on:
        _PACK_CACHE.clear()
    return prev


class frozen_parameters(object):
    """`with ops.frozen_parameters(): ...` -- freeze_parameters(True) for the block (an inference loop over fixed weights)."""

    def __enter__(self):
        self._prev = freeze_parameters(True)
        return self

    def __exit__(self, *exc):
        freeze_parameters(self._prev)
        return False


def _cached_pack(kind, tensors, extra, st, make):
    """Packed forms of PARAMETERS (tap fragments, the fp32 bias). DEFAULT: made on every call from the live parameters -- correct for every
    way PyTorch lets a parameter be written (see freeze_parameters). With `freeze_parameters()` / GCRNN_PACK_CACHE=1 they are kept between
    calls while the parameters are unchanged: the key holds each tensor's storage pointer, shape, strides, dtype and autograd version counter
    plus the parameter epoch, and the entry keeps the tensors alive so that their storage cannot be recycled under the key. Never during a
    stream capture: a miss there would store a buffer whose pack kernel was only RECORDED, and a second capture of the same cell would hit it
    and record no pack at all (its replays would read memory another graph fills) -- every captured graph packs for itself, so its replays
    always see the live weights. Temporaries made under torch.inference_mode (no version counter) are never cached."""
    if (not _PACK_CACHE_ON[0] or os.environ.get('GCRNN_NO_PACK_CACHE') or torch.cuda.is_current_stream_capturing()
            or any(t.is_inference() for t in tensors)):
        return make()
    key = (kind, extra, st.value, _PARAM_EPOCH[0]) + tuple((t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype) for t in tensors)
    hit = _PACK_CACHE.get(key)
    if hit is not None:
        return hit[0]
    out = make()
    while len(_PACK_CACHE) >= 64:
        _PACK_CACHE.pop(next(iter(_PACK_CACHE)))
    _PACK_CACHE[key] = (out, tensors)
    return out


def _bias_f32(bias, st):
    """[F] fp32 copy of the bias (or None), cached while the parameter is unchanged."""
    if bias is None:
        return None
    b = bias.detach()
    if b.dtype == torch.float32 and b.is_contiguous():
        return b.view(-1)
    return _cached_pack('bias', (b,), None, st, lambda: b.float().contiguous().view(-1))


def _fused_pack_weights(wA, wB, st):
    F, G = wA.shape[0], wA.shape[3]
    Kin, Kst = wA.shape[2], wB.shape[2]
    K = max(Kin, Kst)

    def make():
        wpack = torch.empty(((F // 16) * K * ((F + G) // 32) * 64 * 8,), dtype=torch.bfloat16, device=wA.device)
        wAc, wBc = wA.contiguous(), wB.contiguous()
        check(lib.gcrnn_fused_pack_weights(dtype_code(wA.dtype), _p(wAc), _p(wBc), _p(wpack),
                                           F, G, Kin, Kst, st), 'pack_weights')
        return wpack
    return _cached_pack('taps16', (wA.detach(), wB.detach()), None, st, make)


def _fused_pack_weights_wide(wA, wB, uniform_w, st):
    """Taps as the MFMA A fragments of the wide sequence-resident kernel (csrc/gcrnn_fused_seq32.h): 32-feature chunks, feature
    permutation of the rows, tap k scaled by uniform_w^k. wA [Fout][1][Kin][G], wB [Fout][1][Kst][F]: Fout = F for a cell, 2 F for the
    two time gates' sub-cells stacked over their output features."""
    Fout, G, F = wA.shape[0], wA.shape[3], wB.shape[3]
    Kin, Kst = wA.shape[2], wB.shape[2]
    K = max(Kin, Kst)

    def make():
        wpack = torch.empty(((Fout // 32) * K * 2 * ((F + G) // 32) * 64 * 8,), dtype=torch.bfloat16, device=wA.device)
        wAc, wBc, kin = wA.contiguous(), wB.contiguous(), Kin
        if G == 0:                                   # state-only operand (the BPTT chain's transposed taps): no input taps to read
            wAc, kin = wBc, Kst
        check(lib.gcrnn_fused_pack_weights_wide(dtype_code(wB.dtype), _p(wAc), _p(wBc), _p(wpack), Fout, F, G, kin, Kst, float(uniform_w), st),
              'pack_weights_wide')
        return wpack
    return _cached_pack('taps32', (wA.detach(), wB.detach()), float(uniform_w), st, make)


def fused_wide_plan(graph, B, T, N, F, G, K, inline, rank1=False, gated=False):
    """The bf16-image plan when the wide sequence-resident kernel (gcrnn_fused_forward_wide_bf16: un-gated forward as ONE launch, 32-feature
    chunks) takes this problem, else None. GCRNN_SEQ32=0 switches it off (A/B)."""
    plan16 = fused_img16_plan(graph, False, None)
    if plan16 is None and rank1 and not os.environ.get('GCRNN_NO_IMG16'):
        plan16 = graph.fused_plan_rank1()                # rank-1-weighted graph (normalised adjacency): the plan of its 0/1 pattern + the two factor tables
    if plan16 is None or F % 32 or G % 32:
        return None
    ok = lib.gcrnn_fused_forward_wide_supported(int(B), int(T), int(N), int(F), int(G), int(K), int(plan16['entries']),
                                                float(plan16.get('uniform_w', 0.0)), (3 if plan16.get('rank1') else 1) | (4 if gated else 0), 1 if inline else 0)
    return plan16 if ok else None


def _fused_forward_wide(plan16, xs, h0s, hs, wA, wB, b32, B, T, N, F, G, K, H, last_only, Xinline, st, wpw=None, gi=None, gf=None):
    if wpw is None:
        wpw = _fused_pack_weights_wide(wA.detach(), wB.detach(), plan16['uniform_w'], st)
    check(lib.gcrnn_fused_forward_wide_bf16(_p(xs), _p(h0s), _p(hs), _p(wpw), _p(b32), _p(gi), _p(gf), _p(plan16['tile_slots']), _p(plan16['tile_off']),
                                            _p(plan16['ell_col4']), plan16['entries'], B, T, N, F, G, K,
                                            _p(H) if H is not None else None, int(bool(last_only)), _p(Xinline) if Xinline is not None else None,
                                            _p(plan16.get('rank1_a')), _p(plan16.get('rank1_b')), st),
          'fused_forward_wide')


def fused_img16_plan(graph, gated, head):
    """The bf16-image plan of the un-gated forward steps (GraphOperator.fused_plan_img16), or None: uniform-weight graphs only
    (un-gated and time-gated cells, with or without the fused head); GCRNN_NO_IMG16=1 switches it off (A/B, and the fp32-image tests)."""
    import os
    if os.environ.get('GCRNN_NO_IMG16'):
        return None
    return graph.fused_plan_img16()


def _fused_graph_args(plan):
    return (_p(plan['tile_slots']), _p(plan['tile_off']), _p(plan['ell_addr']), _p(plan['ell_val']),
            _p(plan['ell_val4']), _p(plan['ell_col4']), plan['entries'])


_SIDE_STREAMS = {}


def _side_stream(dev):
    s = _SIDE_STREAMS.get(dev)
    if s is None:
        s = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


def fused_inline_pack_ok(plan, N, F, G, K):
    """The step kernel can lay out x_{t+1} itself (gcrnn_fused_forward_bf16 with Xuser_inline): uniform-weight graph image with LDS
    room for the input tile, N % 8 == 0. GCRNN_NO_INLINE_PACK=1 switches it off (A/B)."""
    if os.environ.get('GCRNN_NO_INLINE_PACK'):
        return False
    return bool(lib.gcrnn_fused_inline_pack_supported(int(N), int(F), int(G), int(K), int(plan['entries']), float(plan.get('uniform_w', 0.0))))


def fused_pad_taps(wA):
    """Input taps wA [F][1][K][G] zero-padded to the kernels' input width (differentiable; see fused_pad_operands)."""
    F, G = wA.shape[0], wA.shape[3]
    Gp = fused_padded_inputs(F, G)
    return wA if Gp == G else torch.nn.functional.pad(wA, (0, Gp - G))


def fused_pack_inputs(X, h0, graph, overlap=False, first_only=False, channels=None):
    """user-layout bf16 X [B][T][G][N], h0 [B][F][N] -> sequence-major xs [T][B][NPad][G] and the state buffer
    hs_all [T+1][B][NPad][F] whose slot 0 holds h0 (slots 1..T receive h_1..h_T: hs_all[:T] is then the h_{t-1} operand of
    every step, which the gate-gradient pass reads as one array).
    overlap: only step 0 is packed on the current stream; every later step is packed on a side stream by a kernel small
    enough (4 KiB of LDS, bounded grid) to run beside the step kernels, each followed by an event. Returns (xs, hs_all,
    events) with events[t] = the event step t's launch has to wait for -- the recurrence starts after 1/T of the pack."""
    B, T, G, N = X.shape
    F = h0.shape[1]
    npad = graph.fused_plan()['npad']
    st = _stream()
    Xc, h0c = X.contiguous(), h0.contiguous()
    if channels is not None and channels != G:
        # X keeps its own G channels in the user layout; the sequence-major array gets `channels` of them, the rest zeros
        assert channels > G and not overlap and not first_only
        xs = torch.empty((T, B, npad, channels), dtype=torch.bfloat16, device=X.device)
        hs_all = torch.empty((T + 1, B, npad, F), dtype=torch.bfloat16, device=X.device)
        check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(h0c), _p(hs_all), B, 1, F, N, npad, None, st), 'pack_seq')
        check(lib.gcrnn_pack_seq_major_padded(_p(Xc), _p(xs), B, T, G, channels, N, npad, st), 'pack_seq_padded')
        return xs, hs_all
    xs = torch.empty((T, B, npad, G), dtype=torch.bfloat16, device=X.device)
    hs_all = torch.empty((T + 1, B, npad, F), dtype=torch.bfloat16, device=X.device)
    check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(h0c), _p(hs_all), B, 1, F, N, npad, None, st), 'pack_seq')
    if first_only:                                       # x_0 and x_1 only: the step kernels lay out every later step themselves (inline pack;
        # the sequence-resident kernel works two steps ahead, the chunk-parallel one re-writes x_1 with the same bits)
        check(lib.gcrnn_pack_seq_major_steps(_p(Xc), _p(xs), B, T, G, N, npad, 0, min(2, T), 0, st), 'pack_seq_steps')
        return xs, hs_all
    if not overlap:
        check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(Xc), _p(xs), B, T, G, N, npad, None, st), 'pack_seq')
        return xs, hs_all
    main = torch.cuda.current_stream(X.device)
    side = _side_stream(X.device)
    check(lib.gcrnn_pack_seq_major_steps(_p(Xc), _p(xs), B, T, G, N, npad, 0, 1, 0, st), 'pack_seq_steps')
    ready = torch.cuda.Event()
    ready.record(main)                                   # X and the fresh buffers are valid on the main stream from here on
    side.wait_event(ready)
    events = [None] * T
    sst = C.c_void_p(side.cuda_stream)
    # one workgroup per CU walks the tiles of a step: measured on MI355X the side-stream pack of step t+1 then takes about as
    # long as step t (118 us vs 120 us at B = 256) and slows it by 9 %; an unbounded grid starves the step kernels of memory
    # bandwidth (steps 1.2x .. 3.2x slower: no net gain), half the grid makes the pack the bottleneck
    blocks = int(os.environ.get('GCRNN_PACK_BLOCKS', torch.cuda.get_device_properties(X.device).multi_processor_count))
    for t in range(1, T):
        check(lib.gcrnn_pack_seq_major_steps(_p(Xc), _p(xs), B, T, G, N, npad, t, t + 1, blocks, sst), 'pack_seq_steps')
        ev = torch.cuda.Event()
        ev.record(side)
        events[t] = ev
    Xc.record_stream(side)
    xs.record_stream(side)
    return xs, hs_all, events


def fused_pack_inputs_gated(X, h0, graph, F, K):
    """fused_pack_inputs for a cell whose FIRST consumer of xs is a gate pre-pass (fused_time_gate): on a uniform-weight graph with
    a batch that fills the chip, only the first time step(s) are laid out here and that pre-pass lays out the rest while it runs
    (gcrnn_fused_gate_prepass_pack_bf16: every item packs the operand of its workgroup's next item) -- the 0.5 ms pass over X at
    B = 256, T = 32 goes away. The pending user-layout tensor travels with xs (`_pending_user`) until that pre-pass has run."""
    B, T, G, N = X.shape
    plan = graph.fused_plan()
    plan16 = fused_img16_plan(graph, True, None)
    steps = 0
    if plan16 is not None and X.dtype == torch.bfloat16 and X.is_contiguous() and X.data_ptr() % 16 == 0 and not os.environ.get('GCRNN_NO_INLINE_PACK'):
        _, steps = fused_gate_pair_plan(graph, B, T, N, F, G, K, True)      # (the wide kernel's gate-pair pre-pass lays X out, when it takes the problem)
        if steps <= 0:
            steps = int(lib.gcrnn_fused_gate_prepass_lays_out(B, T, N, F, G, K, plan16['entries'], plan.get('uniform_w', 0.0), 1))
    if steps <= 0 or steps >= T:
        return fused_pack_inputs(X, h0, graph)
    npad = plan['npad']
    st = _stream()
    xs = torch.empty((T, B, npad, G), dtype=torch.bfloat16, device=X.device)
    hs_all = torch.empty((T + 1, B, npad, F), dtype=torch.bfloat16, device=X.device)
    check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(h0.contiguous()), _p(hs_all), B, 1, F, N, npad, None, st), 'pack_seq')
    check(lib.gcrnn_pack_seq_major_steps(_p(X), _p(xs), B, T, G, N, npad, 0, steps, 0, st), 'pack_seq_steps')
    xs._pending_user = X
    xs._pending_steps = steps
    return xs, hs_all


def _pending_layout_for(xs, consumer_steps):
    """The user-layout tensor a gate pre-pass should lay out while it runs (fused_pack_inputs_gated), or None. The pending layout was sized
    for the pre-pass that was expected to run first; when another one runs instead -- it does not lay out at this shape (consumer_steps
    <= 0), or counts on a different number of leading steps -- the rest of X is laid out here, by the plain pack, and nothing is pending."""
    x_user = getattr(xs, '_pending_user', None)
    if x_user is None:
        return None
    done = int(getattr(xs, '_pending_steps', 0))
    if consumer_steps > 0 and consumer_steps == done:
        return x_user
    T, B, npad, G = xs.shape
    check(lib.gcrnn_pack_seq_major_steps(_p(x_user), _p(xs), B, T, G, x_user.shape[3], npad, done, T, 0, _stream()), 'pack_seq_steps')
    del xs._pending_user
    return None


def fused_overlap_ok(X):
    """The overlapped pack (opt-in: GCRNN_FUSED_OVERLAP=1) needs even N and G (4-byte accesses), more than one step, and no
    stream capture in progress. Off by default: measured on MI355X it gains 3 % when the two streams happen to share the CUs
    well (4.04 -> 3.90 ms per forward at B = 256) and loses up to 2x when they do not (same binary, another run) -- DESIGN 4.3."""
    B, T, G, N = X.shape
    return (T > 1 and N % 2 == 0 and G % 2 == 0 and os.environ.get('GCRNN_FUSED_OVERLAP', '0') == '1'
            and not torch.cuda.is_current_stream_capturing())


def fused_h0_zero_flag(h0):
    """int32 [1] on the device: 1 when the initial state is all zeros (every training loop of the reference starts from
    zeros(B, F, N), train_rnn.py:256). The gate kernels read it and skip the state operand's loads and matrix products -- a
    data-dependent short cut with bit-identical results, decided on the device (no host synchronisation)."""
    if h0.dtype == torch.bfloat16 and h0.is_contiguous() and h0.numel() % 8 == 0 and h0.data_ptr() % 16 == 0 and h0.is_cuda:
        flag = torch.empty(1, dtype=torch.int32, device=h0.device)
        check(lib.gcrnn_all_zero_flag_bf16(_p(h0), h0.numel(), _p(flag), _stream()), 'all_zero_flag')      # one pass at HBM rate, no temporaries
        return flag
    return (h0 == 0).all().to(torch.int32).view(1)


def fused_time_gate(xs, h0s, wA_g, wB_g, bias_g, lin_w, lin_b, graph, N, store_states=False, hzero=None):
    """One time gate of the fused path for all (t, b) (graphML.py:2357-2374): sigmoid(Linear(vec(tanh(A_g(S)x_t + B_g(S)h0
    + 2 b_g)))) as ONE pre-pass launch. xs [T][B][NPad][G], h0s [1][B][NPad][F] sequence-major bf16. hzero: device int32
    flag (fused_h0_zero_flag), non-zero when h0 is all zeros -- the kernel then skips the state half of the operand. Returns the gate
    [T][B] fp32 (and, with store_states, the gate cell's states c [T][B][NPad][F] bf16 for its BPTT)."""
    T, B, npad, G = xs.shape
    F = wA_g.shape[0]
    K = max(wA_g.shape[2], wB_g.shape[2])
    plan = graph.fused_plan()
    st = _stream()
    wp = _fused_pack_weights(wA_g.detach(), wB_g.detach(), st)
    bg = bias_g.detach().float().contiguous().view(-1) if bias_g is not None else None
    gw = lin_w.detach().float().view(F, N).t().contiguous()                  # row-major vec over (f, n) -> [N][F]
    parts = torch.empty((T * B, (F // 16) * int(lib.gcrnn_fused_step_waves())), dtype=torch.float32, device=xs.device)
    cs = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=xs.device) if store_states else None
    plan16 = fused_img16_plan(graph, True, None)
    x_user = None
    if getattr(xs, '_pending_user', None) is not None:
        lays = int(lib.gcrnn_fused_gate_prepass_lays_out(B, T, N, F, G, K, plan16['entries'], plan.get('uniform_w', 0.0), 1)) if plan16 is not None else 0
        x_user = _pending_layout_for(xs, lays)
    if x_user is not None:
        # xs holds its first time step(s) only (fused_pack_inputs_gated): this pre-pass lays out the rest
        check(lib.gcrnn_fused_gate_prepass_pack_bf16(_p(x_user), _p(xs), _p(h0s), _p(wp), _p(bg), _p(gw), _p(parts), _p(cs),
                                                     *_fused_graph_args(plan16), B, T, N, F, G, K, _p(hzero), plan.get('uniform_w', 0.0),
                                                     1, st), 'gate_prepass_pack')
        del xs._pending_user
    else:
        check(lib.gcrnn_fused_gate_prepass_bf16(_p(xs), _p(h0s), _p(wp), _p(bg), _p(gw), _p(parts), _p(cs),
                                                *_fused_graph_args(plan16 or plan), B, T, N, F, G, K, _p(hzero), plan.get('uniform_w', 0.0),
                                                1 if plan16 else 0, st), 'gate_prepass')
    acc = parts.sum(dim=1)                                                    # fixed order: deterministic gates
    if lin_b is not None:
        acc = acc + lin_b.detach().float()
    gate = torch.sigmoid(acc).view(T, B).contiguous()
    return (gate, cs, gw) if store_states else gate


def fused_gate_pair_plan(graph, B, T, N, F, G, K, with_pack):
    """(plan16, steps) when the wide kernel's gate-PAIR pre-pass takes this problem (gcrnn_fused_gate_pair_prepass_wide_bf16: both time
    gates of every (t, b) in ONE launch), else (None, 0). steps: with_pack -- the leading time steps of xs the caller lays out itself."""
    plan16 = fused_img16_plan(graph, True, None)
    if plan16 is None and not os.environ.get('GCRNN_NO_IMG16'):
        plan16 = graph.fused_plan_rank1()                # rank-1-weighted graph (normalised adjacency): the plan of its 0/1 pattern + the factor tables
    if plan16 is None or F % 32 or G % 32 or os.environ.get('GCRNN_NO_GATE_PAIR'):
        return None, 0
    steps = int(lib.gcrnn_fused_gate_pair_wide_supported(int(B), int(T), int(N), int(F), int(G), int(K), int(plan16['entries']),
                                                         float(plan16.get('uniform_w', 0.0)), 3 if plan16.get('rank1') else 1, 1 if with_pack else 0))
    return (plan16, steps) if steps > 0 else (None, 0)


def fused_time_gate_pair(xs, h0s, gate_in, gate_f, graph, N, store_states=False, hzero=None):
    """BOTH time gates of the fused path for all (t, b) (graphML.py:2357-2374) as ONE pre-pass launch of the wide sequence-resident kernel: the
    two sub-cells run as one cell of 2 F outputs, so an item's operand (x_t, h0) is loaded -- and, when xs still waits for its layout
    (`_pending_user`), laid out -- once. gate_* = (wA_g, wB_g, bias_g, lin_w, lin_b). Returns (gi, gf) [T][B] fp32, and with store_states
    also ((cs_in, gw_in), (cs_f, gw_f)): the sub-cells' states [T][B][NPad][F] bf16 and the read-out weights [N][F] fp32 for their BPTT."""
    T, B, npad, G = xs.shape
    F = gate_in[0].shape[0]
    K = max(gate_in[0].shape[2], gate_in[1].shape[2])
    st = _stream()
    x_user = None
    if getattr(xs, '_pending_user', None) is not None:
        x_user = _pending_layout_for(xs, fused_gate_pair_plan(graph, B, T, N, F, G, K, True)[1])
    plan16, _ = fused_gate_pair_plan(graph, B, T, N, F, G, K, x_user is not None)
    assert plan16 is not None
    def prepare():      # everything derived from the gates' parameters alone: cached while they are unchanged (_cached_pack)
        wA2 = torch.cat([gate_in[0].detach(), gate_f[0].detach()], dim=0)
        wB2 = torch.cat([gate_in[1].detach(), gate_f[1].detach()], dim=0)
        wp_ = _fused_pack_weights_wide(wA2, wB2, plan16['uniform_w'], st)
        zb = torch.zeros(F, dtype=torch.float32, device=xs.device)
        b2_ = torch.cat([(g[2].detach().float().reshape(-1) if g[2] is not None else zb) for g in (gate_in, gate_f)]).contiguous()
        gws_ = [g[3].detach().float().view(F, N).t().contiguous() for g in (gate_in, gate_f)]      # row-major vec over (f, n) -> [N][F]
        lbs_ = [(g[4].detach().float() if g[4] is not None else None) for g in (gate_in, gate_f)]
        return wp_, b2_, gws_, torch.stack(gws_, dim=0).contiguous(), lbs_
    ptens = tuple(t.detach() for g in (gate_in, gate_f) for t in g if t is not None)
    wp, b2, gws, gw2, lbs = _cached_pack('gatepair', ptens, (float(plan16['uniform_w']), tuple(t is None for g in (gate_in, gate_f) for t in g)), st, prepare)
    nch = 2 * (F // 32)
    waves = int(lib.gcrnn_fused_step_waves())
    parts = torch.empty((T * B, nch * waves), dtype=torch.float32, device=xs.device)
    cs_in = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=xs.device) if store_states else None
    cs_f = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=xs.device) if store_states else None
    check(lib.gcrnn_fused_gate_pair_prepass_wide_bf16(_p(x_user), _p(xs), _p(h0s), _p(wp), _p(b2), _p(gw2), _p(parts), _p(cs_in), _p(cs_f),
                                                      _p(plan16['tile_slots']), _p(plan16['tile_off']), _p(plan16['ell_col4']), plan16['entries'],
                                                      B, T, N, F, G, K, _p(hzero), _p(plan16.get('rank1_a')), _p(plan16.get('rank1_b')), st), 'gate_pair_prepass')
    if x_user is not None:
        del xs._pending_user
    # read-out finish in ONE launch (round 5; until round 4 five torch launches): partials added in a fixed order, + bias, sigmoid
    out = [torch.empty((T, B), dtype=torch.float32, device=xs.device) for _ in range(2)]
    lbp = [(lb.reshape(-1).contiguous() if lb is not None else None) for lb in lbs]
    check(lib.gcrnn_gate_readout_finish(_p(parts), (nch // 2) * waves, _p(lbp[0]), _p(lbp[1]), _p(out[0]), _p(out[1]), T * B, st), 'gate_readout_finish')
    if store_states:
        return out[0], out[1], (cs_in, gws[0]), (cs_f, gws[1])
    return out[0], out[1]


def fused_cell_forward(X, h0, wA, wB, bias, graph, gates=None, return_states=False, gate_values=None, packed=None,
                       last_only=False, head=None, native_out=False):
    """Whole GGCRNNCell forward (un-gated or time-gated) on the fused bf16 step kernel.

    X: B x T x G x N bf16, h0: B x F x N bf16 (user layout) -> H: B x T x F x N bf16.
    gates: None, or {'in': (wA_g, wB_g, bias_g, lin_weight [1, F*N], lin_bias), 'forget': (...)} -- the time-gate
    sub-networks GFL_* / MLP_* of the reference (graphML.py:2248-2278); each gate costs one pre-pass launch over all
    (t, b). gate_values: the two gates themselves, (gi, gf) [T][B] fp32, instead of their sub-networks. packed: the result
    of fused_pack_inputs (shared with the gates by the training path). No autograd graph is recorded here.
    return_states: also returns the state buffer hs_all [T+1][B][NPad][F] (slot 0 = h0) and the plan.
    last_only: H is B x 1 x F x N, the last state alone (the classification models' read-out); the user-layout store of the
    other steps is skipped.
    head: None, or (weight 1 x F, bias [1] or None) of an output head Linear(F -> 1) shared by all nodes (the regression model's
    `multipMlp` head with one output, architectures.py:1616-1627): it is evaluated in the step kernel's epilogue, H is never written
    in the user layout, and the function returns y: B x T x 1 x N (fp32) instead of H.
    native_out: H is returned as a VIEW of the sequence-major state image the recurrence keeps anyway (hs [T][B][NPad][F]:
    hs.permute(1, 0, 3, 2)[..., :N] has the reference's B x T x F x N shape, reference graphML.py:2425-2427) -- the launches skip the
    user-layout copy of every h_t (a third of the bytes they write); callers that need contiguity call .contiguous() themselves.
    """
    require_device(X, h0, wA, wB, bias)
    if packed is None and X.shape[2] < wA.shape[3]:
        # taps already padded to the kernels' input width, X still with its own G channels (the drivers' G = 1): lay X out for the
        # kernels with the channel padding done by the pack itself -- no 32-channel copy of X in the user layout
        if X.shape[3] % 2 == 0 and X.dtype == torch.bfloat16:
            packed = fused_pack_inputs(X, h0, graph, channels=wA.shape[3])
        else:
            Xz = X.new_zeros((X.shape[0], X.shape[1], wA.shape[3], X.shape[3]))
            Xz[:, :, :X.shape[2]] = X
            X = Xz
    elif packed is None:
        X, wA = fused_pad_operands(X, wA.detach())        # G < 32: zero-padded input channels (no-op when already padded)
    B, T, _, N = X.shape
    G = wA.shape[3]
    F = wA.shape[0]
    K = max(wA.shape[2], wB.shape[2])
    plan = graph.fused_plan()
    st = _stream()
    dev = X.device
    X = X.contiguous()
    h0 = h0.contiguous()
    events = None
    inline = False
    if packed is not None:
        xs, hs_all = packed
    elif gates is None and gate_values is None and X.data_ptr() % 16 == 0 and (
            fused_inline_pack_ok(plan, N, F, G, K) or (head is None and not os.environ.get('GCRNN_NO_INLINE_PACK')
                                                       and fused_wide_plan(graph, B, T, N, F, G, K, True, rank1=True) is not None)):
        # un-gated cell on a uniform-weight graph: only x_0 is packed here, launch t lays out x_{t+1} itself (LDS-DMA into the room
        # the missing weight image leaves, read back transposed after the epilogue) -- no pack pass over X
        xs, hs_all = fused_pack_inputs(X, h0, graph, first_only=True)
        inline = True
    elif gates is None and fused_overlap_ok(X):          # (the gate pre-passes read every x_t at once: nothing to hide behind)
        xs, hs_all, events = fused_pack_inputs(X, h0, graph, overlap=True)
    elif gates is not None and gate_values is None:
        xs, hs_all = fused_pack_inputs_gated(X, h0, graph, F, K)       # (the first gate pre-pass lays out x_1 .. x_{T-1})
    else:
        xs, hs_all = fused_pack_inputs(X, h0, graph)
    h0s, hs = hs_all[:1], hs_all[1:]
    gi = gf = None
    if gate_values is not None:
        gi, gf = (g.detach().float().contiguous() for g in gate_values)
        assert tuple(gi.shape) == (T, B) and tuple(gf.shape) == (T, B)
    elif gates is not None:
        hzero = fused_h0_zero_flag(h0)
        g = {}
        gp = {}
        for name in ('in', 'forget'):
            wA_g, wB_g, bias_g, lin_w, lin_b = gates[name]
            if wA_g.shape[3] != G:
                wA_g = torch.nn.functional.pad(wA_g.detach(), (0, G - wA_g.shape[3]))
            assert max(wA_g.shape[2], wB_g.shape[2]) == K and wA_g.shape[0] == F
            gp[name] = (wA_g, wB_g, bias_g, lin_w, lin_b)
        pair16, _ = fused_gate_pair_plan(graph, B, T, N, F, G, K, getattr(xs, '_pending_user', None) is not None)
        if pair16 is not None:
            gi, gf = fused_time_gate_pair(xs, h0s, gp['in'], gp['forget'], graph, N, hzero=hzero)      # ONE launch for both gates
        else:
            for name in ('in', 'forget'):
                g[name] = fused_time_gate(xs, h0s, *gp[name], graph, N, hzero=hzero)
            gi, gf = g['in'], g['forget']
    assert getattr(xs, '_pending_user', None) is None
    b32 = _bias_f32(bias, st)
    direct = (N % 8 == 0)                 # the step kernels write the user layout themselves (16-byte row stores)
    evs = None
    if events is not None:                               # raw hipEvent_t handles, one slot per step (host array, read during the call)
        evs = (C.c_void_p * T)(*[(e.cuda_event if e is not None else None) for e in events])
    if head is not None:
        assert not return_states and not last_only
        hw = head[0].detach().float().reshape(-1).contiguous()
        assert hw.numel() == F
        part = torch.empty((T, B, F // 16, N), dtype=torch.float32, device=dev)
        plan16 = fused_img16_plan(graph, gi is not None, head)
        wpack = _fused_pack_weights(wA, wB, st)
        check(lib.gcrnn_fused_forward_bf16(_p(xs), _p(h0s), _p(hs), _p(wpack), _p(b32), _p(gi), _p(gf), *_fused_graph_args(plan16 or plan),
                                           B, T, N, F, G, K, None, 2 if plan16 else 0, evs, plan.get('uniform_w', 0.0), _p(X) if inline else None,
                                           _p(hw), _p(part), st), 'fused_forward')
        y = part.sum(dim=2)                                          # fixed order over the F / 16 chunks
        if head[1] is not None:
            y = y + head[1].detach().float().reshape(())
        return y.permute(1, 0, 2).unsqueeze(2).contiguous()          # B x T x 1 x N
    wide = fused_wide_plan(graph, B, T, N, F, G, K, inline, rank1=True, gated=(gi is not None)) if (evs is None and head is None and not (gi is not None and inline)) else None
    if wide is not None:
        # un-gated cell, uniform-weight graph, a batch that fills the chip: ONE launch of the wide sequence-resident kernel
        if native_out:
            assert not return_states
            _fused_forward_wide(wide, xs, h0s, hs, wA, wB, b32, B, T, N, F, G, K, None, False, X if inline else None, st, gi=gi, gf=gf)
            Hv = hs.permute(1, 0, 3, 2)[:, :, :, :N]
            return Hv[:, T - 1:] if last_only else Hv
        H = torch.empty((B, 1 if last_only else T, F, N), dtype=torch.bfloat16, device=dev)
        _fused_forward_wide(wide, xs, h0s, hs, wA, wB, b32, B, T, N, F, G, K, H if direct else None, last_only, X if inline else None, st, gi=gi, gf=gf)
        if not direct:
            src = hs[T - 1:] if last_only else hs
            check(lib.gcrnn_unpack_seq_major(_lib.BF16, _p(src), _p(H), B, 1 if last_only else T, F, N, plan['npad'], None, st), 'unpack_seq')
        if return_states:
            return hs_all, plan, H
        return H
    plan16 = fused_img16_plan(graph, gi is not None, None)
    wpack = _fused_pack_weights(wA, wB, st)
    if native_out:
        assert not return_states
        check(lib.gcrnn_fused_forward_bf16(_p(xs), _p(h0s), _p(hs), _p(wpack), _p(b32), _p(gi), _p(gf), *_fused_graph_args(plan16 or plan),
                                           B, T, N, F, G, K, None, (2 if plan16 else 0), evs,
                                           plan.get('uniform_w', 0.0), _p(X) if inline else None, None, None, st),
              'fused_forward')
        Hv = hs.permute(1, 0, 3, 2)[:, :, :, :N]                     # B x T x F x N, strides of the sequence-major image
        return Hv[:, T - 1:] if last_only else Hv
    H = torch.empty((B, 1 if last_only else T, F, N), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_fused_forward_bf16(_p(xs), _p(h0s), _p(hs), _p(wpack), _p(b32), _p(gi), _p(gf), *_fused_graph_args(plan16 or plan),
                                       B, T, N, F, G, K, _p(H) if direct else None, int(last_only) | (2 if plan16 else 0), evs,
                                       plan.get('uniform_w', 0.0), _p(X) if inline else None, None, None, st),
          'fused_forward')
    if not direct:
        src = hs[T - 1:] if last_only else hs
        check(lib.gcrnn_unpack_seq_major(_lib.BF16, _p(src), _p(H), B, 1 if last_only else T, F, N, plan['npad'], None, st), 'unpack_seq')
    if return_states:
        return hs_all, plan, H
    return H


def fused_cell_forward_native(xs, h0s, wA, wB, bias, graph, N):
    """Un-gated GGCRNNCell forward on sequence-major arrays end to end: xs [T][B][NPad][G] bf16 (rows >= N zero; G = 32 or 64: the
    fused kernels' input widths), h0s [B][NPad][F] bf16 or None (= zeros, every call site of the reference: train_rnn.py:256) ->
    hs [T][B][NPad][F] bf16 (rows >= N zero). No pack, no inline pack, no user-layout copy: the launch moves the algorithm's own
    bytes (read x_t, read h_{t-1}, write h_t). `hs.permute(1, 0, 3, 2)[..., :N]` is the reference's B x T x F x N view of it."""
    require_device(xs, wA, wB, bias)
    T, B, npad, G = xs.shape
    F = wA.shape[0]
    K = max(wA.shape[2], wB.shape[2])
    plan = graph.fused_plan()
    assert npad == plan['npad'] and xs.dtype == torch.bfloat16 and xs.is_contiguous() and wA.shape[3] == G
    st = _stream()
    hs_all = torch.empty((T + 1, B, npad, F), dtype=torch.bfloat16, device=xs.device)
    if h0s is None:
        hs_all[0].zero_()
    else:
        assert tuple(h0s.shape) == (B, npad, F) and h0s.dtype == torch.bfloat16
        hs_all[0].copy_(h0s)
    wpack = _fused_pack_weights(wA.detach(), wB.detach(), st)
    b32 = bias.detach().float().contiguous().view(-1) if bias is not None else None
    wide = fused_wide_plan(graph, B, T, N, F, G, K, False)
    if wide is not None:
        _fused_forward_wide(wide, xs, hs_all[:1], hs_all[1:], wA, wB, b32, B, T, N, F, G, K, None, False, None, st)
        return hs_all[1:]
    plan16 = fused_img16_plan(graph, False, None)
    check(lib.gcrnn_fused_forward_bf16(_p(xs), _p(hs_all[:1]), _p(hs_all[1:]), _p(wpack), _p(b32), None, None, *_fused_graph_args(plan16 or plan),
                                       B, T, N, F, G, K, None, (2 if plan16 else 0), None, plan.get('uniform_w', 0.0), None, None, None, st),
          'fused_forward')
    return hs_all[1:]


def to_sequence_major(X, graph):
    """user layout X [B][T][C][N] bf16 -> the fused kernels' sequence-major array [T][B][NPad][C] (rows >= N zero)."""
    require_device(X)
    B, T, Cc, N = X.shape
    npad = graph.fused_plan()['npad']
    Xc = X.contiguous()
    xs = torch.empty((T, B, npad, Cc), dtype=torch.bfloat16, device=X.device)
    check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(Xc), _p(xs), B, T, Cc, N, npad, None, _stream()), 'pack_seq')
    return xs


# ------------------------------------------------------------------------------------------ node-gated cell on the fused path
def fused_filter_output(xs, w, bias, graph, K, N, adjoint=False):
    """(w(S) x_t + bias) for every (t, b) in one launch: [T][B][NPad][F] bf16 sequence-major (gcrnn_fused_filter_output_bf16).
    xs [T][B][NPad][C] bf16; w F x 1 x k x C (k <= K taps); C == F runs the operand as a state, otherwise as [0 | x_t].
    adjoint: the shifts run on the adjoint graph (S^T): with transposed taps this is the filter's input gradient."""
    T, B, npad, Cin = xs.shape
    F = w.shape[0]
    plan = graph.fused_plan(adjoint=adjoint)
    if adjoint:
        plan16 = None if os.environ.get('GCRNN_NO_IMG16') else graph.fused_plan_img16(adjoint=True)
    else:
        plan16 = fused_img16_plan(graph, False, None)
    st = _stream()
    b32 = _bias_f32(bias, st)
    out = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=xs.device)
    if Cin == F:
        wp = _cached_pack('statetaps_f', (w.detach(),), int(K), st, lambda: _fused_pack_state_taps(w, K, st))
        check(lib.gcrnn_fused_filter_output_bf16(_p(xs), None, _p(wp), _p(b32), _p(out), *_fused_graph_args(plan16 or plan), B, T, N, F, 0, K,
                                                 plan.get('uniform_w', 0.0), 1 if plan16 else 0, st),
              'fused_filter_output')
    else:
        def pack_x():
            wd = w.detach()
            if wd.shape[2] < K:
                wd = torch.cat([wd, wd.new_zeros(F, 1, K - wd.shape[2], Cin)], dim=2)
            return _fused_pack_weights(wd, wd.new_zeros((F, 1, K, F)), st), torch.zeros((1, npad, F), dtype=torch.bfloat16, device=xs.device)
        wp, zero_h = _cached_pack('filtertaps_x', (w.detach(),), (int(K), int(npad)), st, pack_x)
        check(lib.gcrnn_fused_filter_output_bf16(_p(zero_h), _p(xs), _p(wp), _p(b32), _p(out), *_fused_graph_args(plan16 or plan), B, T, N, F, Cin, K,
                                                 plan.get('uniform_w', 0.0), 1 if plan16 else 0, st),
              'fused_filter_output')
    return out


def node_gate_logits(cs, wf, bf, graph, N):
    """GraphFilter_{F -> 1} of the gate-cell states (graphML.py:2387), taps first: s_k = cs . w_k per node (one pass over cs), then
    K-1 accumulate-SpMM hops on the ONE-channel signals in node-major layout [N][items]. cs [T][B][NPad][F] bf16, wf 1 x 1 x K x F,
    bf 1 x 1 or None -> logits [T][B][N] fp32 (and the pieces the backward needs)."""
    T, B, npad, F = cs.shape
    K = wf.shape[2]
    items = T * B
    wk = wf.detach().float().reshape(K, F).contiguous()
    s = torch.empty((items, K, 1, N), dtype=torch.float32, device=cs.device)
    for i0 in range(0, items, 32768):            # gridDim.y limit of the dot kernel
        n_i = min(32768, items - i0)
        check(lib.gcrnn_node_gate_dot(_p(cs.view(items, npad, F)[i0:]), _p(wk), _p(s[i0:]), n_i, N, npad, F, K, _stream()), 'node_gate_dot')
    return _node_gate_logits_from_taps(s, bf, graph, T, B, N), wk


def _node_gate_logits_from_taps(s, bf, graph, T, B, N):
    """Second stage of the F -> 1 GraphFilter: s [items][K][1][N] fp32 per-tap dot products -> logits [T][B][N] (K-1 Horner hops on
    the one-channel signals, node-major). s [items][S][K][1][N]: per-slice partials, added by the layout pass (gcrnn_pack_node_major_sum_f32)."""
    if s.dim() == 5:
        items, S_, K = s.shape[0], s.shape[1], s.shape[2]
        sn = torch.empty((K, N, items, 1), dtype=torch.float32, device=s.device)
        check(lib.gcrnn_pack_node_major_sum_f32(_p(s), _p(sn), items, S_, K, 1, N, _stream()), 'pack_node_major_sum')
    else:
        K = s.shape[1]
        sn = _pack_raw(s)                                              # [K][N][items][1]
    acc = sn[K - 1:K]
    csr = graph.fwd[0]
    for k in range(K - 2, -1, -1):
        spmm_raw(csr, acc, out=sn[k:k + 1], accumulate=True)         # sn[k] += P acc  (Horner)
        acc = sn[k:k + 1]
    logit = _unpack_raw(acc).view(T, B, N)                             # [items][1][1][N]
    if bf is not None:
        logit = logit + bf.detach().float().view(())
    return logit


def _tap_fragments(wf, F):
    """The F -> 1 filter's taps wf 1 x 1 x K x F (fp32) as the A fragments gcrnn_fused_gate_prepass_taps_bf16 takes: three bf16 planes
    (p0 + p1 + p2 = w to 24 bits), bf16 [F/16][3][64][4] with lane l = 16 kg + tap holding w_p[tap][16 chunk + 4 kg + e]."""
    K = wf.shape[2]
    w = wf.detach().float().reshape(K, F)
    p0 = w.to(torch.bfloat16)
    r1 = w - p0.float()
    p1 = r1.to(torch.bfloat16)
    p2 = (r1 - p1.float()).to(torch.bfloat16)
    pl = torch.zeros((3, 16, F), dtype=torch.bfloat16, device=w.device)
    pl[:, :K] = torch.stack([p0, p1, p2])
    # [plane][tap][chunk][kg][e] -> [chunk][plane][kg][tap][e]
    return pl.view(3, 16, F // 16, 4, 4).permute(2, 0, 3, 1, 4).contiguous()


def fused_node_gate_taps(xs, h0s, wA_g, wB_g, bias_g, wf, graph, N, hzero=None):
    """Gate cell of a node gate at inference with the per-tap dot products of its F -> 1 filter fused into the pre-pass
    (gcrnn_fused_gate_prepass_taps_bf16): returns s [T*B][K][1][N] fp32, or None where the fused pre-pass does not apply (the caller
    then stores the states and runs gcrnn_node_gate_dot over them). Lays out X too when xs carries a pending user-layout tensor."""
    T, B, npad, G = xs.shape
    F = wA_g.shape[0]
    K = max(wA_g.shape[2], wB_g.shape[2])
    Kt = wf.shape[2]
    plan = graph.fused_plan()
    plan16 = fused_img16_plan(graph, True, None)
    x_user = None
    if getattr(xs, '_pending_user', None) is not None:
        lays = int(lib.gcrnn_fused_gate_prepass_lays_out(B, T, N, F, G, K, plan16['entries'], plan.get('uniform_w', 0.0), 1)) if plan16 is not None else 0
        x_user = _pending_layout_for(xs, lays)
    if plan16 is None or os.environ.get('GCRNN_NO_FUSED_TAPS') or not int(lib.gcrnn_fused_gate_prepass_taps_supported(
            B, T, N, F, G, K, plan16['entries'], plan.get('uniform_w', 0.0), 1, 1 if x_user is not None else 0, Kt)):
        return None
    st = _stream()
    wp = _fused_pack_weights(wA_g.detach(), wB_g.detach(), st)
    bg = _bias_f32(bias_g, st)
    frags = _cached_pack('tapfrags', (wf.detach(),), int(F), st, lambda: _tap_fragments(wf, F))      # (parameter-derived: kept while wf is unchanged)
    s = torch.empty((T * B, Kt, 1, N), dtype=torch.float32, device=xs.device)
    check(lib.gcrnn_fused_gate_prepass_taps_bf16(_p(x_user), _p(xs), _p(h0s), _p(wp), _p(bg), _p(frags), _p(s), Kt, None,
                                                 *_fused_graph_args(plan16), B, T, N, F, G, K, _p(hzero), plan.get('uniform_w', 0.0), 1, st),
          'gate_prepass_taps')
    if x_user is not None:
        del xs._pending_user
    return s


def _tap_fragments32(wfs, F):
    """The F -> 1 filters' taps of BOTH node gates (each 1 x 1 x K x F) as the A fragments gcrnn_fused_gate_pair_prepass_taps_wide_bf16 takes:
    three bf16 planes (p0 + p1 + p2 = w to 24 bits), bf16 [2][F/32][3][64][8] with lane l = 16 kg + tap holding w_p[tap][32 cg + 8 kg .. + 7]."""
    out = []
    for wf in wfs:
        K = wf.shape[2]
        w = wf.detach().float().reshape(K, F)
        p0 = w.to(torch.bfloat16)
        r1 = w - p0.float()
        p1 = r1.to(torch.bfloat16)
        p2 = (r1 - p1.float()).to(torch.bfloat16)
        pl = torch.zeros((3, 16, F), dtype=torch.bfloat16, device=w.device)
        pl[:, :K] = torch.stack([p0, p1, p2])
        # [plane][tap][cg][kg][j] -> [cg][plane][kg][tap][j]
        out.append(pl.view(3, 16, F // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous())
    return torch.stack(out, dim=0).contiguous()


def fused_node_gate_taps_pair(xs, h0s, gate_in, gate_f, graph, N, hzero=None):
    """BOTH node gates' cells of every (t, b) with the per-tap dot products of their F -> 1 filters, as ONE pre-pass launch of the wide
    sequence-resident kernel (gcrnn_fused_gate_pair_prepass_taps_wide_bf16; inference). gate_* = (wA_g, wB_g, bias_g, wf, bf) as in
    fused_node_cell_forward. Returns (s_in, s_f), each [T*B][K][1][N] fp32, or None where the wide pre-pass does not apply."""
    T, B, npad, G = xs.shape
    F = gate_in[0].shape[0]
    K = max(gate_in[0].shape[2], gate_in[1].shape[2])
    Kt = gate_in[3].shape[2]
    if gate_f[3].shape[2] != Kt or max(gate_f[0].shape[2], gate_f[1].shape[2]) != K or os.environ.get('GCRNN_NO_NODE_GATE_PAIR'):
        return None
    if fused_gate_pair_plan(graph, B, T, N, F, G, K, getattr(xs, '_pending_user', None) is not None)[0] is None:
        return None
    x_user = None
    if getattr(xs, '_pending_user', None) is not None:
        x_user = _pending_layout_for(xs, fused_gate_pair_plan(graph, B, T, N, F, G, K, True)[1])
    plan16, _ = fused_gate_pair_plan(graph, B, T, N, F, G, K, x_user is not None)
    if plan16 is None:
        return None
    st = _stream()

    def prepare():
        padx = lambda w: w.detach() if w.shape[3] == G else torch.nn.functional.pad(w.detach(), (0, G - w.shape[3]))      # (G < 32: zero input taps for the padded channels)
        wA2 = torch.cat([padx(gate_in[0]), padx(gate_f[0])], dim=0)
        wB2 = torch.cat([gate_in[1].detach(), gate_f[1].detach()], dim=0)
        wp_ = _fused_pack_weights_wide(wA2, wB2, plan16['uniform_w'], st)
        zb = torch.zeros(F, dtype=torch.float32, device=xs.device)
        b2_ = torch.cat([(g[2].detach().float().reshape(-1) if g[2] is not None else zb) for g in (gate_in, gate_f)]).contiguous()
        return wp_, b2_, _tap_fragments32((gate_in[3], gate_f[3]), F)
    ptens = tuple(t.detach() for g in (gate_in, gate_f) for t in g[:4] if t is not None)
    wp, b2, frags = _cached_pack('nodegatepair', ptens, (float(plan16['uniform_w']), tuple(t is None for g in (gate_in, gate_f) for t in g[:4])), st, prepare)
    nch = F // 32
    parts = torch.empty((T * B, 2, nch, Kt, N), dtype=torch.float32, device=xs.device)
    check(lib.gcrnn_fused_gate_pair_prepass_taps_wide_bf16(_p(x_user), _p(xs), _p(h0s), _p(wp), _p(b2), _p(frags), _p(parts), Kt, None, None,
                                                           _p(plan16['tile_slots']), _p(plan16['tile_off']), _p(plan16['ell_col4']), plan16['entries'],
                                                           B, T, N, F, G, K, _p(hzero), _p(plan16.get('rank1_a')), _p(plan16.get('rank1_b')), st),
          'gate_pair_prepass_taps')
    if x_user is not None:
        del xs._pending_user
    return parts.view(T * B * 2, nch, Kt, 1, N)                         # row 2 i + g: item i, gate g; [.., chunk, tap, 1, node]: the caller's layout pass adds the chunks


def fused_node_supported(graph, N, F, G, Kin, Kst, dtype, E=1):
    """Node-gated cell on the fused kernels: the fused shapes in bf16 plus a state-only instantiation for K = max(Kin, Kst)."""
    return fused_supported(N, F, G, Kin, Kst, dtype, E) and max(Kin, Kst) in (2, 3, 4, 5) and not (max(Kin, Kst) == 4 and F == 32)


def fused_node_cell_forward(X, h0, wA, wB, bias, graph, node_gates, time_gates=None, last_only=False):
    """Node-gated GGCRNNCell forward (optionally time-gated too) on the fused kernels (graphML.py:2379-2407, 2420-2423).
    node_gates = {'in': (wA_g, wB_g, bias_g, wf, bf), 'forget': (...)}: the gate cell GRNN_node_* and its F -> 1 GraphFilter
    GFL_node_*; time_gates as in fused_cell_forward. X B x T x G x N bf16, h0 B x F x N bf16 -> H B x T x F x N bf16.
    Everything that does not depend on h_{t-1} -- both gate cells, their filters, A(S)x_t + b -- runs for all T steps at once.
    Inference (no autograd graph is recorded here; training goes through fused_node_cell_train)."""
    require_device(X, h0, wA, wB, bias)
    X, wA = fused_pad_operands(X, wA.detach())
    B, T, G, N = X.shape
    F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
    K = max(Kin, Kst)
    plan = graph.fused_plan()
    st = _stream()
    xs, hs_all = fused_pack_inputs_gated(X.contiguous(), h0.contiguous(), graph, F, K)      # (the first gate cell's pre-pass lays out the rest of X)
    h0s, hs = hs_all[:1], hs_all[1:]
    hzero = fused_h0_zero_flag(h0)
    zero_lin = torch.zeros((1, F * N), dtype=torch.float32, device=X.device)
    ng = []
    pair = fused_node_gate_taps_pair(xs, h0s, node_gates['in'], node_gates['forget'], graph, N, hzero=hzero)      # both gate cells in ONE launch (wide kernel)
    ngates = None
    if pair is not None and not os.environ.get('GCRNN_NO_NODE_GATE_FILTER') and int(lib.gcrnn_node_gate_filter_supported(pair.shape[2], N, graph.fwd[0].nnz, plan.get('uniform_w', 0.0))):
        # second stage of both F -> 1 filters in ONE pass over the tap dots: chunk sum, the K - 1 one-channel hops (running signal in LDS),
        # each gate's bias, the sigmoid, written as the recurrence reads them (gcrnn_node_gate_filter_f32)
        bfs = tuple(node_gates[name][4] for name in ('in', 'forget'))
        b2 = None
        if any(b is not None for b in bfs):
            b2 = _cached_pack('nodegatebias', tuple(b.detach() for b in bfs if b is not None), tuple(b is None for b in bfs), st,
                              lambda: torch.cat([(b.detach().float().reshape(1) if b is not None else torch.zeros(1, device=X.device)) for b in bfs]).contiguous())
        csr = graph.fwd[0]
        ngates = torch.empty((T, 2, B, N), dtype=torch.float32, device=X.device)
        check(lib.gcrnn_node_gate_filter_f32(_p(pair), _p(ngates), T * B * 2, pair.shape[1], pair.shape[2], N, 2, B, _p(csr.rowptr), _p(csr.col),
                                             _p(csr.val(torch.float32)), csr.nnz, plan.get('uniform_w', 0.0), _p(b2), 1, st), 'node_gate_filter')
    elif pair is not None:
        lg = _node_gate_logits_from_taps(pair, None, graph, T, B * 2, N).view(T, B, 2, N)
        for gidx, name in enumerate(('in', 'forget')):
            bf = node_gates[name][4]
            l1 = lg[:, :, gidx]
            ng.append(torch.sigmoid(l1 + bf.detach().float().view(()) if bf is not None else l1))
    for name in (() if pair is not None else ('in', 'forget')):
        wA_g, wB_g, bias_g, wf, bf = node_gates[name]
        if wA_g.shape[3] != G:
            wA_g = torch.nn.functional.pad(wA_g.detach(), (0, G - wA_g.shape[3]))
        staps = fused_node_gate_taps(xs, h0s, wA_g, wB_g, bias_g, wf, graph, N, hzero=hzero)
        if staps is not None:          # the filter's tap dots came out of the pre-pass itself: no state array, no pass over it
            logit = _node_gate_logits_from_taps(staps, bf, graph, T, B, N)
        else:
            _, cs, _ = fused_time_gate(xs, h0s, wA_g, wB_g, bias_g, zero_lin, None, graph, N, store_states=True, hzero=hzero)
            logit, _ = node_gate_logits(cs, wf, bf, graph, N)
        ng.append(torch.sigmoid(logit))
    if ngates is None:
        ngates = torch.stack(ng, dim=1).contiguous()                    # [T][2][B][N]
    assert getattr(xs, '_pending_user', None) is None
    gi = gf = None
    if time_gates is not None:
        g = {}
        for name in ('in', 'forget'):
            wA_g, wB_g, bias_g, lin_w, lin_b = time_gates[name]
            if wA_g.shape[3] != G:
                wA_g = torch.nn.functional.pad(wA_g.detach(), (0, G - wA_g.shape[3]))
            g[name] = fused_time_gate(xs, h0s, wA_g, wB_g, bias_g, lin_w, lin_b, graph, N, hzero=hzero)
        gi, gf = g['in'], g['forget']
    plan16 = fused_img16_plan(graph, False, None)
    uw = float(plan.get('uniform_w', 0.0))
    b32 = _bias_f32(bias, st)
    # (round 5) both state-size passes on the WIDE kernel when it takes them: A(S) x_t + b over all items (mode 3) and the recurrence with the
    # per-node gates in its epilogue as ONE launch (mode 4); else round 3's 16-feature kernels
    if plan16 is not None and F % 32 == 0 and G % 32 == 0 and lib.gcrnn_fused_filter_output_wide_supported(B, T, N, F, G, K, int(plan16['entries']), uw, 1, 0):
        wpx = _fused_pack_weights_wide(wA.detach(), wA.new_zeros((F, 1, K, F)), uw, st)      # (zero state taps, K of them: the pack's tap count is the kernel's)
        yx = torch.empty((T, B, plan['npad'], F), dtype=torch.bfloat16, device=X.device)
        check(lib.gcrnn_fused_filter_output_wide_bf16(_p(xs), _p(wpx), _p(b32), _p(yx), _p(plan16['tile_slots']), _p(plan16['tile_off']),
                                                      _p(plan16['ell_col4']), plan16['entries'], B, T, N, F, G, K, None, st), 'fused_filter_output_wide')
    else:
        yx = fused_filter_output(xs, wA, bias, graph, K, N)
    H = torch.empty((B, 1 if last_only else T, F, N), dtype=torch.bfloat16, device=X.device)
    direct = (N % 8 == 0)
    if plan16 is not None and F % 32 == 0 and lib.gcrnn_fused_node_forward_wide_supported(B, T, N, F, K, int(plan16['entries']), uw, 1):
        wBk = wB.detach() if Kst == K else torch.cat([wB.detach(), wB.new_zeros(F, 1, K - Kst, F)], dim=2)
        wpBw = _fused_pack_weights_wide(wB.new_zeros((F, 1, 1, 0)), wBk, uw, st)      # (state-only operand: G = 0)
        check(lib.gcrnn_fused_node_forward_wide_bf16(_p(h0s), _p(hs), _p(yx), _p(ngates), _p(gi), _p(gf), _p(wpBw), _p(b32), _p(plan16['tile_slots']),
                                                     _p(plan16['tile_off']), _p(plan16['ell_col4']), plan16['entries'], B, T, N, F, K,
                                                     _p(H) if direct else None, int(last_only), st), 'fused_node_forward_wide')
        if not direct:
            src = hs[T - 1:] if last_only else hs
            check(lib.gcrnn_unpack_seq_major(_lib.BF16, _p(src), _p(H), B, 1 if last_only else T, F, N, plan['npad'], None, st), 'unpack_seq')
        return H
    def pack_state():
        wBk = wB.detach() if Kst == K else torch.cat([wB.detach(), wB.new_zeros(F, 1, K - Kst, F)], dim=2)
        return _fused_pack_state_taps(wBk, K, st)
    wpB = _cached_pack('statetaps', (wB.detach(),), int(K), st, pack_state)
    check(lib.gcrnn_fused_node_forward_bf16(_p(h0s), _p(hs), _p(yx), _p(ngates), _p(gi), _p(gf), _p(wpB), _p(b32), None,
                                            *_fused_graph_args(plan16 or plan), B, T, N, F, K, _p(H) if direct else None,
                                            int(last_only) | (2 if plan16 else 0), plan.get('uniform_w', 0.0), st),
          'fused_node_forward')
    if not direct:
        src = hs[T - 1:] if last_only else hs
        check(lib.gcrnn_unpack_seq_major(_lib.BF16, _p(src), _p(H), B, 1 if last_only else T, F, N, plan['npad'], None, st), 'unpack_seq')
    return H


class _FusedNodeGate(torch.autograd.Function):
    """One node gate of the fused path with its BPTT (graphML.py:2379-2399): gate[t][b][n] = sigmoid(GraphFilter_{F->1}(c_t))[n],
    c_t = tanh(A_g(S)x_t + B_g(S)h0 + 2 b_g) the state of the gate cell GRNN_node_*.
    forward  = the gate pre-pass over all (t, b) storing c, the taps-first F -> 1 filter (node_gate_logits), sigmoid;
    backward = adjoint Horner of the one-channel hops, one pass over c (filter-weight gradient + the gate cell's pre-activation
               gradient in place), then the weight-gradient kernel with h0 as every item's state operand. No gradient for X / h0."""

    @staticmethod
    def forward(ctx, xs, h0s, X, h0, wA_g, wB_g, bias_g, wf, bf, graph, hzero):
        N, F = X.shape[3], wA_g.shape[0]
        zero_lin = torch.zeros((1, F * N), dtype=torch.float32, device=X.device)
        _, cs, _ = fused_time_gate(xs, h0s, wA_g, wB_g, bias_g, zero_lin, None, graph, N, store_states=True, hzero=hzero)
        logit, wk = node_gate_logits(cs, wf, bf, graph, N)
        gate = torch.sigmoid(logit)
        ctx.save_for_backward(X, h0, wA_g, wB_g, bias_g, wf, bf, gate, cs, wk, hzero)
        ctx.graph = graph
        return gate

    @staticmethod
    def backward(ctx, dgate):
        X, h0, wA_g, wB_g, bias_g, wf, bf, gate, cs, wk, hzero = ctx.saved_tensors
        if getattr(ctx, 'consumed', False):
            raise GcrnnError('the fused node gate was already back-propagated (its saved states are turned into gradients in place)')
        ctx.consumed = True
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            raise GcrnnError('the fused node gate does not produce gradients w.r.t. X or h0')
        graph = ctx.graph
        B, T, G, N = X.shape
        F, Kin, Kst = wA_g.shape[0], wA_g.shape[2], wB_g.shape[2]
        Kc = max(Kin, Kst)
        K = wf.shape[2]
        npad = cs.shape[2]
        items = T * B
        dlogit = (dgate.float() * gate * (1.0 - gate)).contiguous()                     # through the sigmoid: [T][B][N]
        # logit = sum_k P^k s_k + b  =>  d s_k = (P^T)^k dlogit: K-1 one-channel hops with the adjoint graph (CSR(S))
        dn = torch.empty((K, N, items, 1), dtype=torch.float32, device=X.device)
        dn[0:1] = _pack_raw(dlogit.view(items, 1, 1, N))
        for k in range(1, K):
            spmm_raw(graph.adj[0], dn[k - 1:k], out=dn[k:k + 1])
        ds = _unpack_raw(dn)                                                              # [items][K][1][N]
        dw_part = torch.empty((items, K, F), dtype=torch.float32, device=X.device)
        check(lib.gcrnn_node_gate_dot_backward(_p(cs), _p(ds), _p(wk), _p(dw_part), items, N, npad, F, K, _stream()), 'node_gate_dot_backward')
        gwf = dw_part.sum(dim=0).view(1, 1, K, F).to(wf.dtype) if ctx.needs_input_grad[7] else None
        gbf = dlogit.sum().view_as(bf).to(bf.dtype) if (bf is not None and ctx.needs_input_grad[8]) else None
        # cs now holds the gate cell's pre-activation gradient
        dW, dbs = fused_backward_weight(cs, X, None, h0, graph, F, G, Kc, want_bias=True, h_is_h0=True, hzero=hzero)
        gA = dW[:, :Kin, F:].unsqueeze(1).to(wA_g.dtype) if ctx.needs_input_grad[4] else None
        gB = dW[:, :Kst, :F].unsqueeze(1).to(wB_g.dtype) if ctx.needs_input_grad[5] else None
        gb = dbs.view_as(bias_g).to(bias_g.dtype) if (bias_g is not None and ctx.needs_input_grad[6]) else None
        return None, None, None, None, gA, gB, gb, gwf, gbf, None, None


class _FusedNodeCell(torch.autograd.Function):
    """Node-gated GGCRNNCell (optionally time-gated too) on the fused kernels with its BPTT; the gates enter as differentiable
    inputs ni, nf [T][B][N] (and gi, gf [T][B] or None). backward = data chain with the per-node forget gates folded into every
    launch's operand -> one all-items pass for the gate gradients and the x filter's pre-activation gradient -> two weight-gradient
    launches (input filter on gi ni . dpre, state filter on gf nf . dpre). No gradient for X or h0."""

    @staticmethod
    def forward(ctx, X, h0, wA, wB, bias, ni, nf, gi, gf, graph, xs, hs_all):
        B, T, G, N = X.shape
        F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
        K = max(Kin, Kst)
        plan = graph.fused_plan()
        st = _stream()
        h0s, hs = hs_all[:1], hs_all[1:]
        yx = fused_filter_output(xs, wA, bias, graph, K, N)
        ngates = torch.stack([ni.detach().float(), nf.detach().float()], dim=1).contiguous()
        gic = gi.detach().float().contiguous() if gi is not None else None
        gfc = gf.detach().float().contiguous() if gf is not None else None
        wBk = wB.detach() if Kst == K else torch.cat([wB.detach(), wB.new_zeros(F, 1, K - Kst, F)], dim=2)
        wpB = _fused_pack_state_taps(wBk, K, st)
        b32 = bias.detach().float().contiguous().view(-1) if bias is not None else None
        H = torch.empty((B, T, F, N), dtype=torch.bfloat16, device=X.device)
        yh = torch.empty_like(yx)
        plan16 = fused_img16_plan(graph, False, None)
        check(lib.gcrnn_fused_node_forward_bf16(_p(h0s), _p(hs), _p(yx), _p(ngates), _p(gic), _p(gfc), _p(wpB), _p(b32), _p(yh),
                                                *_fused_graph_args(plan16 or plan), B, T, N, F, K, _p(H), 2 if plan16 else 0,
                                                plan.get('uniform_w', 0.0), st), 'fused_node_forward')
        ctx.save_for_backward(X, h0, wA, wB, bias, H, hs_all, yx, yh, ngates, gic, gfc)
        ctx.graph, ctx.npad = graph, plan['npad']
        return H

    @staticmethod
    def backward(ctx, dH):
        X, h0, wA, wB, bias, H, hs_all, yx, yh, ngates, gi, gf = ctx.saved_tensors
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise GcrnnError('the fused node-gated BPTT does not produce gradients w.r.t. X or h0')
        graph, npad = ctx.graph, ctx.npad
        B, T, G, N = X.shape
        F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
        K = max(Kin, Kst)
        st = _stream()
        hs = hs_all[1:]
        dH = dH.to(torch.bfloat16).contiguous()
        dHs, dHu = fused_pack_upstream(dH, graph, K)
        ngf = (ngates[:, 1] * gf.unsqueeze(2)).contiguous() if gf is not None else ngates[:, 1].contiguous()       # [T][B][N]
        wBk = wB.detach() if Kst == K else torch.cat([wB.detach(), wB.new_zeros(F, 1, K - Kst, F)], dim=2)
        wBt = wBk[:, 0].permute(2, 1, 0).contiguous().unsqueeze(1)                      # transposed taps [F_in][1][K][F_out]
        wpT = _fused_pack_state_taps(wBt, K, st)
        aplan = graph.fused_plan(adjoint=True)
        aplan16 = None if os.environ.get('GCRNN_NO_IMG16') else graph.fused_plan_img16(adjoint=True)
        dpre = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=X.device)
        dyh = torch.empty_like(dpre)
        check(lib.gcrnn_fused_node_backward_data_bf16(_p(dHs), _p(hs), _p(dpre), _p(dyh), _p(ngf), _p(wpT), *_fused_graph_args(aplan16 or aplan),
                                                      B, T, N, F, K, aplan.get('uniform_w', 0.0), _p(dHu), 1 if aplan16 else 0, st), 'fused_node_backward_data')
        dyx = torch.empty_like(dpre)
        dni = torch.empty((T, B, N), dtype=torch.float32, device=X.device)
        dnf = torch.empty_like(dni)
        dgi = torch.empty((T, B), dtype=torch.float32, device=X.device) if gi is not None else None
        dgf = torch.empty((T, B), dtype=torch.float32, device=X.device) if gf is not None else None
        check(lib.gcrnn_node_cell_backward(_p(dpre), _p(yx), _p(yh), _p(ngates), _p(gi), _p(gf), _p(dyx), _p(dni), _p(dnf), _p(dgi), _p(dgf),
                                           B, T, N, npad, F, st), 'node_cell_backward')
        one = torch.ones((T, B), dtype=torch.float32, device=X.device)
        zero = torch.zeros_like(one)
        dWx, dbx = fused_backward_weight(dyx, X, H, h0, graph, F, G, K, want_bias=True, gi=one, gf=zero)      # input-filter columns
        dWh, dbh = fused_backward_weight(dyh, X, H, h0, graph, F, G, K, want_bias=True, gi=zero, gf=one)      # state-filter columns
        gA = dWx[:, :Kin, F:].unsqueeze(1).to(wA.dtype) if ctx.needs_input_grad[2] else None
        gB = dWh[:, :Kst, :F].unsqueeze(1).to(wB.dtype) if ctx.needs_input_grad[3] else None
        gb = (dbx + dbh).view_as(bias).to(bias.dtype) if (bias is not None and ctx.needs_input_grad[4]) else None
        return (None, None, gA, gB, gb, dni if ctx.needs_input_grad[5] else None, dnf if ctx.needs_input_grad[6] else None,
                dgi if (gi is not None and ctx.needs_input_grad[7]) else None, dgf if (gf is not None and ctx.needs_input_grad[8]) else None,
                None, None, None)


def fused_node_cell_train(X, h0, wA, wB, bias, graph, node_gates, time_gates=None):
    """Training forward of the node-gated (optionally time + node gated) cell on the fused kernels: every gate sub-network is an
    autograd node of its own (_FusedNodeGate / _FusedTimeGate) and enters the cell (_FusedNodeCell) as a differentiable input."""
    require_device(X, h0, wA, wB, bias)
    with torch.no_grad():
        xs, hs_all = fused_pack_inputs_gated(X.contiguous(), h0, graph, wA.shape[0], max(wA.shape[2], wB.shape[2]))
        hzero = fused_h0_zero_flag(h0)
    ni = _FusedNodeGate.apply(xs, hs_all[:1], X, h0, *node_gates['in'], graph, hzero)
    nf = _FusedNodeGate.apply(xs, hs_all[:1], X, h0, *node_gates['forget'], graph, hzero)
    assert getattr(xs, '_pending_user', None) is None          # (the first pre-pass laid out the rest of X)
    gi = gf = None
    if time_gates is not None:
        gi, gf = fused_train_time_gates(xs, hs_all[:1], X, h0, time_gates, graph, hzero)
    return _FusedNodeCell.apply(X, h0, wA, wB, bias, ni, nf, gi, gf, graph, xs, hs_all)


# ------------------------------------------------------------------------------------------ edge-gated cell, fused path
def fused_edge_supported(graph, N, F, G, Kin, Kst, dtype, E=1):
    """Edge-gated cell on the fused kernels: the node-gated path's shapes (state-only instantiation of the step kernel for the
    filter passes) plus the attention kernel's (N % 8 == 0, z [N][F] bf16 and the per-node scalars in LDS)."""
    return fused_node_supported(graph, N, F, G, Kin, Kst, dtype, E) and bool(lib.gcrnn_fused_edge_attention_supported(int(N), int(F)))


def _edge_composite(w, bias, att_w):
    """The attention's mixing matrix folded into a filter: z = W (sum_k S^k u C_k + b) = sum_k S^k u (W C_k) + W b
    (W acts on features, S on nodes). w F x 1 x k x C, bias F x 1 or None, att_w 1 x 1 x F x F -> (W C) F x 1 x k x C fp32, W b [F]."""
    W = att_w[0, 0].float()
    wc = torch.einsum('pf,fekc->pekc', W, w.float())
    bc = (W @ bias.float().view(-1)) if bias is not None else None
    return wc, bc


def fused_edge_attention(z, a12, graph, gx=None, gi=None, gf=None, out=None, r_out=None, Huser=None, huser_item_stride=0, N=None,
                         negative_slope=0.2):
    """The attention of the fused edge gate (gcrnn_fused_edge_attention_bf16) on the items of z [..][NPad][F] bf16 (leading dims =
    items). gx None: relu(att(z)); else tanh(gi gx + gf relu(att(z))) with per-item scalars gi / gf (fp32, or None)."""
    npad, F = z.shape[-2], z.shape[-1]
    items = z.numel() // (npad * F)
    ep = graph.edge_plan()
    if out is None:
        out = torch.empty_like(z)
    check(lib.gcrnn_fused_edge_attention_bf16(_p(z), _p(a12), _p(gx), _p(gi), _p(gf), _p(ep['rowptr']), _p(ep['r_edge']),
                                              _p(ep['t_rowptr']), _p(ep['t_edge']), _p(ep['t_order']), _p(out), _p(r_out), _p(Huser), int(huser_item_stride),
                                              items, int(N if N is not None else graph.N), npad, F, float(negative_slope), _stream()),
          'fused_edge_attention')
    return out


def fused_edge_cell_forward(X, h0, wA, wB, bias, graph, att_in, att_f, time_gates=None, last_only=False, negative_slope=0.2):
    """Edge-gated GGCRNNCell forward (optionally time-gated too) on the fused kernels (graphML.py:2409-2416, 2420-2423).
    att_in / att_f = (mixer 1 x 1 x 2F, weight 1 x 1 x F x F) of input_attention / forget_attention; time_gates as in
    fused_cell_forward. X B x T x G x N bf16, h0 B x F x N bf16 -> H B x T x F x N bf16 (B x 1 x F x N with last_only).
    The x branch relu(att_in(A(S)x_t + b)) does not depend on the recurrence: one filter pass and one attention pass over all
    T * B items; every step is then two launches: the state filter pass (composite taps) and the attention + tanh epilogue.
    Inference only (training goes through fused_edge_cell_train)."""
    require_device(X, h0, wA, wB, bias)
    X, wA = fused_pad_operands(X, wA.detach())
    B, T, G, N = X.shape
    F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
    K = max(Kin, Kst)
    plan = graph.fused_plan()
    npad = plan['npad']
    st = _stream()
    x_pending = None
    if time_gates is not None:      # (the first time-gate pre-pass lays out the rest of X)
        xs, hs_all = fused_pack_inputs_gated(X.contiguous(), h0.contiguous(), graph, F, K)
    else:
        # (round 5) without time gates the first consumer of xs is the x branch's filter pass: when the wide kernel takes it, only the leading
        # time step(s) are laid out here and its items lay out the rest while they run -- no separate pass over X (0.5 ms at B = 256, T = 32)
        Xc = X.contiguous()
        p16_ = fused_img16_plan(graph, True, None)
        steps = 0
        if (p16_ is not None and F % 32 == 0 and G % 32 == 0 and Xc.dtype == torch.bfloat16 and Xc.data_ptr() % 16 == 0
                and not os.environ.get('GCRNN_NO_INLINE_PACK')):
            steps = int(lib.gcrnn_fused_filter_output_wide_supported(B, T, N, F, G, K, int(p16_['entries']), float(plan.get('uniform_w', 0.0)), 1, 1))
        if 0 < steps < T:
            st0 = _stream()
            xs = torch.empty((T, B, npad, G), dtype=torch.bfloat16, device=X.device)
            hs_all = torch.empty((T + 1, B, npad, F), dtype=torch.bfloat16, device=X.device)
            check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(h0.contiguous()), _p(hs_all), B, 1, F, N, npad, None, st0), 'pack_seq')
            check(lib.gcrnn_pack_seq_major_steps(_p(Xc), _p(xs), B, T, G, N, npad, 0, steps, 0, st0), 'pack_seq_steps')
            x_pending = Xc
        else:
            xs, hs_all = fused_pack_inputs(Xc, h0.contiguous(), graph)
    gi = gf = None
    if time_gates is not None:
        hzero = fused_h0_zero_flag(h0)
        g = {}
        for name in ('in', 'forget'):
            wA_g, wB_g, bias_g, lin_w, lin_b = time_gates[name]
            if wA_g.shape[3] != G:
                wA_g = torch.nn.functional.pad(wA_g.detach(), (0, G - wA_g.shape[3]))
            g[name] = fused_time_gate(xs, hs_all[:1], wA_g, wB_g, bias_g, lin_w, lin_b, graph, N, hzero=hzero)
        gi, gf = g['in'].contiguous(), g['forget'].contiguous()
    wAc, bA = _edge_composite(wA.detach(), bias.detach() if bias is not None else None, att_in[1].detach())
    wBc, bB = _edge_composite(wB.detach(), bias.detach() if bias is not None else None, att_f[1].detach())
    a_in = att_in[0].detach().float().reshape(2, F).contiguous()
    a_f = att_f[0].detach().float().reshape(2, F).contiguous()
    plan16 = fused_img16_plan(graph, False, None)
    uw = float(plan.get('uniform_w', 0.0))
    # (round 5) both filter passes on the WIDE kernel's filter-output mode when it takes them (csrc/gcrnn_fused_seq32.h mode 3: 32-feature chunks,
    # one workgroup per item): the x branch over all T B items, and per step the state filter with h_{t-1} as the mode's "input" operand
    wide_x = plan16 is not None and F % 32 == 0 and G % 32 == 0 and bool(lib.gcrnn_fused_filter_output_wide_supported(B, T, N, F, G, K, int(plan16['entries']), uw, 1, 0))
    wide_h = plan16 is not None and F % 32 == 0 and bool(lib.gcrnn_fused_filter_output_wide_supported(B, 1, N, F, F, K, int(plan16['entries']), uw, 1, 0))
    p16 = (lambda: (_p(plan16['tile_slots']), _p(plan16['tile_off']), _p(plan16['ell_col4']), plan16['entries']))
    if wide_x:
        wpx = _fused_pack_weights_wide(wAc.contiguous(), wAc.new_zeros((F, 1, K, F)), uw, st)
        zx = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=X.device)
        check(lib.gcrnn_fused_filter_output_wide_bf16(_p(xs), _p(wpx), _p(bA.contiguous() if bA is not None else None), _p(zx), *p16(), B, T, N, F, G, K, _p(x_pending), st),
              'fused_filter_output_wide')
    else:
        assert x_pending is None      # (the partial layout above was sized by the same query)
        zx = fused_filter_output(xs, wAc, bA, graph, K, N)                  # [T][B][NPad][F]
    gx = fused_edge_attention(zx, a_in, graph, out=zx, N=N, negative_slope=negative_slope)      # in place: every workgroup reads its item first
    if Kst < K:
        wBc = torch.cat([wBc, wBc.new_zeros(F, 1, K - Kst, F)], dim=2)
    bB32 = bB.contiguous() if bB is not None else None
    if wide_h:
        wpBw = _fused_pack_weights_wide(wBc.contiguous(), wBc.new_zeros((F, 1, K, F)), uw, st)      # the state taps as the "input" taps of [0 | h_{t-1}]
    else:
        wpB = _fused_pack_state_taps(wBc, K, st)
    H = torch.empty((B, 1 if last_only else T, F, N), dtype=torch.bfloat16, device=X.device)
    zh = torch.empty((1, B, npad, F), dtype=torch.bfloat16, device=X.device)
    ga = _fused_graph_args(plan16 or plan)
    for t in range(T):
        if wide_h:
            check(lib.gcrnn_fused_filter_output_wide_bf16(_p(hs_all[t]), _p(wpBw), _p(bB32), _p(zh), *p16(), B, 1, N, F, F, K, None, st), 'fused_filter_output_wide')
        else:
            check(lib.gcrnn_fused_filter_output_bf16(_p(hs_all[t]), None, _p(wpB), _p(bB32), _p(zh), *ga, B, 1, N, F, 0, K, uw, 1 if plan16 else 0, st),
                  'fused_filter_output')
        hu = None
        if not last_only:
            hu = H[:, t]
        elif t == T - 1:
            hu = H[:, 0]
        fused_edge_attention(zh, a_f, graph, gx=gx[t], gi=gi[t] if gi is not None else None, gf=gf[t] if gf is not None else None,
                             out=hs_all[t + 1], Huser=hu, huser_item_stride=(1 if last_only else T) * F * N, N=N,
                             negative_slope=negative_slope)
    return H


def fused_edge_training_supported(graph, N, F, G, Kin, Kst, E=1):
    """Edge-gated training on the fused kernels: the fused BPTT's shapes and the attention kernels' (hub rows beyond the backward
    kernel's 32 register-resident edge records take its slower chunked loop)."""
    if not (fused_training_supported(graph, N, F, G, Kin, Kst, E) and fused_edge_supported(graph, N, F, G, Kin, Kst, torch.bfloat16, E)):
        return False
    return bool(lib.gcrnn_fused_edge_attention_backward_supported(int(N), int(F), int(graph.edge_plan()['max_out_degree'])))


def fused_edge_attention_backward(dpre, r, g, z, a12, graph, N, scratch=None, negative_slope=0.2, dz=None, da_part=None, dgate=None):
    """Backward of fused_edge_attention for one branch over the items of dpre [..][NPad][F] (gcrnn_fused_edge_attention_backward_bf16):
    returns (dz like z, da_part fp32 [items][2][F], dgate fp32 [items] or None); dz / da_part / dgate: optional output buffers."""
    npad, F = z.shape[-2], z.shape[-1]
    items = z.numel() // (npad * F)
    ep = graph.edge_plan()
    if dz is None:
        dz = torch.empty_like(z)
    if da_part is None:
        da_part = torch.empty((items, 2, F), dtype=torch.float32, device=z.device)
    if dgate is None and g is not None:
        dgate = torch.empty((items,), dtype=torch.float32, device=z.device)
    if scratch is None:
        scratch = torch.empty((items, ep['nnz']), dtype=torch.float32, device=z.device)
    assert scratch.numel() >= items * ep['nnz']
    check(lib.gcrnn_fused_edge_attention_backward_bf16(_p(dpre), _p(r), _p(g), _p(z), _p(a12), _p(ep['rowptr']), _p(ep['r_edge']),
                                                       _p(ep['r_order']), _p(ep['t_rowptr']), _p(ep['t_pos']), _p(scratch), _p(dz),
                                                       _p(da_part), _p(dgate), items, int(N), npad, F, ep['nnz'], int(ep['max_out_degree']), float(negative_slope),
                                                       _stream()), 'fused_edge_attention_backward')
    return dz, da_part, dgate


class _FusedEdgeCell(torch.autograd.Function):
    """Edge-gated GGCRNNCell (optionally time-gated too) on the fused kernels with its BPTT (graphML.py:2409-2416 under autograd).
    forward: the x branch for all steps (filter pass with composite taps W_in A_k, attention), then two launches per step (state
    filter pass with W_f B_k, attention + tanh), keeping z, relu(att(z)) of both branches.
    backward, per step from the end: attention backward (dz_t, mixer partials, d gf_t) -> one launch of the data chain on dz_t
    (-> dpre_{t-1}); then the x branch's attention backward over all items, two weight-gradient launches (composite taps of the
    two filters) and the chain rule from the composite taps back to (taps, mixing matrix, bias). No gradient for X or h0."""

    @staticmethod
    def forward(ctx, X, h0, wA, wB, bias, mix_in, w_in, mix_f, w_f, gi, gf, graph, xs, hs_all, slope):
        B, T, G, N = X.shape
        F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
        K = max(Kin, Kst)
        plan = graph.fused_plan()
        npad = plan['npad']
        st = _stream()
        bd = bias.detach() if bias is not None else None
        wAc, bA = _edge_composite(wA.detach(), bd, w_in.detach())
        wBc, bB = _edge_composite(wB.detach(), bd, w_f.detach())
        a_in = mix_in.detach().float().reshape(2, F).contiguous()
        a_f = mix_f.detach().float().reshape(2, F).contiguous()
        gic = gi.detach().float().contiguous() if gi is not None else None
        gfc = gf.detach().float().contiguous() if gf is not None else None
        zx = fused_filter_output(xs, wAc, bA, graph, K, N)
        gx = fused_edge_attention(zx, a_in, graph, N=N, negative_slope=slope)
        wBk = wBc if Kst == K else torch.cat([wBc, wBc.new_zeros(F, 1, K - Kst, F)], dim=2)
        wpB = _fused_pack_state_taps(wBk, K, st)
        bB32 = bB.contiguous() if bB is not None else None
        H = torch.empty((B, T, F, N), dtype=torch.bfloat16, device=X.device)
        zh = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=X.device)
        rh = torch.empty_like(zh)
        plan16 = fused_img16_plan(graph, False, None)
        ga = _fused_graph_args(plan16 or plan)
        uw = plan.get('uniform_w', 0.0)
        for t in range(T):
            check(lib.gcrnn_fused_filter_output_bf16(_p(hs_all[t]), None, _p(wpB), _p(bB32), _p(zh[t]), *ga, B, 1, N, F, 0, K, uw, 1 if plan16 else 0, st),
                  'fused_filter_output')
            fused_edge_attention(zh[t], a_f, graph, gx=gx[t], gi=gic[t] if gic is not None else None,
                                 gf=gfc[t] if gfc is not None else None, out=hs_all[t + 1], r_out=rh[t], Huser=H[:, t],
                                 huser_item_stride=T * F * N, N=N, negative_slope=slope)
        ctx.save_for_backward(X, h0, wA, wB, bias, mix_in, w_in, mix_f, w_f, H, hs_all, zx, gx, zh, rh, gic, gfc, wBk)
        ctx.graph, ctx.npad, ctx.slope = graph, npad, slope
        return H

    @staticmethod
    def backward(ctx, dH):
        X, h0, wA, wB, bias, mix_in, w_in, mix_f, w_f, H, hs_all, zx, gx, zh, rh, gi, gf, wBk = ctx.saved_tensors
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise GcrnnError('the fused edge-gated BPTT does not produce gradients w.r.t. X or h0')
        graph, npad, slope = ctx.graph, ctx.npad, ctx.slope
        B, T, G, N = X.shape
        F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
        K = max(Kin, Kst)
        st = _stream()
        dev = X.device
        hs = hs_all[1:]
        a_in = mix_in.detach().float().reshape(2, F).contiguous()
        a_f = mix_f.detach().float().reshape(2, F).contiguous()
        dH = dH.to(torch.bfloat16).contiguous()
        dHs, dHu = fused_pack_upstream(dH, graph, K)          # uniform graphs: only the last two steps; the chain launches lay out the rest
        wBt = wBk[:, 0].permute(2, 1, 0).contiguous().unsqueeze(1)                  # transposed composite taps [F_in][1][K][F_out]
        wpT = _fused_pack_state_taps(wBt, K, st)
        aplan = graph.fused_plan(adjoint=True)
        aplan16 = None if os.environ.get('GCRNN_NO_IMG16') else graph.fused_plan_img16(adjoint=True)
        aga = _fused_graph_args(aplan16 or aplan)
        auw = aplan.get('uniform_w', 0.0)
        nnz = graph.edge_plan()['nnz']
        tchunk = max(1, min(T, (1 << 26) // max(1, B * nnz)))                       # <= 256 MiB of per-edge scratch
        scratch = torch.empty((tchunk * B, nnz), dtype=torch.float32, device=dev)
        dpre = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=dev)
        dzh = torch.empty_like(dpre)
        da_f = torch.empty((T, B, 2, F), dtype=torch.float32, device=dev)
        dgf = torch.empty((T, B), dtype=torch.float32, device=dev) if gf is not None else None
        step = B * npad * F
        check(lib.gcrnn_fused_backward_seed_bf16(_p(dHs[T - 1]), _p(hs[T - 1]), _p(dpre[T - 1]), step, st), 'backward_seed')
        for t in range(T - 1, -1, -1):
            fused_edge_attention_backward(dpre[t], rh[t], gf[t] if gf is not None else None, zh[t], a_f, graph, N, scratch=scratch,
                                          negative_slope=slope, dz=dzh[t], da_part=da_f[t], dgate=dgf[t] if dgf is not None else None)
            if t > 0:
                nxt = dHu is not None and t >= 2
                check(lib.gcrnn_fused_backward_step_bf16(_p(dzh[t]), _p(dHs[t - 1]), _p(hs[t - 1]), _p(dpre[t - 1]), _p(wpT), *aga,
                                                         B, N, F, K, auw, _p(dHu[:, t - 2]) if nxt else None, _p(dHs[t - 2]) if nxt else None, T,
                                                         1 if aplan16 else 0, st),
                      'fused_backward_step')
        dzx = torch.empty_like(dpre)
        da_i = torch.empty((T, B, 2, F), dtype=torch.float32, device=dev)
        dgi = torch.empty((T, B), dtype=torch.float32, device=dev) if gi is not None else None
        for t0 in range(0, T, tchunk):
            t1 = min(T, t0 + tchunk)
            fused_edge_attention_backward(dpre[t0:t1], gx[t0:t1], gi[t0:t1] if gi is not None else None, zx[t0:t1], a_in, graph, N,
                                          scratch=scratch, negative_slope=slope, dz=dzx[t0:t1], da_part=da_i[t0:t1],
                                          dgate=dgi[t0:t1] if dgi is not None else None)
        one = torch.ones((T, B), dtype=torch.float32, device=dev)
        zero = torch.zeros_like(one)
        dWx, dbx = fused_backward_weight(dzx, X, H, h0, graph, F, G, K, want_bias=True, gi=one, gf=zero)       # composite input taps
        dWh, dbh = fused_backward_weight(dzh, X, H, h0, graph, F, G, K, want_bias=True, gi=zero, gf=one)       # composite state taps
        dAc, dBc = dWx[:, :Kin, F:], dWh[:, :Kst, :F]                                 # [F'][k][C] fp32
        Wi, Wf = w_in.detach()[0, 0].float(), w_f.detach()[0, 0].float()
        A32, B32 = wA.detach()[:, 0].float(), wB.detach()[:, 0].float()              # [F][k][C]
        gA = torch.einsum('pf,pkc->fkc', Wi, dAc).unsqueeze(1).to(wA.dtype) if ctx.needs_input_grad[2] else None
        gB = torch.einsum('pf,pkc->fkc', Wf, dBc).unsqueeze(1).to(wB.dtype) if ctx.needs_input_grad[3] else None
        gWi = torch.einsum('pkc,fkc->pf', dAc, A32)
        gWf = torch.einsum('pkc,fkc->pf', dBc, B32)
        gb = None
        if bias is not None:
            b32 = bias.detach().float().view(-1)
            gWi = gWi + torch.outer(dbx, b32)
            gWf = gWf + torch.outer(dbh, b32)
            if ctx.needs_input_grad[4]:
                gb = (Wi.t() @ dbx + Wf.t() @ dbh).view_as(bias).to(bias.dtype)
        g_mix_in = da_i.view(T * B, 2 * F).sum(dim=0).view_as(mix_in).to(mix_in.dtype) if ctx.needs_input_grad[5] else None
        g_w_in = gWi.view_as(w_in).to(w_in.dtype) if ctx.needs_input_grad[6] else None
        g_mix_f = da_f.view(T * B, 2 * F).sum(dim=0).view_as(mix_f).to(mix_f.dtype) if ctx.needs_input_grad[7] else None
        g_w_f = gWf.view_as(w_f).to(w_f.dtype) if ctx.needs_input_grad[8] else None
        return (None, None, gA, gB, gb, g_mix_in, g_w_in, g_mix_f, g_w_f,
                dgi if (gi is not None and ctx.needs_input_grad[9]) else None,
                dgf if (gf is not None and ctx.needs_input_grad[10]) else None, None, None, None, None)


def fused_edge_cell_train(X, h0, wA, wB, bias, graph, att_in, att_f, time_gates=None, negative_slope=0.2):
    """Training forward of the edge-gated (optionally time + edge gated) cell on the fused kernels; the time gates are autograd
    nodes of their own (_FusedTimeGate) and enter _FusedEdgeCell as differentiable [T][B] inputs."""
    require_device(X, h0, wA, wB, bias)
    with torch.no_grad():
        if time_gates is not None:
            xs, hs_all = fused_pack_inputs_gated(X.contiguous(), h0, graph, wA.shape[0], max(wA.shape[2], wB.shape[2]))
        else:
            xs, hs_all = fused_pack_inputs(X, h0, graph)
        hzero = fused_h0_zero_flag(h0)
    gi = gf = None
    if time_gates is not None:
        gi, gf = fused_train_time_gates(xs, hs_all[:1], X, h0, time_gates, graph, hzero)
        assert getattr(xs, '_pending_user', None) is None
    return _FusedEdgeCell.apply(X, h0, wA, wB, bias, att_in[0], att_in[1], att_f[0], att_f[1], gi, gf, graph, xs, hs_all, float(negative_slope))


def fused_x3_supported(graph, N, F, G, Kin, Kst, dtype, E=1, B=None, T=None):
    """fp32-accurate fused inference (gcrnn_fused_forward_x3): fp32 tensors, un-gated cell, N <= 1024 with N % 4 == 0, the fused
    shapes, and a UNIFORM-weight graph (all non-zeros equal: the drivers' W / lambda_max) with >= 16 padding rows."""
    if E != 1 or dtype != torch.float32 or N > int(lib.gcrnn_fused_padded_nodes()):
        return False
    Gp = fused_padded_inputs(F, G)
    if Gp is None:
        return False
    plan = graph.fused_plan_x3()
    if B is not None and T is not None:
        # limits of the x3 pack (gridDim.z = B * T) and of the 32-bit offsets into the three-plane arrays: such batches fall back to the
        # composed path instead of failing inside the forward (or, worse, inside the backward after the forward has run)
        if B * T > 65535 or 3 * B * plan['npad'] * max(F, Gp) * 2 > 2 ** 31 - 1:
            return False
    return plan.get('uniform_w', 0.0) != 0.0 and bool(lib.gcrnn_fused_x3_supported(int(N), int(F), int(Gp), int(max(Kin, Kst)), plan['entries']))


def time_fused_x3_kernel(X, h0, wA, wB, bias, graph, reps=3):
    """Average duration of ONE launch of the fp32-accurate step kernel (HIP events on the launch stream, inputs pre-packed)."""
    X, wA = fused_pad_operands(X, wA.detach())
    B, T, G, N = X.shape
    F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
    K = max(Kin, Kst)
    plan = graph.fused_plan_x3()
    npad, st, dev = plan['npad'], _stream(), X.device
    Xc, h0c = X.contiguous(), h0.contiguous()
    xs3 = torch.empty((T, 3, B, npad, G), dtype=torch.bfloat16, device=dev)
    h03 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    hs3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_pack_seq_major_x3(_p(Xc), _p(xs3), B, T, G, N, npad, st), 'pack_seq_x3')
    check(lib.gcrnn_pack_seq_major_x3(_p(h0c), _p(h03), B, 1, F, N, npad, st), 'pack_seq_x3')
    wAc, wBc = wA.float().contiguous(), wB.detach().float().contiguous()
    wp3 = torch.empty((3 * (F // 16) * K * ((F + G) // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_fused_pack_weights_x3(_p(wAc), _p(wBc), _p(wp3), F, G, Kin, Kst, st), 'pack_weights_x3')
    b32 = bias.detach().float().contiguous().view(-1) if bias is not None else None
    H = torch.empty((B, T, F, N), dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        check(lib.gcrnn_fused_forward_x3(_p(xs3), _p(h03), _p(hs3), _p(wp3), _p(b32), _p(plan['tile_slots']), _p(plan['tile_off']),
                                         _p(plan['ell_col4']), plan['entries'], B, T, N, F, G, K, plan['uniform_w'], _p(H), 0, _p(plan.get('rank1_x3')), st),
              'fused_forward_x3')
    e1.record()
    torch.cuda.synchronize()
    return {'avg_us': 1e3 * e0.elapsed_time(e1) / (reps * T), 'launches': reps * T}


def fused_cell_forward_x3(X, h0, wA, wB, bias, graph, last_only=False, keep=False):
    """Un-gated GGCRNNCell forward to fp32 accuracy on the fused kernels (three bf16 planes per fp32 operand, six partial
    products per tap product on the bf16 matrix cores, fp32 hops / tanh). X: B x T x G x N fp32, h0: B x F x N fp32 ->
    H: B x T x F x N fp32 (B x 1 x F x N with last_only). No autograd graph here; keep: also returns the planes of the states
    hs3 [T][3][B][NPad][F] and the (channel-padded) X the kernels ran on -- what the fp32-accurate BPTT needs."""
    require_device(X, h0, wA, wB, bias)
    X, wA = fused_pad_operands(X, wA.detach())
    B, T, G, N = X.shape
    F = wA.shape[0]
    Kin, Kst = wA.shape[2], wB.shape[2]
    K = max(Kin, Kst)
    plan = graph.fused_plan_x3()
    npad = plan['npad']
    st = _stream()
    dev = X.device
    Xc, h0c = X.contiguous(), h0.contiguous()
    xs3 = torch.empty((T, 3, B, npad, G), dtype=torch.bfloat16, device=dev)
    h03 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    hs3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_pack_seq_major_x3(_p(Xc), _p(xs3), B, T, G, N, npad, st), 'pack_seq_x3')
    check(lib.gcrnn_pack_seq_major_x3(_p(h0c), _p(h03), B, 1, F, N, npad, st), 'pack_seq_x3')
    wAc, wBc = wA.float().contiguous(), wB.detach().float().contiguous()
    wp3 = torch.empty((3 * (F // 16) * K * ((F + G) // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_fused_pack_weights_x3(_p(wAc), _p(wBc), _p(wp3), F, G, Kin, Kst, st), 'pack_weights_x3')
    b32 = bias.detach().float().contiguous().view(-1) if bias is not None else None
    H = torch.empty((B, 1 if last_only else T, F, N), dtype=torch.float32, device=dev)
    check(lib.gcrnn_fused_forward_x3(_p(xs3), _p(h03), _p(hs3), _p(wp3), _p(b32), _p(plan['tile_slots']), _p(plan['tile_off']),
                                     _p(plan['ell_col4']), plan['entries'], B, T, N, F, G, K, plan['uniform_w'], _p(H),
                                     int(last_only), _p(plan.get('rank1_x3')), st), 'fused_forward_x3')
    if keep:
        return H, hs3, Xc
    return H


def fused_cell_forward_x3_gated(X, h0, wA, wB, bias, graph, gates, last_only=False):
    """Time-gated GGCRNNCell forward to fp32 accuracy on the fp32-accurate fused kernels (round 3; round 4: the gate cells run on the planes
    of X instead of per-item copies -- _x3_time_gated_forward). gates = {'in': (wA_g, wB_g, bias_g, lin_w, lin_b), 'forget': (...)} as in
    fused_cell_forward. X: B x T x G x N fp32, h0: B x F x N fp32 -> H: B x T x F x N fp32 (B x 1 x F x N with last_only). No autograd graph."""
    return _x3_time_gated_forward(X, h0, wA, wB, bias, graph, gates, keep=False, last_only=last_only)


def fused_x3_training_supported(graph, N, F, G, Kin, Kst, dtype, E=1, B=None, T=None):
    """fp32-accurate training of the un-gated cell on the fused kernels: the x3 forward's conditions, and the ADJOINT graph uniform
    too (a symmetric-support GSO like the drivers' W / lambda_max) with an image that fits next to the backward's two fp32 images."""
    if not fused_x3_supported(graph, N, F, G, Kin, Kst, dtype, E, B, T):
        return False
    pa = graph.fused_plan_x3(adjoint=True)
    Gp = fused_padded_inputs(F, G)
    return pa.get('uniform_w', 0.0) != 0.0 and bool(lib.gcrnn_fused_x3_training_supported(int(N), int(F), int(Gp), int(max(Kin, Kst)), int(pa['entries'])))


class _FusedCellX3(torch.autograd.Function):
    """Un-gated GGCRNNCell at fp32 accuracy on the fused kernels, forward AND BPTT (round 3): x3 forward; data chain on the x3 step
    kernel with the chain epilogue (gcrnn_fused_backward_data_x3); weight gradient on exact-fp32 matrix instructions
    (gcrnn_fused_backward_weight_f32). Gradients for the taps, the bias and h0; X gets none (the training loops never ask)."""

    @staticmethod
    def forward(ctx, X, h0, wA, wB, bias, graph):
        H, hs3, Xp = fused_cell_forward_x3(X, h0, wA, wB, bias, graph, keep=True)
        ctx.save_for_backward(Xp, h0, wA, wB, bias, H, hs3)
        ctx.graph = graph
        ctx.G = X.shape[2]
        return H

    @staticmethod
    def backward(ctx, dH):
        Xp, h0, wA, wB, bias, H, hs3 = ctx.saved_tensors
        graph = ctx.graph
        if ctx.needs_input_grad[0]:
            raise GcrnnError('the fp32-accurate fused BPTT does not produce the gradient w.r.t. the input sequence X')
        B, T, Gp, N = Xp.shape
        F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
        K = max(Kin, Kst)
        plan = graph.fused_plan_x3(adjoint=True)
        npad, st, dev = plan['npad'], _stream(), Xp.device
        dHc = dH.float().contiguous()
        dH3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_pack_seq_major_x3(_p(dHc), _p(dH3), B, T, F, N, npad, st), 'pack_seq_x3')
        wBk = wB.detach().float()
        if Kst < K:
            wBk = torch.cat([wBk, wBk.new_zeros(F, 1, K - Kst, F)], dim=2)
        wBt = wBk[:, 0].permute(2, 1, 0).contiguous()                       # [F_in][K][F_out]: transposed taps
        wp3T = torch.empty((3 * (F // 16) * K * (F // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_fused_pack_weights_x3(_p(wBt), _p(wBt), _p(wp3T), F, 0, K, K, st), 'pack_weights_x3')
        dpre3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
        dh03 = torch.empty((3, B, npad, F), dtype=torch.bfloat16, device=dev) if ctx.needs_input_grad[1] else None
        check(lib.gcrnn_fused_backward_data_x3(_p(dH3), _p(hs3), _p(dpre3), _p(dh03), _p(wp3T), _p(plan['tile_slots']), _p(plan['tile_off']),
                                               _p(plan['ell_col4']), plan['entries'], B, T, N, F, K, plan['uniform_w'], _p(plan.get('rank1_x3')), st), 'fused_backward_data_x3')
        slots = int(lib.gcrnn_fused_wgrad_slots(T * B, F))
        dWp = torch.zeros((slots, F, K, F + Gp), dtype=torch.float32, device=dev)
        dbp = torch.zeros((slots, F), dtype=torch.float32, device=dev)
        h0c = h0.detach().float().contiguous()
        check(lib.gcrnn_fused_backward_weight_f32(_p(dpre3), _p(Xp), _p(H), _p(h0c), _p(dWp), _p(dbp), _p(plan['tile_slots']), _p(plan['tile_off']),
                                                  _p(plan['ell_col4']), plan['entries'], B, T, N, F, Gp, K, plan['uniform_w'], _p(plan.get('rank1_x3')), st), 'fused_backward_weight_f32')
        dW = dWp.sum(dim=0)                                                 # fixed order over the slots: bit-reproducible
        G = ctx.G
        gA = dW[:, :Kin, F:F + G].unsqueeze(1).to(wA.dtype) if ctx.needs_input_grad[2] else None
        gB = dW[:, :Kst, :F].unsqueeze(1).to(wB.dtype) if ctx.needs_input_grad[3] else None
        gb = dbp.sum(dim=0).view_as(bias).to(bias.dtype) if (bias is not None and ctx.needs_input_grad[4]) else None
        gh0 = None
        if dh03 is not None:
            gh0 = (dh03[2].float() + dh03[1].float() + dh03[0].float())[:, :N, :].permute(0, 2, 1).contiguous().to(h0.dtype)
        return None, gh0, gA, gB, gb, None


def fused_cell_train_x3(X, h0, wA, wB, bias, graph):
    """Training forward of the un-gated cell at fp32 accuracy on the fused kernels (fp32 tensors and parameters)."""
    require_device(X, h0, wA, wB, bias)
    return _FusedCellX3.apply(X, h0, wA, wB, bias, graph)


def _x3_planes_to_f32(p3):
    """[.., 3, B, NPad, C] bf16 planes -> fp32 (small to large: the sum is exact for planes cut from one fp32 value)."""
    return p3.select(-4, 2).float() + p3.select(-4, 1).float() + p3.select(-4, 0).float()


def _x3_item_slices(B, T):
    """(t0, nt) slices of the T*B items (t, b) that keep the x3 kernels' 32-bit buffer offsets and the pack's grid limit."""
    per = max(1, min(T, 2048 // B))
    return [(t0, min(per, T - t0)) for t0 in range(0, T, per)]


def fused_x3_time_training_supported(graph, N, F, G, Kin, Kst, dtype, E=1, B=None, T=None):
    """fp32-accurate training of the TIME-GATED cell on the fused kernels (ops.fused_cell_train_x3_gated): the un-gated x3 training's
    conditions; batches up to 2048 sequences (the gate cells run over slices of the T*B items)."""
    return (B is None or B <= 2048) and fused_x3_training_supported(graph, N, F, G, Kin, Kst, dtype, E, B, T)


def _x3_time_gated_forward(X, h0, wA, wB, bias, graph, gates, keep=False, last_only=False):
    """Forward of the time-gated cell at fp32 accuracy (reference graphML.py:2357-2374, 2420-2423): both gate cells as T x B one-step x3 cells
    that read (x_t, h0) on the planes of X (gcrnn_fused_gate_cells_x3), the Linear(F N -> 1) read-outs as GEMVs over their fp32 states, the
    recurrence as one scaled x3 step per launch -- gi (A(S)x_t + b) + gf (B(S)h_{t-1} + b) = A(S)(gi x_t) + B(S)(gf h_{t-1}) + (gi + gf) b, the
    operands scaled in fp32 while they are cut into planes (gcrnn_pack_seq_major_x3_ex). keep: what the BPTT needs."""
    require_device(X, h0, wA, wB, bias)
    Xp, wAp = fused_pad_operands(X, wA.detach())
    B, T, Gp, N = Xp.shape
    F, Kin, Kst = wAp.shape[0], wAp.shape[2], wB.shape[2]
    K = max(Kin, Kst)
    plan = graph.fused_plan_x3()
    npad, st, dev = plan['npad'], _stream(), X.device
    Xp = Xp.float().contiguous()
    h0c = h0.detach().float().contiguous()
    # h0 == 0 (every training loop of the reference, train_rnn.py:256): the gate cells skip the state operand. A host-side read, so not under
    # stream capture (a captured step takes the general form: same results)
    hzero = False if torch.cuda.is_current_stream_capturing() else not bool(h0c.any())
    gargs = (_p(plan['tile_slots']), _p(plan['tile_off']), _p(plan['ell_col4']), plan['entries'])
    xs3 = torch.empty((T, 3, B, npad, Gp), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_pack_seq_major_x3(_p(Xp), _p(xs3), B, T, Gp, N, npad, st), 'pack_seq_x3')
    h03 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_pack_seq_major_x3(_p(h0c), _p(h03), B, 1, F, N, npad, st), 'pack_seq_x3')
    h3 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    gvals, cs = [], []
    for (wA_g, wB_g, bias_g, lin_w, lin_b) in (gates['in'], gates['forget']):
        Kg = max(wA_g.shape[2], wB_g.shape[2])
        wAg, wBg = wA_g.detach().float().contiguous(), wB_g.detach().float().contiguous()
        wpg = torch.empty((3 * (F // 16) * Kg * ((F + Gp) // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_fused_pack_weights_x3(_p(wAg), _p(wBg), _p(wpg), F, Gp, wA_g.shape[2], wB_g.shape[2], st), 'pack_weights_x3')
        bg = bias_g.detach().float().contiguous().view(-1) if bias_g is not None else None
        c = torch.empty((B, T, F, N), dtype=torch.float32, device=dev)          # the gate cells' states (training keeps them for their BPTT)
        check(lib.gcrnn_fused_gate_cells_x3(_p(xs3), _p(h03), _p(h3), _p(wpg), _p(bg), *gargs, B, T, N, F, Gp, Kg, plan['uniform_w'], _p(c), int(hzero), _p(plan.get('rank1_x3')), st),
              'fused_gate_cells_x3')
        logit = (c.view(B * T, F * N) @ lin_w.detach().float().reshape(-1)).view(B, T).t()
        if lin_b is not None:
            logit = logit + lin_b.detach().float().view(())
        gvals.append(torch.sigmoid(logit).contiguous())                          # [T][B]
        cs.append(c if keep else None)
        del c
    gi, gf = gvals
    xu3 = xs3 if (keep and Gp == F) else None                                               # (the backward's filter pass A(S) x_t reads the planes of X again)
    if xu3 is not None:
        xs3 = torch.empty_like(xu3)
    check(lib.gcrnn_pack_seq_major_x3_ex(_p(Xp), _p(xs3), B, T, Gp, N, npad, _p(gi), None, 0, 0, st), 'pack_seq_x3_ex')      # planes of gi x_t
    wAc, wBc = wAp.float().contiguous(), wB.detach().float().contiguous()
    wp3 = torch.empty((3 * (F // 16) * K * ((F + Gp) // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_fused_pack_weights_x3(_p(wAc), _p(wBc), _p(wp3), F, Gp, Kin, Kst, st), 'pack_weights_x3')
    b32 = bias.detach().float().contiguous().view(-1) if bias is not None else None
    bsc = (gi + gf).contiguous()
    H = torch.empty((B, T, F, N), dtype=torch.float32, device=dev)
    hs3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    for t in range(T):
        hprev = h0c if t == 0 else H[:, t - 1]
        check(lib.gcrnn_pack_seq_major_x3_ex(_p(hprev), _p(h3), B, 1, F, N, npad, _p(gf[t]), None, 0, 0 if t == 0 else T * F * N, st),
              'pack_seq_x3_ex')                                                                  # planes of gf_t h_{t-1}
        check(lib.gcrnn_fused_forward_x3_scaled(_p(xs3[t]), _p(h3), _p(hs3[t]), _p(wp3), _p(b32), _p(bsc[t]), *gargs, B, 1, N, F, Gp, K,
                                                plan['uniform_w'], _p(H[:, t]), T * F * N, _p(plan.get('rank1_x3')), st), 'fused_forward_x3_scaled')
    del xs3
    if last_only:
        return H[:, T - 1:].contiguous()
    return (H, Xp, h0c, hs3, gi, gf, cs[0], cs[1], hzero, xu3) if keep else H



def fused_node_x3_supported(graph, N, F, G, Kin, Kst, dtype, E=1, B=None, T=None):
    """Node-gated cell at fp32 accuracy on the x3 kernels (fused_node_cell_forward_x3): the x3 forward's conditions with G == F (the input
    filter runs as an F -> F x3 filter pass over the planes of X)."""
    return G == F and fused_x3_supported(graph, N, F, G, Kin, Kst, dtype, E, B, T) and not os.environ.get('GCRNN_NO_X3_NODE')


def fused_node_cell_forward_x3(X, h0, wA, wB, bias, graph, node_gates, time_gates=None, last_only=False):
    """Node-gated GGCRNNCell forward (optionally time-gated too) to fp32 accuracy on the fp32-accurate fused kernels (round 5; reference
    Utils/graphML.py:2379-2407, 2420-2423, in the drivers' precision class, kStepPredGRNNs.py:44):
        h_t = tanh( gi_t ni_t (.) (A(S) x_t + b) + gf_t nf_t (.) (B(S) h_{t-1} + b) )
    * both gate cells as T x B one-step x3 cells on the planes of X (gcrnn_fused_gate_cells_x3: they read (x_t, h0), never h_{t-1}), their
      F -> 1 graph filters on the fp32 any-shape filter kernels (lsigf_node_major), sigmoid;
    * A(S) x_t for all steps and B(S) h_{t-1} per step as x3 filter passes (gcrnn_fused_filter_x3: three bf16 planes per operand, exact fp32
      products on the matrix cores), the per-node gating and the tanh in fp32 between them -- the per-node gates multiply the filters'
      OUTPUTS, so (unlike the scalar time gates) they cannot be folded into the operands.
    node_gates = {'in': (wA_g, wB_g, bias_g, wf, bf), 'forget': (...)}, time_gates as in fused_cell_forward_x3_gated. X: B x T x G x N fp32 with
    G == F, h0: B x F x N fp32 -> H: B x T x F x N fp32 (B x 1 x F x N with last_only). Inference (no autograd graph)."""
    require_device(X, h0, wA, wB, bias)
    B, T, G, N = X.shape
    F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
    assert G == F, 'the x3 filter pass takes F -> F filters'
    K = max(Kin, Kst)
    plan = graph.fused_plan_x3()
    npad, st, dev = plan['npad'], _stream(), X.device
    Xc = X.detach().float().contiguous()
    h0c = h0.detach().float().contiguous()
    hzero = False if torch.cuda.is_current_stream_capturing() else not bool(h0c.any())
    gargs = (_p(plan['tile_slots']), _p(plan['tile_off']), _p(plan['ell_col4']), plan['entries'])
    r1 = _p(plan.get('rank1_x3'))
    xs3 = torch.empty((T, 3, B, npad, G), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_pack_seq_major_x3(_p(Xc), _p(xs3), B, T, G, N, npad, st), 'pack_seq_x3')
    h03 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_pack_seq_major_x3(_p(h0c), _p(h03), B, 1, F, N, npad, st), 'pack_seq_x3')
    h3 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)

    def gate_cell_states(wA_g, wB_g, bias_g):
        Kg = max(wA_g.shape[2], wB_g.shape[2])
        wAg, wBg = wA_g.detach().float().contiguous(), wB_g.detach().float().contiguous()
        wpg = torch.empty((3 * (F // 16) * Kg * ((F + G) // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_fused_pack_weights_x3(_p(wAg), _p(wBg), _p(wpg), F, G, wA_g.shape[2], wB_g.shape[2], st), 'pack_weights_x3')
        bg = bias_g.detach().float().contiguous().view(-1) if bias_g is not None else None
        c = torch.empty((B, T, F, N), dtype=torch.float32, device=dev)
        check(lib.gcrnn_fused_gate_cells_x3(_p(xs3), _p(h03), _p(h3), _p(wpg), _p(bg), *gargs, B, T, N, F, G, Kg, plan['uniform_w'], _p(c), int(hzero), r1, st),
              'fused_gate_cells_x3')
        return c

    # per-node gates [T][B][npad][1] fp32 (zero on the padding rows)
    Kf = node_gates['in'][3].shape[2]
    uw_f = graph.fused_plan().get('uniform_w', 0.0)      # (the x3 plan of a rank-1-weighted graph is its 0 / 1 pattern with uniform_w = 1: the gate filter reads the GSO itself)
    taps_first = (node_gates['forget'][3].shape[2] == Kf and not os.environ.get('GCRNN_NO_NODE_GATE_FILTER')
                  and bool(lib.gcrnn_node_gate_filter_supported(Kf, N, graph.fwd[0].nnz, uw_f)))
    if taps_first:
        # the F -> 1 filters taps-first (as the bf16 path): u_k = c w_k per node in fp32, then the K - 1 hops on ONE channel with the bias and the
        # sigmoid in the same pass (gcrnn_node_gate_filter_f32) -- the filter on the F-channel states cost half of this forward (34 of 69 ms)
        parts = torch.empty((T, B, 2, 1, Kf, N), dtype=torch.float32, device=dev)
        for gidx, name in enumerate(('in', 'forget')):
            wA_g, wB_g, bias_g, wf, bf = node_gates[name]
            c = gate_cell_states(wA_g, wB_g, bias_g)
            parts[:, :, gidx, 0] = torch.einsum('kf,btfn->tbkn', wf.detach().float()[0, 0], c)
            del c
        bfs = tuple(node_gates[name][4] for name in ('in', 'forget'))
        b2 = None
        if any(b is not None for b in bfs):
            b2 = torch.cat([(b.detach().float().reshape(1) if b is not None else torch.zeros(1, device=dev)) for b in bfs]).contiguous()
        csr = graph.fwd[0]
        ng2 = torch.empty((T, 2, B, N), dtype=torch.float32, device=dev)
        check(lib.gcrnn_node_gate_filter_f32(_p(parts), _p(ng2), T * B * 2, 1, Kf, N, 2, B, _p(csr.rowptr), _p(csr.col), _p(csr.val(torch.float32)), csr.nnz,
                                             uw_f, _p(b2), 1, st), 'node_gate_filter')
        del parts
        ngate = []
        for gidx in range(2):
            g = torch.zeros((T, B, npad, 1), dtype=torch.float32, device=dev)
            g[:, :, :N, 0] = ng2[:, gidx]
            ngate.append(g)
        del ng2
    else:
        ngate = []
        for name in ('in', 'forget'):
            wA_g, wB_g, bias_g, wf, bf = node_gates[name]
            c = gate_cell_states(wA_g, wB_g, bias_g)
            logit = lsigf_node_major(pack_node_major(c), wf.detach().float(), bf.detach().float() if bf is not None else None, graph, 1.0)      # [T][N][B][1]
            del c
            g = torch.zeros((T, B, npad, 1), dtype=torch.float32, device=dev)
            g[:, :, :N] = torch.sigmoid(logit).permute(0, 2, 1, 3)
            ngate.append(g)
            del logit
    ni, nf = ngate
    if time_gates is not None:
        for which, name in ((0, 'in'), (1, 'forget')):
            wA_g, wB_g, bias_g, lin_w, lin_b = time_gates[name]
            c = gate_cell_states(wA_g, wB_g, bias_g)
            logit = (c.view(B * T, F * N) @ lin_w.detach().float().reshape(-1)).view(B, T).t()
            if lin_b is not None:
                logit = logit + lin_b.detach().float().view(())
            tgv = torch.sigmoid(logit).contiguous().view(T, B, 1, 1)
            del c
            if which == 0:
                ni = ni * tgv
            else:
                nf = nf * tgv

    def state_taps(w, k):      # F x 1 x k x F taps -> the x3 pack of a state-only operand with K taps
        wk = w.detach().float()
        if k < K:
            wk = torch.cat([wk, wk.new_zeros(F, 1, K - k, F)], dim=2)
        wk = wk.contiguous()
        wp = torch.empty((3 * (F // 16) * K * (F // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_fused_pack_weights_x3(_p(wk), _p(wk), _p(wp), F, 0, K, K, st), 'pack_weights_x3')
        return wp
    wp3A, wp3B = state_taps(wA, Kin), state_taps(wB, Kst)
    b32 = bias.detach().float().contiguous().view(-1) if bias is not None else None
    ya3 = torch.empty((3, B, npad, F), dtype=torch.bfloat16, device=dev)
    yb3 = torch.empty((3, B, npad, F), dtype=torch.bfloat16, device=dev)
    H = torch.empty((B, T, F, N), dtype=torch.float32, device=dev)
    check(lib.gcrnn_pack_seq_major_x3(_p(h0c), _p(h3), B, 1, F, N, npad, st), 'pack_seq_x3')
    for t in range(T):
        # both filter passes, then ONE pass: planes -> fp32, bias, per-node gating, tanh, the planes of h_t (the next step's operand) and H[:, t]
        check(lib.gcrnn_fused_filter_x3(_p(xs3[t]), _p(ya3), _p(wp3A), *gargs, B, N, F, K, plan['uniform_w'], r1, st), 'fused_filter_x3')      # A(S) x_t
        check(lib.gcrnn_fused_filter_x3(_p(h3[0]), _p(yb3), _p(wp3B), *gargs, B, N, F, K, plan['uniform_w'], r1, st), 'fused_filter_x3')       # B(S) h_{t-1}
        check(lib.gcrnn_x3_node_gate_step(_p(ya3), _p(yb3), _p(ni[t]), _p(nf[t]), _p(b32), _p(h3), _p(H[:, t]), T * F * N, B, N, npad, F, st), 'x3_node_gate_step')
    if last_only:
        return H[:, T - 1:].contiguous()
    return H


def fused_edge_cell_forward_x3(X, h0, wA, wB, bias, graph, att_in, att_f, time_gates=None, last_only=False):
    """Edge-gated GGCRNNCell forward (optionally time-gated too) to fp32 accuracy (round 5; reference Utils/graphML.py:2409-2416, 2420-2423,
    graphAttention :521-627):  h_t = tanh( gi_t GAT_in(A(S) x_t + b) + gf_t GAT_f(B(S) h_{t-1} + b) ).
    Both filters as x3 filter passes (gcrnn_fused_filter_x3: three bf16 planes per operand, exact fp32 products), the attentions on the fp32
    CSR edge-softmax kernels (att_in / att_f: callables on node-major [T][N][B][F] fp32 tensors, GraphAttentional.forward_node_major), the time
    gates' cells as one-step x3 cells. X: B x T x G x N fp32 with G == F, h0: B x F x N fp32 -> H: B x T x F x N fp32. Inference."""
    require_device(X, h0, wA, wB, bias)
    B, T, G, N = X.shape
    F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
    assert G == F, 'the x3 filter pass takes F -> F filters'
    K = max(Kin, Kst)
    plan = graph.fused_plan_x3()
    npad, st, dev = plan['npad'], _stream(), X.device
    Xc = X.detach().float().contiguous()
    h0c = h0.detach().float().contiguous()
    gargs = (_p(plan['tile_slots']), _p(plan['tile_off']), _p(plan['ell_col4']), plan['entries'])
    r1 = _p(plan.get('rank1_x3'))
    xs3 = torch.empty((T, 3, B, npad, G), dtype=torch.bfloat16, device=dev)
    check(lib.gcrnn_pack_seq_major_x3(_p(Xc), _p(xs3), B, T, G, N, npad, st), 'pack_seq_x3')
    h3 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
    gi = gf = None
    if time_gates is not None:
        hzero = False if torch.cuda.is_current_stream_capturing() else not bool(h0c.any())
        h03 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_pack_seq_major_x3(_p(h0c), _p(h03), B, 1, F, N, npad, st), 'pack_seq_x3')
        gv = []
        for name in ('in', 'forget'):
            wA_g, wB_g, bias_g, lin_w, lin_b = time_gates[name]
            Kg = max(wA_g.shape[2], wB_g.shape[2])
            wAg, wBg = wA_g.detach().float().contiguous(), wB_g.detach().float().contiguous()
            wpg = torch.empty((3 * (F // 16) * Kg * ((F + G) // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
            check(lib.gcrnn_fused_pack_weights_x3(_p(wAg), _p(wBg), _p(wpg), F, G, wA_g.shape[2], wB_g.shape[2], st), 'pack_weights_x3')
            bg = bias_g.detach().float().contiguous().view(-1) if bias_g is not None else None
            c = torch.empty((B, T, F, N), dtype=torch.float32, device=dev)
            check(lib.gcrnn_fused_gate_cells_x3(_p(xs3), _p(h03), _p(h3), _p(wpg), _p(bg), *gargs, B, T, N, F, G, Kg, plan['uniform_w'], _p(c), int(hzero), r1, st),
                  'fused_gate_cells_x3')
            logit = (c.view(B * T, F * N) @ lin_w.detach().float().reshape(-1)).view(B, T).t()
            if lin_b is not None:
                logit = logit + lin_b.detach().float().view(())
            gv.append(torch.sigmoid(logit).contiguous())      # [T][B]
            del c
        gi, gf = gv

    def state_taps(w, k):
        wk = w.detach().float()
        if k < K:
            wk = torch.cat([wk, wk.new_zeros(F, 1, K - k, F)], dim=2)
        wk = wk.contiguous()
        wp = torch.empty((3 * (F // 16) * K * (F // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_fused_pack_weights_x3(_p(wk), _p(wk), _p(wp), F, 0, K, K, st), 'pack_weights_x3')
        return wp
    wp3A, wp3B = state_taps(wA, Kin), state_taps(wB, Kst)
    bvec = bias.detach().float().view(1, 1, F) if bias is not None else None
    y3 = torch.empty((3, B, npad, F), dtype=torch.bfloat16, device=dev)

    def filt(z3, wp):      # [3][B][npad][F] planes -> filter output + bias, node-major [1][N][B][F] fp32
        check(lib.gcrnn_fused_filter_x3(_p(z3), _p(y3), _p(wp), *gargs, B, N, F, K, plan['uniform_w'], r1, st), 'fused_filter_x3')
        y = torch.sum(y3, dim=0, dtype=torch.float32)      # (the three planes add up exactly in fp32)
        if bvec is not None:
            y = y + bvec
        return y[:, :N].permute(1, 0, 2).unsqueeze(0).contiguous()

    H = torch.empty((B, T, F, N), dtype=torch.float32, device=dev)
    hprev = h0c
    for t in range(T):
        gx = att_in(filt(xs3[t], wp3A))[0]                                   # [N][B][F]
        hp = hprev if hprev.is_contiguous() else hprev.contiguous()
        check(lib.gcrnn_pack_seq_major_x3(_p(hp), _p(h3), B, 1, F, N, npad, st), 'pack_seq_x3')
        gh = att_f(filt(h3[0], wp3B))[0]
        if gi is not None:
            pre = gi[t].view(1, B, 1) * gx + gf[t].view(1, B, 1) * gh
        else:
            pre = gx + gh
        H[:, t] = torch.tanh(pre).permute(1, 2, 0)
        hprev = H[:, t]
    if last_only:
        return H[:, T - 1:].contiguous()
    return H


class _FusedTimeCellX3(torch.autograd.Function):
    """Time-gated GGCRNNCell (the reference's default, Utils/graphML.py:2196, :2357-2374, :2420-2423) at fp32 accuracy on the fused kernels,
    forward AND BPTT (round 4). Forward: the two gate cells as T x B one-step x3 cells that all read h0 (gcrnn_fused_gate_cells_x3, on the
    planes of X) + the Linear(F N -> 1) read-out; the recurrence as scaled x3 steps (operands scaled while they are cut into planes), keeping
    the states' planes. Backward: the gated x3 data chain (gcrnn_fused_backward_data_x3_gated; the forget gate's gradient is read off the
    chain), d loss / d gi off one x3 filter pass A(S) x_t (gcrnn_fused_filter_x3 + gcrnn_x3_item_dots), the cell's weight gradient on
    exact-fp32 matrix instructions with the gates as operand weights (gcrnn_fused_backward_weight_f32_gated), and the gate cells' gradients:
    read-out by one GEMV over the kept gate states, taps / biases by the same weight-gradient kernel over all items (state operand = h0,
    skipped when it is zero). Gradients for every parameter; X and h0 get none (the training loops never ask: train_rnn.py:256 starts from
    zeros)."""

    @staticmethod
    def forward(ctx, X, h0, wA, wB, bias, gA_i, gB_i, gb_i, lw_i, lb_i, gA_f, gB_f, gb_f, lw_f, lb_f, graph):
        gates = {'in': (gA_i, gB_i, gb_i, lw_i, lb_i), 'forget': (gA_f, gB_f, gb_f, lw_f, lb_f)}
        H, Xp, h0c, hs3, gi, gf, c_i, c_f, hzero, xu3 = _x3_time_gated_forward(X, h0, wA, wB, bias, graph, gates, keep=True)
        ctx.save_for_backward(Xp, h0c, wA, wB, bias, gA_i, gB_i, gb_i, lw_i, lb_i, gA_f, gB_f, gb_f, lw_f, lb_f, H, hs3, gi, gf, c_i, c_f)
        ctx.graph, ctx.G, ctx.hzero, ctx.xu3 = graph, X.shape[2], hzero, xu3
        return H

    @staticmethod
    def backward(ctx, dH):
        (Xp, h0c, wA, wB, bias, gA_i, gB_i, gb_i, lw_i, lb_i, gA_f, gB_f, gb_f, lw_f, lb_f, H, hs3, gi, gf, c_i, c_f) = ctx.saved_tensors
        graph = ctx.graph
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise GcrnnError('the fp32-accurate fused BPTT of the time-gated cell does not produce the gradients w.r.t. X and h0')
        B, T, Gp, N = Xp.shape
        F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
        K = max(Kin, Kst)
        plan, pa = graph.fused_plan_x3(), graph.fused_plan_x3(adjoint=True)
        npad, st, dev = pa['npad'], _stream(), Xp.device
        gargs = (_p(plan['tile_slots']), _p(plan['tile_off']), _p(plan['ell_col4']), plan['entries'])
        aargs = (_p(pa['tile_slots']), _p(pa['tile_off']), _p(pa['ell_col4']), pa['entries'])
        nparts = (F // 16) * 8
        # ---- the gated data chain, d gf read off it ----
        dHc = dH.float().contiguous()
        dH3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_pack_seq_major_x3(_p(dHc), _p(dH3), B, T, F, N, npad, st), 'pack_seq_x3')
        h03 = torch.empty((1, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_pack_seq_major_x3(_p(h0c), _p(h03), B, 1, F, N, npad, st), 'pack_seq_x3')
        wBk = wB.detach().float()
        if Kst < K:
            wBk = torch.cat([wBk, wBk.new_zeros(F, 1, K - Kst, F)], dim=2)
        wBt = wBk[:, 0].permute(2, 1, 0).contiguous()
        wp3T = torch.empty((3 * (F // 16) * K * (F // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_fused_pack_weights_x3(_p(wBt), _p(wBt), _p(wp3T), F, 0, K, K, st), 'pack_weights_x3')
        dpre3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
        dh03 = torch.empty((3, B, npad, F), dtype=torch.bfloat16, device=dev)
        parts = torch.empty((T, B, nparts), dtype=torch.float32, device=dev)
        check(lib.gcrnn_fused_backward_data_x3_gated(_p(dH3), _p(hs3), _p(dpre3), _p(dh03), _p(wp3T), *aargs, B, T, N, F, K, pa['uniform_w'],
                                                     _p(gf), _p(h03), _p(parts), _p(pa.get('rank1_x3')), st), 'fused_backward_data_x3_gated')
        del dH3, dh03
        # ---- d gi = <A(S) x_t + b, dpre_t>: one filter pass per step over the planes of X (the input taps as an F -> F filter, zero columns
        # beyond G), dotted with dpre per item; the bias part <b, sum_n dpre_t> is shared with d gf ----
        bvec = bias.detach().float().contiguous().view(-1) if bias is not None else torch.zeros(F, dtype=torch.float32, device=dev)
        wAk = wA.detach().float()[:, 0]                                                              # [F][Kin][G]
        wAsq = torch.zeros((F, K, F), dtype=torch.float32, device=dev)
        wAsq[:, :Kin, :wAk.shape[2]] = wAk
        wp3A = torch.empty((3 * (F // 16) * K * (F // 32) * 64 * 8,), dtype=torch.bfloat16, device=dev)
        check(lib.gcrnn_fused_pack_weights_x3(_p(wAsq), _p(wAsq), _p(wp3A), F, 0, K, K, st), 'pack_weights_x3')
        z3 = ctx.xu3
        ctx.xu3 = None
        if z3 is None:
            Xf = Xp if Gp == F else torch.nn.functional.pad(Xp, (0, 0, 0, F - Gp)).contiguous()
            z3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
            check(lib.gcrnn_pack_seq_major_x3(_p(Xf), _p(z3), B, T, F, N, npad, st), 'pack_seq_x3')
            del Xf
        y3 = torch.empty((T, 3, B, npad, F), dtype=torch.bfloat16, device=dev)
        for t in range(T):
            check(lib.gcrnn_fused_filter_x3(_p(z3[t]), _p(y3[t]), _p(wp3A), *gargs, B, N, F, K, plan['uniform_w'], _p(plan.get('rank1_x3')), st), 'fused_filter_x3')
        del z3
        dgi = torch.empty((T, B), dtype=torch.float32, device=dev)
        cb = torch.empty((T, B), dtype=torch.float32, device=dev)
        check(lib.gcrnn_x3_item_dots(_p(dpre3), _p(y3), _p(bvec), _p(dgi), _p(cb), B, T, npad, F, st), 'x3_item_dots')
        del y3
        dgi = dgi + cb
        dgf = parts.sum(dim=2) + cb
        # ---- the cell's taps and bias: exact-fp32 weight gradient, the gates as operand weights ----
        slots = int(lib.gcrnn_fused_wgrad_slots(T * B, F))
        dWp = torch.zeros((slots, F, K, F + Gp), dtype=torch.float32, device=dev)
        dbp = torch.zeros((slots, F), dtype=torch.float32, device=dev)
        check(lib.gcrnn_fused_backward_weight_f32_gated(_p(dpre3), _p(Xp), _p(H), _p(h0c), _p(dWp), _p(dbp), *aargs, B, T, N, F, Gp, K,
                                                        pa['uniform_w'], _p(gi), _p(gf), 0, _p(pa.get('rank1_x3')), st), 'fused_backward_weight_f32_gated')
        dW = dWp.sum(dim=0)
        G = ctx.G
        need = ctx.needs_input_grad
        gA = dW[:, :Kin, F:F + G].unsqueeze(1).to(wA.dtype) if need[2] else None
        gB = dW[:, :Kst, :F].unsqueeze(1).to(wB.dtype) if need[3] else None
        gb = dbp.sum(dim=0).view_as(bias).to(bias.dtype) if (bias is not None and need[4]) else None
        del dWp, dbp
        # ---- the gate cells: read-out by a GEMV over the kept states; taps / bias by the weight-gradient kernel over all items, the upstream
        # gradient d logit * w * (1 - c^2) formed while it is cut into planes (dpre3's buffer is reused) ----
        out = []
        for base, dgate, g, c, (wA_g, wB_g, bias_g, lin_w, lin_b) in ((5, dgi, gi, c_i, (gA_i, gB_i, gb_i, lw_i, lb_i)),
                                                                      (10, dgf, gf, c_f, (gA_f, gB_f, gb_f, lw_f, lb_f))):
            dlogit = (dgate * g * (1.0 - g)).contiguous()                                            # [T][B]
            Kg_in, Kg_st = wA_g.shape[2], wB_g.shape[2]
            Kg = max(Kg_in, Kg_st)
            Ggp = wA_g.shape[3]
            g_lw = (dlogit.t().reshape(1, B * T) @ c.view(B * T, F * N)).view_as(lin_w).to(lin_w.dtype) if need[base + 3] else None
            g_lb = dlogit.sum().view_as(lin_b).to(lin_b.dtype) if (lin_b is not None and need[base + 4]) else None
            g_wA = g_wB = g_b = None
            if need[base] or need[base + 1] or (bias_g is not None and need[base + 2]):
                lwv = lin_w.detach().float().contiguous().view(F, N)
                check(lib.gcrnn_pack_seq_major_x3_ex(_p(c), _p(dpre3), B, T, F, N, npad, _p(dlogit), _p(lwv), 1, 0, st), 'pack_seq_x3_ex')
                Xg = Xp if Ggp == Gp else Xp[:, :, :Ggp].contiguous()
                dWs = torch.zeros((slots, F, Kg, F + Ggp), dtype=torch.float32, device=dev)
                dbs = torch.zeros((slots, F), dtype=torch.float32, device=dev)
                check(lib.gcrnn_fused_backward_weight_f32_gated(_p(dpre3), _p(Xg), None, None if ctx.hzero else _p(h0c), _p(dWs), _p(dbs), *aargs,
                                                                B, T, N, F, Ggp, Kg, pa['uniform_w'], None, None, 1, _p(pa.get('rank1_x3')), st), 'fused_backward_weight_f32_gated')
                dWg, dbg = dWs.sum(dim=0), dbs.sum(dim=0)
                g_wA = dWg[:, :Kg_in, F:F + Ggp].unsqueeze(1).to(wA_g.dtype) if need[base] else None
                g_wB = dWg[:, :Kg_st, :F].unsqueeze(1).to(wB_g.dtype) if need[base + 1] else None
                g_b = dbg.view_as(bias_g).to(bias_g.dtype) if (bias_g is not None and need[base + 2]) else None
                del dWs, dbs
            out += [g_wA, g_wB, g_b, g_lw, g_lb]
        return (None, None, gA, gB, gb, *out, None)


def fused_cell_train_x3_gated(X, h0, wA, wB, bias, graph, gates):
    """Training forward of the TIME-GATED cell at fp32 accuracy on the fused kernels (fp32 tensors and parameters).
    gates = {'in': (wA_g, wB_g, bias_g, lin_w, lin_b), 'forget': (...)} as in fused_cell_forward (live parameters: they receive gradients)."""
    require_device(X, h0, wA, wB, bias)
    gi_, gf_ = gates['in'], gates['forget']
    return _FusedTimeCellX3.apply(X, h0, wA, wB, bias, *gi_, *gf_, graph)


def _fused_pack_state_taps(w, K, st):
    """Taps of ONE filter, w [F_out][1][k][C_in] (C_in == F_out on the fused path), packed as a state-only operand (G = 0)
    with K taps (zero beyond k)."""
    F, _, k, Cin = w.shape
    assert Cin == F
    wc = w.detach().contiguous()
    wpack = torch.empty(((F // 16) * K * (F // 32) * 64 * 8,), dtype=torch.bfloat16, device=w.device)
    check(lib.gcrnn_fused_pack_weights(dtype_code(wc.dtype), _p(wc), _p(wc), _p(wpack), F, 0, K, k, st), 'pack_weights')
    return wpack


def fused_pack_upstream(dH, graph, K):
    """dH [B][T][F][N] bf16 (user layout) -> (dHs [T][B][NPad][F] sequence-major, dH_user or None). On uniform-weight graphs only the
    last two steps are packed here and dH_user = dH: the BPTT launch that consumes dHs[t-1] lays out dHs[t-2] itself (inline pack,
    gcrnn_fused_backward_data_bf16 with dHuser_inline); otherwise the whole tensor is packed and dH_user is None."""
    B, T, F, N = dH.shape
    plan = graph.fused_plan(adjoint=True)
    npad = plan['npad']
    st = _stream()
    dHs = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=dH.device)
    if fused_inline_pack_ok(plan, N, F, F, K) and dH.data_ptr() % 16 == 0:
        check(lib.gcrnn_pack_seq_major_steps(_p(dH), _p(dHs), B, T, F, N, npad, max(T - 2, 0), T, 0, st), 'pack_seq_steps')
        return dHs, dH
    check(lib.gcrnn_pack_seq_major(_lib.BF16, _p(dH), _p(dHs), B, T, F, N, npad, None, st), 'pack_seq')
    return dHs, None


def fused_backward_data(dHs, hs, wB, graph, want_dh0=True, gf=None, h0s=None, bias=None, dH_user=None):
    """BPTT data-gradient chain of the fused cell. dHs, hs: [T][B][NPad][F] bf16 sequence-major (gradient of the loss
    w.r.t. every state; the states). gf: [T][B] fp32 forget gates of the time-gated cell or None.
    Returns (dpre [T][B][NPad][F] bf16, dh0 [B][NPad][F] bf16 or None), and with h0s ([1][B][NPad][F], the initial state)
    also d loss / d gf [T][B] fp32 = <B(S) h_{t-1} + bias, dpre_t>: its filter part is <h_{t-1}, adjoint chain of dpre_t>,
    read off the chain these launches evaluate anyway; the bias part is a column sum of dpre."""
    T, B, npad, F = hs.shape
    K = wB.shape[2]
    plan = graph.fused_plan(adjoint=True)
    st = _stream()
    wBt = wB.detach()[:, 0].permute(2, 1, 0).contiguous().unsqueeze(1)      # [F_in][1][K][F_out]: transposed taps
    dpre = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=hs.device)
    dh0 = torch.empty((B, npad, F), dtype=torch.bfloat16, device=hs.device) if want_dh0 else None
    import os
    plan16 = None if os.environ.get('GCRNN_NO_IMG16') else graph.fused_plan_img16(adjoint=True)      # bf16 hop image, matrix-core sums (uniform graphs)
    pw = plan16
    if pw is None and not os.environ.get('GCRNN_NO_IMG16'):
        pw = graph.fused_plan_rank1(adjoint=True)        # rank-1-weighted graph (normalised adjacency): the adjoint plan of its pattern + the swapped factors
    if pw is not None and F % 32 == 0 and not os.environ.get('GCRNN_NO_WIDE_CHAIN') and lib.gcrnn_fused_backward_data_wide_supported(
            B, T, graph.N, F, K, int(pw['entries']), float(pw.get('uniform_w', 0.0)), 3 if pw.get('rank1') else 1, 1 if dH_user is not None else 0):
        plan16 = pw
        # the whole chain (seed, T - 1 steps, d h0 / the forget gate's step 0) as ONE launch of the wide sequence-resident kernel
        wpw = _fused_pack_weights_wide(wBt.new_zeros((F, 1, K, 0)), wBt, plan16['uniform_w'], st)
        parts = torch.empty((T * B, (F // 32) * int(lib.gcrnn_fused_step_waves())), dtype=torch.float32, device=hs.device) if h0s is not None else None
        check(lib.gcrnn_fused_backward_data_wide_bf16(_p(dHs), _p(hs), _p(dpre), _p(dh0), _p(wpw), _p(plan16['tile_slots']), _p(plan16['tile_off']),
                                                      _p(plan16['ell_col4']), plan16['entries'], B, T, graph.N, F, K, _p(gf), _p(h0s), _p(parts),
                                                      _p(dH_user), _p(plan16.get('rank1_a')), _p(plan16.get('rank1_b')), st), 'fused_backward_data_wide')
        if h0s is None:
            return dpre, dh0
        dgf = parts.sum(dim=1).view(T, B)
        if bias is not None:
            dgf = dgf + dpre.sum(dim=2, dtype=torch.float32) @ bias.detach().float().view(-1)
        return dpre, dh0, dgf
    wpack = _fused_pack_state_taps(wBt, K, st)
    parts = None
    if h0s is not None:
        parts = torch.empty((T * B, (F // 16) * int(lib.gcrnn_fused_step_waves())), dtype=torch.float32, device=hs.device)
    check(lib.gcrnn_fused_backward_data_bf16(_p(dHs), _p(hs), _p(dpre), _p(dh0), _p(wpack), *_fused_graph_args(plan16 or plan),
                                             B, T, graph.N, F, K, _p(gf), _p(h0s), _p(parts), plan.get('uniform_w', 0.0), _p(dH_user),
                                             1 if plan16 else 0, st),
          'fused_backward_data')
    if h0s is None:
        return dpre, dh0
    dgf = parts.sum(dim=1).view(T, B)
    if bias is not None:
        dgf = dgf + dpre.sum(dim=2, dtype=torch.float32) @ bias.detach().float().view(-1)
    return dpre, dh0, dgf


def fused_backward_weight(dpre, X, H, h0, graph, F, G, K, want_bias=False, gi=None, gf=None, h_is_h0=False, hzero=None):
    """dW [F][K][F+G] fp32 (columns: F state features, then G input features) from dpre (sequence-major bf16) and the
    user-layout bf16 tensors X [B][T][G][N], H [B][T][F][N] (forward output), h0 [B][F][N]. gi / gf [T][B] fp32: item
    (t, b) enters the input-filter columns with weight gi and the state-filter columns with gf (time-gated cell).
    h_is_h0: every item's state operand is h0 (gate sub-cells; H may be None).
    With want_bias also returns the bias gradient [F]: sum_{t,b} (gi + gf) sum_n dpre (the one bias enters both filters)."""
    T, B = dpre.shape[0], dpre.shape[1]
    # the kernel addresses dpre with 32-bit byte offsets: batches whose dpre passes 2 GiB (B = 512 at T = 32, F = 64) run as batch chunks -- sequences
    # are independent, the chunks' sums add (in chunk order: still bit-reproducible); the chunk of dpre is copied once to make it contiguous
    lim = (2 ** 31 - 1) // (T * dpre.shape[2] * F * 2)
    if B > lim >= 1:
        step = max(256 * (lim // 256), 1) if lim >= 256 else lim
        dW = dbs = None
        for b0 in range(0, B, step):
            b1 = min(B, b0 + step)
            r = fused_backward_weight(dpre[:, b0:b1].contiguous(), X[b0:b1], H[b0:b1] if H is not None else None, h0[b0:b1], graph, F, G, K, want_bias,
                                      gi[:, b0:b1].contiguous() if gi is not None else None, gf[:, b0:b1].contiguous() if gf is not None else None, h_is_h0, hzero)
            w, bsum = r if want_bias else (r, None)
            dW = w if dW is None else dW + w
            if want_bias:
                dbs = bsum if dbs is None else dbs + bsum
        return (dW, dbs) if want_bias else dW
    plan = graph.fused_plan(adjoint=True, kernel='wgrad')
    Xc, h0c = X.contiguous(), h0.contiguous()
    Hc = H.contiguous() if H is not None else None
    uw = 0.0 if os.environ.get('GCRNN_WGRAD_NO_UNIFORM') else plan.get('uniform_w', 0.0)      # env: A/B switch
    plan16 = None if (uw == 0.0 or os.environ.get('GCRNN_NO_IMG16')) else graph.fused_plan_img16(adjoint=True)      # bf16 hop image
    if plan16 is None and uw == 0.0 and not os.environ.get('GCRNN_NO_IMG16') and not os.environ.get('GCRNN_WGRAD_NO_UNIFORM'):
        plan16 = graph.fused_plan_rank1(adjoint=True)    # rank-1-weighted graph: the adjoint plan of its 0/1 pattern + the factors of S^T (kernel variant R1)
        if plan16 is not None:
            uw = 1.0
    pl = plan16 or plan
    slots = int(lib.gcrnn_fused_wgrad_bf16_slots(T * B, F, K, pl['entries'], 1 if plan16 else 0))
    dWp = torch.zeros((slots, F, K, F + G), dtype=torch.float32, device=dpre.device)      # per-slot partial sums (plain stores)
    dbp = torch.zeros((slots, F), dtype=torch.float32, device=dpre.device) if want_bias else None
    check(lib.gcrnn_fused_backward_weight_bf16(_p(dpre), _p(Xc), _p(Hc), _p(h0c), _p(dWp),
                                               _p(dbp), _p(pl['tile_slots']), _p(pl['tile_off']), _p(pl['ell_val4']),
                                               _p(pl['ell_col4']), pl['entries'], B, T, graph.N, F, G, K,
                                               _p(gi), _p(gf), int(h_is_h0) | (2 if plan16 else 0), _p(hzero), uw,
                                               _p(plan16.get('rank1_a')) if plan16 else None, _p(plan16.get('rank1_b')) if plan16 else None, _stream()),
          'fused_backward_weight')
    dW = dWp.sum(dim=0)                                   # fixed order over the slots: bit-reproducible
    dbs = dbp.sum(dim=0) if want_bias else None
    return (dW, dbs) if want_bias else dW


def fused_gate_grad(zs, dpre, w, bias, graph, K):
    """d loss / d gate [T][B] fp32 of one filter of the time-gated cell: sum_{f,n} (w(S) z + bias) . dpre per item.
    zs: that filter's operand [T][B][NPad][C] bf16 sequence-major (h_{t-1}, or x_t), dpre: [T][B][NPad][F]; w: its taps
    F x 1 x k x C; bias F x 1 or None. C == F runs the operand alone; an input filter with C != F runs as [0 | x_t]."""
    T, B, npad, F = dpre.shape
    Cin = w.shape[3]
    plan = graph.fused_plan()
    plan16 = fused_img16_plan(graph, True, None)
    st = _stream()
    b32 = bias.detach().float().contiguous().view(-1) if bias is not None else None
    parts = torch.empty((T * B, (F // 16) * int(lib.gcrnn_fused_step_waves())), dtype=torch.float32, device=dpre.device)
    if Cin == F:
        wp = _fused_pack_state_taps(w, K, st)
        check(lib.gcrnn_fused_gate_grad_bf16(_p(zs), None, _p(dpre), _p(wp), _p(b32), _p(parts), *_fused_graph_args(plan16 or plan),
                                             B, T, graph.N, F, 0, K, plan.get('uniform_w', 0.0), 1 if plan16 else 0, st), 'fused_gate_grad')
    else:
        wd = w.detach()
        wz = wd.new_zeros((F, 1, K, F))
        if wd.shape[2] < K:
            wd = torch.cat([wd, wd.new_zeros(F, 1, K - wd.shape[2], Cin)], dim=2)
        wp = _fused_pack_weights(wd, wz, st)
        zero_h = torch.zeros((1, npad, F), dtype=torch.bfloat16, device=dpre.device)
        check(lib.gcrnn_fused_gate_grad_bf16(_p(zero_h), _p(zs), _p(dpre), _p(wp), _p(b32), _p(parts), *_fused_graph_args(plan16 or plan),
                                             B, T, graph.N, F, Cin, K, plan.get('uniform_w', 0.0), 1 if plan16 else 0, st), 'fused_gate_grad')
    return parts.sum(dim=1).view(T, B)


def fused_training_supported(graph, N, F, G, Kin, Kst, E=1):
    """The fused BPTT needs the forward kernel's shapes, node-contiguous rows that are 16-byte aligned (N % 8 == 0)
    and the adjoint graph image next to the state and the transposed tile in LDS."""
    Gp = fused_padded_inputs(F, G)
    if E != 1 or N % 8 != 0 or Gp is None or not bool(lib.gcrnn_fused_supported(int(N), int(F), int(Gp), int(max(Kin, Kst)))):
        return False
    entries = graph.fused_plan(adjoint=True)['entries']
    return 65536 + 96 * entries + 16 * 1056 + 8 * 16 * 4 <= 160 * 1024


class _FusedTimeGate(torch.autograd.Function):
    """One time gate of the fused path with its BPTT (reference graphML.py:2357-2374): gate[t][b] = sigmoid(lin(vec(c_t))),
    c_t = tanh(A_g(S) x_t + B_g(S) h0 + 2 b_g).

    forward  = ONE pre-pass launch over all (t, b) that also stores c (bf16);
    backward = one in-place pass over c (read-out gradient + dpre_g = dlogit w (1 - c^2)), then the weight-gradient kernel
               over all items with h0 as every item's state operand. No gradient for X or h0 (the training loops start
               from h0 = 0 and never ask for either, train_rnn.py:247-276)."""

    @staticmethod
    def forward(ctx, xs, h0s, X, h0, wA_g, wB_g, bias_g, lin_w, lin_b, graph, hzero):
        N = X.shape[3]
        gate, cs, gw = fused_time_gate(xs, h0s, wA_g, wB_g, bias_g, lin_w, lin_b, graph, N, store_states=True, hzero=hzero)
        ctx.save_for_backward(X, h0, wA_g, wB_g, bias_g, lin_w, lin_b, gate, cs, gw, hzero)
        ctx.graph = graph
        return gate

    @staticmethod
    def backward(ctx, dgate):
        X, h0, wA_g, wB_g, bias_g, lin_w, lin_b, gate, cs, gw, hzero = ctx.saved_tensors
        if getattr(ctx, 'consumed', False):
            raise GcrnnError('the fused time gate was already back-propagated: its saved states are turned into gradients in '
                             'place, so a second backward over the same graph (retain_graph) is not supported')
        ctx.consumed = True
        wx, wh = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        if (wx or wh) and not fused_input_grad_ok(wA_g.shape[0], X.shape[2]):
            raise GcrnnError('the fused time gate produces gradients w.r.t. X or h0 only for G == F')
        res = _time_gate_backward(X, h0, wA_g, wB_g, bias_g, lin_w, lin_b, gate, cs, gw, hzero, ctx.graph, dgate, ctx.needs_input_grad[4:9], wx, wh)
        if wx or wh:
            return (None, None, res[0], res[1]) + res[2:] + (None, None)
        return (None, None, None, None) + res + (None, None)


def _time_gate_input_grads(X, h0, wA_g, wB_g, dpre_g, graph, want_x, want_h0):
    """d loss / d X and d loss / d h0 THROUGH a time gate (round 4; the reference's autograd gives both, graphML.py:2362, 2370: the gate cell
    reads (x_t, h0)): the gate cell's filters' adjoints on its dpre_g -- dX_t = sum_k (S^T)^k (dpre_g,t A_g,k), d h0 = sum_t sum_k (S^T)^k
    (dpre_g,t B_g,k) -- each ONE all-items launch of the filter-output pass on the adjoint graph with the transposed taps (G == F:
    fused_input_grad_ok). dpre_g [T][B][NPad][F] bf16. Returns (gX [B][T][G][N], gh0 [B][F][N]) in X's / h0's dtype, or None."""
    B, T, G, N = X.shape
    F, Kin, Kst = wA_g.shape[0], wA_g.shape[2], wB_g.shape[2]
    K = max(Kin, Kst)
    npad = dpre_g.shape[2]
    st = _stream()
    gX = gh0 = None
    if want_x:
        wAk = wA_g if Kin == K else torch.cat([wA_g, wA_g.new_zeros(F, 1, K - Kin, G)], dim=2)
        wAt = wAk.detach()[:, 0].permute(2, 1, 0).contiguous().unsqueeze(1)       # [G][1][K][F]: transposed taps
        dxs = fused_filter_output(dpre_g, wAt, None, graph, K, N, adjoint=True)   # [T][B][NPad][G] bf16
        gX = torch.empty((B, T, G, N), dtype=torch.bfloat16, device=X.device)
        check(lib.gcrnn_unpack_seq_major(_lib.BF16, _p(dxs), _p(gX), B, T, G, N, npad, None, st), 'unpack_seq')
        gX = gX.to(X.dtype)
    if want_h0:
        wBk = wB_g if Kst == K else torch.cat([wB_g, wB_g.new_zeros(F, 1, K - Kst, F)], dim=2)
        wBt = wBk.detach()[:, 0].permute(2, 1, 0).contiguous().unsqueeze(1)
        dhs = fused_filter_output(dpre_g, wBt, None, graph, K, N, adjoint=True)   # every item's contribution to its sequence's h0
        dh = dhs.float().sum(dim=0).to(torch.bfloat16).unsqueeze(0).contiguous()  # [1][B][NPad][F]: fp32 sum over the T items of a sequence
        gh0 = torch.empty((B, 1, F, N), dtype=torch.bfloat16, device=X.device)
        check(lib.gcrnn_unpack_seq_major(_lib.BF16, _p(dh), _p(gh0), B, 1, F, N, npad, None, st), 'unpack_seq')
        gh0 = gh0.view(B, F, N).to(h0.dtype)
    return gX, gh0


def _time_gate_backward(X, h0, wA_g, wB_g, bias_g, lin_w, lin_b, gate, cs, gw, hzero, graph, dgate, needs, want_x=False, want_h0=False):
    """BPTT of one time gate from its stored sub-cell states cs (turned into dpre_g IN PLACE): read-out gradient, then the weight-gradient
    kernel over all items with h0 as every item's state operand. Returns (gA, gB, gb, glw, glb); needs = which of them are wanted.
    want_x / want_h0: two more results in front, (gX, gh0) -- the gradients the gate hands to its inputs."""
    B, T, G, N = X.shape
    F, Kin, Kst = wA_g.shape[0], wA_g.shape[2], wB_g.shape[2]
    K = max(Kin, Kst)
    T_, B_, npad, _ = cs.shape
    items = T * B
    dlogit = (dgate.float() * gate * (1.0 - gate)).contiguous().view(-1)          # through the sigmoid
    slabs = int(lib.gcrnn_fused_gate_readout_slabs(items))
    dw_part = torch.empty((slabs, npad * F), dtype=torch.float32, device=X.device)
    # cs becomes dpre_g in place: a second backward through the same graph is not supported (retain_graph)
    check(lib.gcrnn_fused_gate_readout_backward_bf16(_p(cs), _p(dlogit), _p(gw), _p(dw_part), items, N, F, _stream()),
          'gate_readout_backward')
    dW, dbs = fused_backward_weight(cs, X, None, h0, graph, F, G, K, want_bias=True, h_is_h0=True, hzero=hzero)
    gA = dW[:, :Kin, F:].unsqueeze(1).to(wA_g.dtype) if needs[0] else None
    gB = dW[:, :Kst, :F].unsqueeze(1).to(wB_g.dtype) if needs[1] else None
    gb = dbs.view_as(bias_g).to(bias_g.dtype) if (bias_g is not None and needs[2]) else None
    glw = None
    if needs[3]:
        glw = dw_part.sum(dim=0).view(npad, F)[:N].t().reshape(1, F * N).to(lin_w.dtype)     # [N][F] -> vec over (f, n)
    glb = dlogit.sum().view(1).to(lin_b.dtype) if (lin_b is not None and needs[4]) else None
    if want_x or want_h0:
        return _time_gate_input_grads(X, h0, wA_g, wB_g, cs, graph, want_x, want_h0) + (gA, gB, gb, glw, glb)
    return gA, gB, gb, glw, glb


class _FusedTimeGatePair(torch.autograd.Function):
    """BOTH time gates of the fused path with their BPTT: forward = ONE pre-pass launch of the wide sequence-resident kernel over all (t, b)
    (fused_time_gate_pair: the two sub-cells as one cell of 2 F outputs; states of both stored), backward = _FusedTimeGate's, once per gate."""

    @staticmethod
    def forward(ctx, xs, h0s, X, h0, wA_i, wB_i, b_i, lw_i, lb_i, wA_f, wB_f, b_f, lw_f, lb_f, graph, hzero):
        N = X.shape[3]
        gi, gf, (cs_i, gw_i), (cs_f, gw_f) = fused_time_gate_pair(xs, h0s, (wA_i, wB_i, b_i, lw_i, lb_i), (wA_f, wB_f, b_f, lw_f, lb_f), graph, N,
                                                                  store_states=True, hzero=hzero)
        ctx.save_for_backward(X, h0, wA_i, wB_i, b_i, lw_i, lb_i, wA_f, wB_f, b_f, lw_f, lb_f, gi, gf, cs_i, gw_i, cs_f, gw_f, hzero)
        ctx.graph = graph
        return gi, gf

    @staticmethod
    def backward(ctx, dgi, dgf):
        X, h0, wA_i, wB_i, b_i, lw_i, lb_i, wA_f, wB_f, b_f, lw_f, lb_f, gi, gf, cs_i, gw_i, cs_f, gw_f, hzero = ctx.saved_tensors
        if getattr(ctx, 'consumed', False):
            raise GcrnnError('the fused time gates were already back-propagated: their saved states are turned into gradients in '
                             'place, so a second backward over the same graph (retain_graph) is not supported')
        ctx.consumed = True
        wx, wh = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        if (wx or wh) and not fused_input_grad_ok(wA_i.shape[0], X.shape[2]):
            raise GcrnnError('the fused time gates produce gradients w.r.t. X or h0 only for G == F')
        zi = torch.zeros_like(gi) if dgi is None else dgi
        zf = torch.zeros_like(gf) if dgf is None else dgf
        g_in = _time_gate_backward(X, h0, wA_i, wB_i, b_i, lw_i, lb_i, gi, cs_i, gw_i, hzero, ctx.graph, zi, ctx.needs_input_grad[4:9], wx, wh)
        g_f = _time_gate_backward(X, h0, wA_f, wB_f, b_f, lw_f, lb_f, gf, cs_f, gw_f, hzero, ctx.graph, zf, ctx.needs_input_grad[9:14], wx, wh)
        gX = gh0 = None
        if wx or wh:
            gX = (g_in[0] + g_f[0]) if wx else None
            gh0 = (g_in[1] + g_f[1]) if wh else None
            g_in, g_f = g_in[2:], g_f[2:]
        return (None, None, gX, gh0) + g_in + g_f + (None, None)


def fused_train_time_gates(xs, h0s, X, h0, gates, graph, hzero):
    """The two differentiable time gates [T][B] of a training step: ONE pre-pass launch for the pair where the wide kernel takes the problem
    (uniform-weight graph, a batch that fills the chip), else one launch per gate."""
    T, B, npad, G = xs.shape
    N = X.shape[3]
    gin, gfo = gates['in'], gates['forget']
    F = gin[0].shape[0]
    K = max(gin[0].shape[2], gin[1].shape[2])
    pair16 = None
    if gin[0].shape == gfo[0].shape and gin[1].shape == gfo[1].shape and gin[0].shape[3] == G and not os.environ.get('GCRNN_NO_GATE_PAIR_TRAIN'):
        pair16, _ = fused_gate_pair_plan(graph, B, T, N, F, G, K, getattr(xs, '_pending_user', None) is not None)
    if pair16 is not None:
        return _FusedTimeGatePair.apply(xs, h0s, X, h0, *gin, *gfo, graph, hzero)
    gi = _FusedTimeGate.apply(xs, h0s, X, h0, *gin, graph, hzero)
    gf = _FusedTimeGate.apply(xs, h0s, X, h0, *gfo, graph, hzero)
    return gi, gf


class _FusedCell(torch.autograd.Function):
    """GGCRNNCell (un-gated, or time-gated with the gates as differentiable inputs) on the fused kernels, forward and BPTT
    (bf16 activations, fp32 or bf16 parameters).

    backward = pack(dH) -> data-gradient chain (T launches of the step kernel on the adjoint graph with transposed taps,
    the recurrent part scaled by the forget gate) -> ONE weight-gradient launch over all T*B items (item weights gi / gf)
    -> with gates:  d gf = <B(S)h_{t-1} + b, dpre_t> = <h_{t-1}, adjoint chain of dpre_t> + b . colsum(dpre_t) falls out of
    the data-gradient launches (adjoint identity); d gi = <A(S)x_t + b, dpre_t> is one gate-gradient pass over all items.
    The gradient w.r.t. X (no training loop of the reference asks for it, train_rnn.py:247-276, but its autograd gives it) is the
    input filter's adjoint on dpre, gi_t (S^T)^k (dpre_t A_k): ONE all-items launch of the filter-output pass on the adjoint graph with
    the transposed input taps (G == F; ops.fused_input_grad_ok)."""

    @staticmethod
    def forward(ctx, X, h0, wA, wB, bias, gi, gf, graph, xs, hs_all):
        gv = (gi, gf) if gi is not None else None
        hs_all, plan, H = fused_cell_forward(X, h0, wA, wB, bias, graph, return_states=True, gate_values=gv,
                                             packed=(xs, hs_all) if xs is not None else None)
        ctx.save_for_backward(X, h0, wA, wB, bias, H, hs_all, gi, gf, xs)
        ctx.graph = graph
        ctx.npad = plan['npad']
        return H

    @staticmethod
    def backward(ctx, dH):
        X, h0, wA, wB, bias, H, hs_all, gi, gf, xs = ctx.saved_tensors
        graph, npad = ctx.graph, ctx.npad
        B, T, G, N = X.shape
        F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
        K = max(Kin, Kst)
        if ctx.needs_input_grad[0] and not fused_input_grad_ok(F, G):
            raise GcrnnError('the fused BPTT produces the gradient w.r.t. the input sequence X only for G == F')
        st = _stream()
        hs = hs_all[1:]
        gated = gi is not None
        if gated:
            gi, gf = gi.detach().float().contiguous(), gf.detach().float().contiguous()
        dH = dH.to(torch.bfloat16).contiguous()
        dHs, dHu = fused_pack_upstream(dH, graph, K)
        wBk = wB if Kst == K else torch.cat([wB, wB.new_zeros(F, 1, K - Kst, F)], dim=2)
        dgf = None
        if gated and ctx.needs_input_grad[6]:
            dpre, dh0s, dgf = fused_backward_data(dHs, hs, wBk, graph, want_dh0=ctx.needs_input_grad[1], gf=gf,
                                                  h0s=hs_all[:1], bias=bias, dH_user=dHu)
        else:
            dpre, dh0s = fused_backward_data(dHs, hs, wBk, graph, want_dh0=ctx.needs_input_grad[1], gf=gf if gated else None, dH_user=dHu)
        want_b = bias is not None and ctx.needs_input_grad[4]
        dW, dbs = fused_backward_weight(dpre, X, H, h0, graph, F, G, K, want_bias=True,
                                        gi=gi if gated else None, gf=gf if gated else None)       # [F][K][F+G], [F] fp32
        gA = dW[:, :Kin, F:].unsqueeze(1).to(wA.dtype) if ctx.needs_input_grad[2] else None
        gB = dW[:, :Kst, :F].unsqueeze(1).to(wB.dtype) if ctx.needs_input_grad[3] else None
        gb = dbs.view_as(bias).to(bias.dtype) if want_b else None
        dgi = None
        if gated and ctx.needs_input_grad[5]:
            xsq = xs if xs is not None else fused_pack_inputs(X, h0, graph)[0]
            dgi = fused_gate_grad(xsq, dpre, wA, bias, graph, K)
        gh0 = None
        if ctx.needs_input_grad[1]:
            gh0 = torch.empty((B, 1, F, N), dtype=torch.bfloat16, device=X.device)
            check(lib.gcrnn_unpack_seq_major(_lib.BF16, _p(dh0s), _p(gh0), B, 1, F, N, npad, None, st), 'unpack_seq')
            gh0 = gh0.view(B, F, N).to(h0.dtype)
        gX = None
        if ctx.needs_input_grad[0]:
            # dX_t = gi_t sum_k (dpre_t A_k) shifted k times by S^T: the input filter's adjoint, every item in one launch
            wAk = wA if Kin == K else torch.cat([wA, wA.new_zeros(F, 1, K - Kin, G)], dim=2)
            wAt = wAk.detach()[:, 0].permute(2, 1, 0).contiguous().unsqueeze(1)       # [G][1][K][F]: transposed taps
            dxs = fused_filter_output(dpre, wAt, None, graph, K, N, adjoint=True)     # [T][B][NPad][G] bf16
            gX = torch.empty((B, T, G, N), dtype=torch.bfloat16, device=X.device)
            check(lib.gcrnn_unpack_seq_major(_lib.BF16, _p(dxs), _p(gX), B, T, G, N, npad, None, st), 'unpack_seq')
            if gated:
                gX = gX * gi.t().reshape(B, T, 1, 1).to(gX.dtype)
            gX = gX.to(X.dtype)
        return gX, gh0, gA, gB, gb, dgi, dgf, None, None, None


def fused_input_grad_ok(F, G):
    """d loss / d X on the fused path: the adjoint of the input filter runs as a square (state-like) operand of the filter-output pass."""
    return int(F) == int(G)


def fused_cell_train(X, h0, wA, wB, bias, graph, gates=None):
    """Training forward of the fused cell. gates: None, or the time-gate sub-networks {'in': (wA_g, wB_g, bias_g, lin_w,
    lin_b), 'forget': (...)} (graphML.py:2248-2278): both gates are evaluated (and back-propagated) on the fused kernels and
    enter the cell as differentiable [T][B] inputs."""
    require_device(X, h0, wA, wB, bias)
    if gates is None:
        return _FusedCell.apply(X, h0, wA, wB, bias, None, None, graph, None, None)
    with torch.no_grad():
        xs, hs_all = fused_pack_inputs_gated(X.contiguous(), h0, graph, wA.shape[0], max(wA.shape[2], wB.shape[2]))
        hzero = fused_h0_zero_flag(h0)
    gi, gf = fused_train_time_gates(xs, hs_all[:1], X, h0, gates, graph, hzero)
    assert getattr(xs, '_pending_user', None) is None          # (the first pre-pass laid out the rest of X)
    return _FusedCell.apply(X, h0, wA, wB, bias, gi, gf, graph, xs, hs_all)


def fused_cell_train_with_gates(X, h0, wA, wB, bias, graph, gi, gf):
    """Training forward of the time-gated fused cell with the gates [T][B] computed elsewhere (any autograd graph)."""
    require_device(X, h0, wA, wB, bias, gi, gf)
    return _FusedCell.apply(X, h0, wA, wB, bias, gi, gf, graph, None, None)


# ------------------------------------------------------------------------------------------ small-graph persistent path
def small_supported(N, nnz, G, F, Kin, Kst, dtype, E=1):
    if E != 1 or dtype not in (torch.float32, torch.float64):
        return False
    return bool(lib.gcrnn_small_supported(dtype_code(dtype), int(N), int(nnz), int(G), int(F), int(Kin), int(Kst)))


def small_dense_supported(N, G, F, Kin, Kst, dtype, backward, gated):
    if dtype not in (torch.float32, torch.float64) or os.environ.get('GCRNN_SMALL_GATHER'):   # env: A/B against the gather kernels
        return False
    return bool(lib.gcrnn_small_dense_supported(dtype_code(dtype), int(N), int(G), int(F), int(Kin), int(Kst),
                                                int(backward), int(gated)))


def _gate_strides(g, B, T, N):
    """Element strides (b, t, n) with which the matrix-core kernels read a gate tensor: [T][B] scalar gates or [B][T][N]."""
    if g is None:
        return 0, 0, 0
    if g.dim() == 2:
        assert tuple(g.shape) == (T, B)
        return 1, B, 0
    assert tuple(g.shape) == (B, T, N)
    return T * N, N, 1


def small_cell_forward(X, h0, wA, wB, bias, graph, gi=None, gf=None):
    """Whole recurrence in one launch, one workgroup per sequence (small graphs). X: B x T x G x N, h0: B x F x N
    (user layout, fp32 / fp64) -> H: B x T x F x N. gi / gf: gates of the input / state filter or None: [T][B] scalars per
    sequence and step (time gating) or [B][T][N] per node (node gating, or the product of both; matrix-core kernels only).
    Inference only."""
    require_device(X, h0, wA, wB, bias)
    B, T, G, N = X.shape
    F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
    csr = graph.fwd[0]
    H = torch.empty((B, T, F, N), dtype=X.dtype, device=X.device)
    bvec = bias.detach().contiguous().view(-1) if bias is not None else None
    if gi is not None:
        gi, gf = gi.to(X.dtype).contiguous(), gf.to(X.dtype).contiguous()
    Xc, h0c, wAc, wBc, vals = X.contiguous(), h0.contiguous(), wA.contiguous(), wB.contiguous(), csr.val(X.dtype)
    if small_dense_supported(N, G, F, Kin, Kst, X.dtype, backward=False, gated=gi is not None):
        Sd = graph.dense(X.dtype)
        gs = _gate_strides(gi, B, T, N)
        assert gs == _gate_strides(gf, B, T, N)
        check(lib.gcrnn_small_dense_forward(dtype_code(X.dtype), _p(Xc), _p(h0c), _p(wAc), _p(wBc), _p(bvec), _p(gi), _p(gf),
                                            _p(Sd), _p(H), B, T, N, G, F, Kin, Kst, gs[0], gs[1], gs[2], _stream()),
              'small_dense_forward')
        return H
    assert gi is None or gi.dim() == 2, 'per-node gates need the matrix-core kernels (small_dense_supported)'
    check(lib.gcrnn_small_forward(dtype_code(X.dtype), _p(Xc), _p(h0c), _p(wAc),
                                  _p(wBc), _p(bvec), _p(gi), _p(gf), _p(csr.rowptr), _p(csr.col),
                                  _p(vals), _p(H), B, T, N, G, F, Kin, Kst, csr.nnz, _stream()),
          'small_forward')
    return H


def small_training_supported(N, nnz, G, F, Kin, Kst, dtype, E=1):
    return small_supported(N, nnz, G, F, Kin, Kst, dtype, E) and bool(
        lib.gcrnn_small_backward_supported(dtype_code(dtype), int(N), int(nnz), int(G), int(F), int(Kin), int(Kst)))


class _SmallCell(torch.autograd.Function):
    """Small-graph cell with BPTT: forward = one launch (small_cell_kernel), backward = one launch
    (small_cell_bwd_kernel). Gradients for the parameters, the time gates and h0; none for X."""

    @staticmethod
    def forward(ctx, X, h0, wA, wB, bias, gi, gf, graph):
        X, h0 = X.contiguous(), h0.contiguous()
        H = small_cell_forward(X, h0, wA.detach(), wB.detach(), bias.detach() if bias is not None else None, graph,
                               gi.detach() if gi is not None else None, gf.detach() if gf is not None else None)
        ctx.save_for_backward(X, h0, H, wA, wB, bias, gi, gf)
        ctx.graph = graph
        return H

    @staticmethod
    def backward(ctx, dH):
        X, h0, H, wA, wB, bias, gi, gf = ctx.saved_tensors
        B, T, G, N = X.shape
        F, Kin, Kst = wA.shape[0], wA.shape[2], wB.shape[2]
        dt, dev = X.dtype, X.device
        fwd, adj = ctx.graph.fwd[0], ctx.graph.adj[0]
        dense = small_dense_supported(N, G, F, Kin, Kst, dt, backward=True, gated=gi is not None)
        pA = torch.empty((B, 1 if dense else 2, F, Kin, G), dtype=dt, device=dev)
        pB = torch.empty((B, 1 if dense else 2, F, Kst, F), dtype=dt, device=dev)
        pb = torch.empty((B, F), dtype=dt, device=dev)
        dgi = dgf = None
        scalar_gates = gi is not None and gi.dim() == 2
        if gi is not None:
            assert dense or scalar_gates, 'per-node gates need the matrix-core kernels (small_dense_supported)'
            gi, gf = gi.to(dt).contiguous(), gf.to(dt).contiguous()
            shape = (B, T, N) if dense else (T, B)
            dgi = torch.empty(shape, dtype=dt, device=dev)
            dgf = torch.empty(shape, dtype=dt, device=dev)
        dh0 = torch.empty_like(h0) if ctx.needs_input_grad[1] else None
        bvec = bias.detach().contiguous().view(-1) if bias is not None else None
        # every operand is a named local: a temporary (dH.contiguous(), a first-use adj.val(dt)) would be released --
        # and its block handed to the next allocation -- before the launch
        dHc, wAc, wBc, fval, aval = dH.contiguous(), wA.contiguous(), wB.contiguous(), fwd.val(dt), adj.val(dt)
        if dense:
            Sd = ctx.graph.dense(dt)
            gs = _gate_strides(gi, B, T, N)
            check(lib.gcrnn_small_dense_backward(dtype_code(dt), _p(X), _p(h0), _p(H), _p(dHc), _p(wAc), _p(wBc), _p(bvec),
                                                 _p(gi), _p(gf), _p(Sd), _p(pA), _p(pB), _p(pb), _p(dgi), _p(dgf), _p(dh0),
                                                 B, T, N, G, F, Kin, Kst, gs[0], gs[1], gs[2], _stream()), 'small_dense_backward')
            if scalar_gates:                                  # a scalar gate collects the gradients of all its nodes
                dgi, dgf = dgi.sum(dim=2).t(), dgf.sum(dim=2).t()
        else:
            check(lib.gcrnn_small_backward(dtype_code(dt), _p(X), _p(h0), _p(H), _p(dHc), _p(wAc),
                                           _p(wBc), _p(bvec), _p(gi), _p(gf), _p(fwd.rowptr), _p(fwd.col),
                                           _p(fval), _p(adj.rowptr), _p(adj.col), _p(aval), _p(pA), _p(pB), _p(pb),
                                           _p(dgi), _p(dgf), _p(dh0), B, T, N, G, F, Kin, Kst, fwd.nnz, _stream()),
                  'small_backward')
        dwA = pA.sum(dim=(0, 1)).view(F, 1, Kin, G)
        dwB = pB.sum(dim=(0, 1)).view(F, 1, Kst, F)
        db = pb.sum(dim=0).view(F, 1) if bias is not None else None
        return None, dh0, dwA, dwB, db, dgi, dgf, None


def small_cell_train(X, h0, wA, wB, bias, graph, gi=None, gf=None):
    return _SmallCell.apply(X, h0, wA, wB, bias, gi, gf, graph)


def small_gates_supported(N, G, F, Kin, Kst, dtype, backward):
    if dtype not in (torch.float32, torch.float64) or os.environ.get('GCRNN_SMALL_GATHER'):
        return False
    return bool(lib.gcrnn_small_gates_supported(dtype_code(dtype), int(N), int(G), int(F), int(Kin), int(Kst), int(backward)))


class _SmallTimeGates(torch.autograd.Function):
    """Both time gates of a small-graph cell for all steps: one launch forward, one backward (reference
    graphML.py:2357-2374). Parameters stacked over (input, forget). Returns gates [2][T][B]. No gradient for X."""

    @staticmethod
    def forward(ctx, X, h0, wA2, wB2, bias2, lw2, lb2, graph):
        X, h0, wA2, wB2, lw2 = X.contiguous(), h0.contiguous(), wA2.contiguous(), wB2.contiguous(), lw2.contiguous()
        bias2 = bias2.contiguous() if bias2 is not None else None
        lb2 = lb2.contiguous() if lb2 is not None else None
        B, T, G, N = X.shape
        F, Kin, Kst = wA2.shape[1], wA2.shape[2], wB2.shape[2]
        Sd = graph.dense(X.dtype)
        gates = torch.empty((2, T, B), dtype=X.dtype, device=X.device)
        check(lib.gcrnn_small_gates_forward(dtype_code(X.dtype), _p(X), _p(h0), _p(wA2), _p(wB2), _p(bias2), _p(lw2), _p(lb2),
                                            _p(Sd), _p(gates), B, T, N, G, F, Kin, Kst, _stream()), 'small_gates_forward')
        ctx.save_for_backward(X, h0, wA2, wB2, bias2, lw2, gates)
        ctx.graph, ctx.has_lb = graph, lb2 is not None
        return gates

    @staticmethod
    def backward(ctx, dgates):
        X, h0, wA2, wB2, bias2, lw2, gates = ctx.saved_tensors
        B, T, G, N = X.shape
        F, Kin, Kst = wA2.shape[1], wA2.shape[2], wB2.shape[2]
        dt, dev = X.dtype, X.device
        dsum = (dgates * gates * (1 - gates)).contiguous()                    # through the sigmoid
        Sd = ctx.graph.dense(dt)
        pA = torch.empty((B, 2, F, Kin, G), dtype=dt, device=dev)
        pB = torch.empty((B, 2, F, Kst, F), dtype=dt, device=dev)
        pb = torch.empty((B, 2, F), dtype=dt, device=dev)
        plw = torch.empty((B, 2, F * N), dtype=dt, device=dev)
        plb = torch.empty((B, 2), dtype=dt, device=dev)
        pdh0 = torch.empty((B, 2, F, N), dtype=dt, device=dev) if ctx.needs_input_grad[1] else None
        check(lib.gcrnn_small_gates_backward(dtype_code(dt), _p(X), _p(h0), _p(wA2), _p(wB2), _p(bias2), _p(lw2), _p(Sd),
                                             _p(dsum), _p(pA), _p(pB), _p(pb), _p(plw), _p(plb), _p(pdh0), B, T, N, G, F, Kin,
                                             Kst, _stream()), 'small_gates_backward')
        return (None, pdh0.sum(dim=1) if pdh0 is not None else None, pA.sum(dim=0), pB.sum(dim=0),
                pb.sum(dim=0) if bias2 is not None else None, plw.sum(dim=0), plb.sum(dim=0) if ctx.has_lb else None, None)


def small_time_gates(X, h0, wA2, wB2, bias2, lw2, lb2, graph):
    return _SmallTimeGates.apply(X, h0, wA2, wB2, bias2, lw2, lb2, graph)


# ------------------------------------------------------------------------------------------ per-node head
def node_linear_supported(F, O, dtype, N=None, wdtype=None):
    if dtype == torch.bfloat16:            # bf16 activations (the fused cell's output), fp32 master or bf16 parameters
        return N is not None and wdtype in (torch.float32, torch.bfloat16) and \
            bool(lib.gcrnn_node_linear_bf16_supported(int(N), int(F), int(O)))
    return dtype in (torch.float32, torch.float64) and (wdtype is None or wdtype == dtype) and 0 < F <= 64 and 0 < O <= 8


class _NodeLinear(torch.autograd.Function):
    """y[r][o][n] = sum_f w[o][f] h[r][f][n] + b[o] on the user layout (the 'multipMlp' head, architectures.py:1616-1627).
    fp32 / fp64 throughout, or bf16 activations with fp32 / bf16 parameters (fp32 accumulation, bf16 output)."""

    @staticmethod
    def forward(ctx, h, w, b):
        require_device(h, w)
        hc, wc = h.contiguous(), w.contiguous()
        bc = b.contiguous() if b is not None else None
        R, F, N = hc.shape
        O = wc.shape[0]
        y = torch.empty((R, O, N), dtype=hc.dtype, device=hc.device)
        if hc.dtype == torch.bfloat16:
            assert bc is None or bc.dtype == wc.dtype
            check(lib.gcrnn_node_linear_bf16_forward(dtype_code(wc.dtype), _p(hc), _p(wc), _p(bc), _p(y), R, N, F, O, _stream()),
                  'node_linear_bf16_forward')
        else:
            check(lib.gcrnn_node_linear_forward(dtype_code(hc.dtype), _p(hc), _p(wc), _p(bc), _p(y), R, N, F, O, _stream()),
                  'node_linear_forward')
        ctx.save_for_backward(hc, wc)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w = ctx.saved_tensors
        R, F, N = h.shape
        O = w.shape[0]
        dyc = dy.contiguous()
        dh = torch.empty_like(h) if ctx.needs_input_grad[0] else None
        if h.dtype == torch.bfloat16:
            dyc = dyc.to(torch.bfloat16)
            nb = int(lib.gcrnn_node_linear_bf16_blocks(R, N))
            pw = torch.empty((nb, O, F), dtype=torch.float32, device=h.device)
            pb = torch.empty((nb, O), dtype=torch.float32, device=h.device)
            check(lib.gcrnn_node_linear_bf16_backward(dtype_code(w.dtype), _p(h), _p(w), _p(dyc), _p(dh), _p(pw), _p(pb), R, N, F, O,
                                                      _stream()), 'node_linear_bf16_backward')
            return dh, pw.sum(dim=0).to(w.dtype), (pb.sum(dim=0).to(w.dtype) if ctx.has_bias else None)
        nb = int(lib.gcrnn_node_linear_blocks(R, N))
        pw = torch.empty((nb, O, F), dtype=h.dtype, device=h.device)
        pb = torch.empty((nb, O), dtype=h.dtype, device=h.device)
        check(lib.gcrnn_node_linear_backward(dtype_code(h.dtype), _p(h), _p(w), _p(dyc), _p(dh), _p(pw), _p(pb), R, N, F, O,
                                             _stream()), 'node_linear_backward')
        return dh, pw.sum(dim=0), (pb.sum(dim=0) if ctx.has_bias else None)


def node_linear(h, weight, bias=None):
    return _NodeLinear.apply(h, weight, bias)


# ------------------------------------------------------------------------------------------ loss
class _L1Loss(torch.autograd.Function):
    """mean |x - y| with the gradient produced in the same pass (reference batchTimeL1Loss, miscTools.py:112-119)."""

    @staticmethod
    def forward(ctx, x, y):
        require_device(x, y)
        xc, yc = x.contiguous(), y.contiguous()
        if xc.data_ptr() % 16:                       # the kernel moves 16-byte vectors: re-base a slice with an odd offset
            xc = xc.clone()
        if yc.data_ptr() % 16:
            yc = yc.clone()
        n = xc.numel()
        want = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        grad = torch.empty_like(xc) if want else None
        acc_dt = torch.float64 if xc.dtype == torch.float64 else torch.float32
        partial = torch.empty((int(lib.gcrnn_l1_loss_blocks(n)),), dtype=acc_dt, device=xc.device)
        check(lib.gcrnn_l1_loss(dtype_code(xc.dtype), _p(xc), _p(yc), _p(grad), _p(partial), n, 1.0 / n, _stream()), 'l1_loss')
        ctx.grad = grad
        ctx.used = False
        ctx.operands = (xc, yc) if want else None      # (references, no copies: a second backward recomputes the gradient from them)
        return (partial.sum() / n).to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad
        if g is not None:
            if ctx.used:
                # the buffer was scaled in place and handed out by the first backward (other tensors may alias it): a second backward over a
                # retained graph (several heads, gradient-penalty loops -- torch's own L1Loss allows it, miscTools.py:112-119) gets a FRESH
                # tensor recomputed from the operands, sign(x - y) * gout / n, instead
                xc, yc = ctx.operands
                g2 = torch.sign(xc - yc) * (gout.detach().to(xc.dtype) / xc.numel())
                return (g2 if ctx.needs_input_grad[0] else None), (-g2 if ctx.needs_input_grad[1] else None)
            ctx.used = True
            # chain rule through the scalar loss WITHOUT a pass over g when the upstream gradient is 1 (loss.backward()): the kernel
            # reads the device scalar r = gout and returns at once when r == 1 (no host sync: capturable)
            r = gout.detach().to(torch.float64 if g.dtype == torch.float64 else torch.float32).reshape(1).contiguous()
            check(lib.gcrnn_scale_unless_one(dtype_code(g.dtype), _p(g), _p(r), g.numel(), _stream()), 'scale_unless_one')
        return (g if ctx.needs_input_grad[0] else None), (-g if (g is not None and ctx.needs_input_grad[1]) else None)


def l1_loss(x, y):
    assert x.shape == y.shape and x.dtype == y.dtype
    return _L1Loss.apply(x, y)


def batch_time_mse(xv, yv):
    """mean_c sqrt(sum_r (x - y)^2) / sqrt(sum_r y^2) of two [R][C] matrices (reference batchTimeMSELoss,
    miscTools.py:121-130); no autograd. Returns a 0-dim tensor (fp64 for fp64 inputs, else fp32)."""
    require_device(xv, yv)
    R, Cc = xv.shape
    acc_dt = torch.float64 if xv.dtype == torch.float64 else torch.float32
    slabs = int(lib.gcrnn_batch_time_mse_slabs(R, Cc))
    part = torch.empty((slabs, 2, Cc), dtype=acc_dt, device=xv.device)
    out = torch.empty((1,), dtype=acc_dt, device=xv.device)
    check(lib.gcrnn_batch_time_mse(dtype_code(xv.dtype), _p(xv), _p(yv), _p(part), _p(out), R, Cc, _stream()), 'batch_time_mse')
    return out[0]


# ------------------------------------------------------------------------------------------ row-linear layers
class _RowLinear(torch.autograd.Function):
    """y = x W^T (+ b) over a huge number of rows (x: R x in; R = T*N*B node-rows of the per-node head and of the
    attention projections). The weight gradient dW = dy^T x reduces over R: a single out x R x in GEMM / GEMV with
    out, in ~ 1..64 and R ~ 1e5..1e7 is a shape the BLAS back ends handle badly (measured: 4.3 ms per call in fp64
    for 1 x 40000 x 20), so the reduction is split into R/C-deep batched GEMMs followed by a sum over the C chunks."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        y = x @ w.t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        R = x.shape[0]
        dx = dy @ w if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            C = max(1, min(256, R // 512))
            Rc = (R // C) * C
            dw = torch.bmm(dy[:Rc].reshape(C, R // C, -1).transpose(1, 2), x[:Rc].reshape(C, R // C, -1)).sum(0)
            if Rc < R:
                dw = dw + dy[Rc:].t() @ x[Rc:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def row_linear(x2d, weight, bias=None):
    return _RowLinear.apply(x2d, weight, bias)


# ------------------------------------------------------------------------------------------ edge gate (attention)
class _EdgeAttention(torch.autograd.Function):
    """y[t, n] = sum_m softmax_row_m(LeakyReLU(s1[n] + s2[m])) (S+I)[m, n] Wx[t, m] on the CSR support (reference
    graphAttention, graphML.py:605-625). Wx: [T][N][B][F], s1 / s2: [T][N][B]."""

    @staticmethod
    def forward(ctx, Wx, s1, s2, graph, e, slope):
        Wx, s1, s2 = Wx.contiguous(), s1.contiguous(), s2.contiguous()
        require_device(Wx, s1, s2)
        T, N, B, F = Wx.shape
        m = graph.mask
        val = graph.mask_vals[e].to(Wx.dtype)
        trp, trow, tpos, _ = graph.mask_transposed()
        alpha = torch.empty((T, m.nnz, B), dtype=Wx.dtype, device=Wx.device)
        y = torch.empty_like(Wx)
        check(lib.gcrnn_attention_forward(dtype_code(Wx.dtype), _p(m.rowptr), _p(m.col), _p(val), _p(trp), _p(trow),
                                          _p(tpos), _p(Wx), _p(s1), _p(s2), _p(alpha), _p(y), T, N, B, F, m.nnz,
                                          float(slope), _stream()), 'attention_forward')
        ctx.save_for_backward(Wx, s1, s2, alpha, val)
        ctx.graph, ctx.slope = graph, float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        Wx, s1, s2, alpha, val = ctx.saved_tensors
        T, N, B, F = Wx.shape
        m = ctx.graph.mask
        trp, trow, tpos, erow = ctx.graph.mask_transposed()
        dy = dy.contiguous()
        dWx = torch.empty_like(Wx)
        ds1 = torch.empty_like(s1)
        ds2 = torch.empty_like(s2)
        dz = torch.empty_like(alpha)
        check(lib.gcrnn_attention_backward(dtype_code(Wx.dtype), _p(m.rowptr), _p(m.col), _p(val), _p(erow), _p(trp),
                                           _p(tpos), _p(Wx), _p(s1), _p(s2), _p(alpha), _p(dy), _p(dWx), _p(ds1), _p(ds2), _p(dz),
                                           T, N, B, F, m.nnz, ctx.slope, _stream()), 'attention_backward')
        return dWx, ds1, ds2, None, None, None


def edge_attention(Wx, s1, s2, graph, e=0, negative_slope=0.2):
    return _EdgeAttention.apply(Wx, s1, s2, graph, e, negative_slope)


# ------------------------------------------------------------------------------------------ hipGraph replay
class FusedForwardGraph(object):
    """The fused forward of one fixed problem (shapes, graph) captured as a hipGraph: pack -> gate pre-passes ->
    T step launches become one graph launch per call (re-captured when a parameter of the cell has changed since). The capture runs on the CALLER's tensors when they are given
    (zero-copy: refill X / h0 in place, or keep feeding the same resident batch, and call the runner with no arguments);
    without them the runner owns static inputs and `runner(X, h0)` copies into them first. The output tensor is reused
    between calls (clone it if it must survive the next replay).

        runner = FusedForwardGraph(cell, B, T, X=X, h0=h0);  H = runner()          # replays on X, h0 as they are now
        runner = FusedForwardGraph(cell, B, T);              H = runner(X2, h02)   # copies, then replays
    """

    def __init__(self, cell, B, T, device=None, X=None, h0=None):
        dev = device if device is not None else cell.weight_A.device
        self.cell = cell
        self.X = X if X is not None else torch.zeros((B, T, cell.G, cell.N), dtype=torch.bfloat16, device=dev)
        self.h0 = h0 if h0 is not None else torch.zeros((B, cell.F, cell.N), dtype=torch.bfloat16, device=dev)
        assert tuple(self.X.shape) == (B, T, cell.G, cell.N) and self.X.is_contiguous() and self.h0.is_contiguous()
        cell.graph.fused_plan()                                   # host-side preparation happens outside the capture
        self._dev = dev
        self._stream = torch.cuda.Stream(device=dev)
        # the fused forward of THIS cell: node- and edge-gated cells have their own (a runner that replayed the un-gated / time-gated path for them
        # would compute another cell)
        with torch.no_grad():
            if cell.spatial_gating == 'node' and cell._use_fused_node(self.X, self.h0):
                self._forward = cell._forward_fused_node
            elif cell.spatial_gating == 'edge' and cell._use_fused_edge(self.X, self.h0):
                self._forward = cell._forward_fused_edge
            elif cell.spatial_gating is None and cell._use_fused(self.X, self.h0):
                self._forward = cell._forward_fused
            elif cell._state_padded(self.X, self.h0) is not None:
                self._forward = lambda X_, h0_: cell(X_, h0_)      # (a state width between the kernels': the zero-padded shadow cell, pads captured with it)
            else:
                raise ValueError('FusedForwardGraph: this cell / problem does not run on the fused forward kernels (bf16, tanh, N <= 1024, F <= 64)')
        self._capture()

    def _capture(self):
        """Warm-up and capture on ONE side stream. The capture records the parameter packs too (ops._cached_pack never answers from its cache
        during a capture), so every replay packs from the LIVE parameters: in-place updates, optimiser steps through raw pointers and `.data`
        writes are all seen without capturing again. Only a parameter whose STORAGE was replaced (`p.data = t`, `.to(...)`) needs a new capture:
        the graph holds the old pointers."""
        cell, dev, s = self.cell, self._dev, self._stream
        with torch.no_grad():
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(2):                                # warm-up on the side stream (allocator, func attributes)
                    self._forward(self.X, self.h0)
            torch.cuda.current_stream(dev).wait_stream(s)
            self._ptrs = tuple((p.data_ptr(), p.dtype, tuple(p.shape)) for p in cell.parameters())
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=s):
                self.H = self._forward(self.X, self.h0)

    def __call__(self, X=None, h0=None):
        if X is not None and X.data_ptr() != self.X.data_ptr():
            self.X.copy_(X)
        if h0 is not None and h0.data_ptr() != self.h0.data_ptr():
            self.h0.copy_(h0)
        if self._ptrs != tuple((p.data_ptr(), p.dtype, tuple(p.shape)) for p in self.cell.parameters()):
            self._capture()                                       # a parameter's storage was replaced: the graph holds the old pointer
        self.graph.replay()
        return self.H
