#!/usr/bin/env python3
"""Generate golden vectors from the IMPORTED reference (build container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference (luanaruiz9/gated_gcrnns) is imported read-only from
/root/reference, run on CPU in float64, and its inputs / parameters (by
state_dict key) / outputs / autograd gradients are stored as small .npz files
next to this script. Only data is stored; no reference source travels.
The fixtures are what pins oracle/gcrnn_oracle.py and the HIP path
(SURVEY.md section 8c, items G1-G8). The script also asserts, at generation
time, that the numpy oracle reproduces every reference output to <= 1e-12.
"""
import os
import sys
import pickle

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('GCRNN_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

torch.set_default_dtype(torch.float64)     # the drivers' default (kStepPredGRNNs.py:44)

import Utils.graphML as gml                # noqa: E402  (reference)
import Modules.architectures as archit    # noqa: E402  (reference)
import Utils.miscTools as misc             # noqa: E402  (reference)
from oracle import gcrnn_oracle as orc     # noqa: E402

TOL = 1e-12


def directed_gso(N, density, seed):
    rng = np.random.default_rng(seed)
    M = (rng.random((N, N)) < density) * rng.uniform(0.2, 1.5, (N, N)) * rng.choice([-1.0, 1.0], (N, N), p=[0.2, 0.8])
    np.fill_diagonal(M, 0.0)
    lam = np.max(np.abs(np.linalg.eigvals(M)))
    return (M / lam).reshape(1, N, N)


def sbm_gso(N, C, p_in, p_out, seed):
    """Undirected SBM adjacency / lambda_max, redrawn until connected (recipe of graphTools.py:581-634)."""
    rng = np.random.default_rng(seed)
    labels = np.repeat(np.arange(C), N // C)
    labels = np.concatenate([labels, np.arange(N - labels.size) % C])
    while True:
        P = np.where(labels[:, None] == labels[None, :], p_in, p_out)
        U = rng.random((N, N)) < P
        W = np.triu(U, 1)
        W = (W + W.T).astype(np.float64)
        # connectivity by BFS
        seen = np.zeros(N, bool)
        seen[0] = True
        frontier = [0]
        while frontier:
            nxt = np.nonzero(W[frontier].sum(0) * (~seen))[0]
            seen[nxt] = True
            frontier = list(nxt)
        if seen.all():
            break
    lam = np.max(np.linalg.eigvalsh(W))
    return (W / lam).reshape(1, N, N), W


def sd_np(module, prefix=''):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix=''):
    out = {}
    for k, p in module.named_parameters():
        out[prefix + k] = (p.grad.detach().numpy().copy() if p.grad is not None else None)
    return out


def save(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                if vv is not None:
                    flat['%s/%s' % (k, kk)] = vv
        elif v is not None:
            flat[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **flat)
    print('wrote %-34s %7.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def check(a, b, what):
    err = np.max(np.abs(a - b)) if a.size else 0.0
    assert err <= TOL, '%s: oracle vs reference max|diff| = %g' % (what, err)
    return err


# --------------------------------------------------------------------------- G1 / G2
def g1_lsigf():
    N, B, G, F, K = 30, 3, 2, 5, 3
    S = directed_gso(N, 0.15, 1)
    rng = np.random.default_rng(11)
    h = rng.standard_normal((F, 1, K, G))
    x = rng.standard_normal((B, G, N))
    b = rng.standard_normal((F, 1))
    y_b = gml.LSIGF(torch.tensor(h), torch.tensor(S), torch.tensor(x), torch.tensor(b)).numpy()
    y_nb = gml.LSIGF(torch.tensor(h), torch.tensor(S), torch.tensor(x)).numpy()
    check(orc.lsigf(h, S, x, b), y_b, 'G1 bias')
    check(orc.lsigf(h, S, x, None), y_nb, 'G1 nobias')
    # two edge features as well (E=2)
    S2 = np.concatenate([S, directed_gso(N, 0.1, 2)], axis=0)
    h2 = rng.standard_normal((F, 2, K, G))
    y_e2 = gml.LSIGF(torch.tensor(h2), torch.tensor(S2), torch.tensor(x), torch.tensor(b)).numpy()
    check(orc.lsigf(h2, S2, x, b), y_e2, 'G1 E=2')
    # gradients (autograd) wrt h, x, b for loss = sum(y * r)
    r = rng.standard_normal(y_b.shape)
    ht, xt, bt = (torch.tensor(v, requires_grad=True) for v in (h, x, b))
    (gml.LSIGF(ht, torch.tensor(S), xt, bt) * torch.tensor(r)).sum().backward()
    save('g1_lsigf', S=S, h=h, x=x, b=b, y_bias=y_b, y_nobias=y_nb, S2=S2, h2=h2, y_e2=y_e2,
         r=r, grad_h=ht.grad.numpy(), grad_x=xt.grad.numpy(), grad_b=bt.grad.numpy())


def g2_graphfilter():
    N, B, G, F, K = 30, 3, 2, 5, 3
    S = directed_gso(N, 0.15, 1)
    torch.manual_seed(2)
    gf = gml.GraphFilter(G, F, K)
    gf.addGSO(torch.tensor(S))
    rng = np.random.default_rng(12)
    x = rng.standard_normal((B, G, N))
    xs = rng.standard_normal((B, G, N - 7))            # Nin < N zero-pad path (graphML.py:1181-1193)
    y = gf(torch.tensor(x)).detach().numpy()
    ys = gf(torch.tensor(xs)).detach().numpy()
    p = sd_np(gf)
    check(orc.graph_filter(p['weight'], p['bias'], S, x), y, 'G2')
    check(orc.graph_filter(p['weight'], p['bias'], S, xs), ys, 'G2 short')
    save('g2_graphfilter', S=S, x=x, x_short=xs, y=y, y_short=ys, params=p)


# --------------------------------------------------------------------------- G3 / G4
VARIANTS = [('none', False, None), ('time', True, None), ('node', False, 'node'),
            ('edge', False, 'edge'), ('time_node', True, 'node'), ('time_edge', True, 'edge')]


def g3_g4_cells():
    N, T, G, F, K, B = 30, 4, 2, 5, 3, 3
    S = directed_gso(N, 0.15, 1)
    rng = np.random.default_rng(13)
    X = rng.standard_normal((B, T, G, N))
    h0 = 0.5 * rng.standard_normal((B, F, N))          # non-zero h0 exposes the h0-not-h_{t-1} gate quirk
    target = rng.standard_normal((B, T, F, N))
    for name, tg, sg in VARIANTS:
        torch.manual_seed(30)
        cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, sg, 1, True)
        cell.addGSO(torch.tensor(S))
        p = sd_np(cell)
        Xt = torch.tensor(X, requires_grad=True)
        h0t = torch.tensor(h0, requires_grad=True)
        H = cell(Xt, h0t)
        check(orc.ggcrnn_cell(p, S, X, h0, tg, sg), H.detach().numpy(), 'G3 ' + name)
        cell.zero_grad()
        H.sum().backward(retain_graph=True)
        g_sum = grads_np(cell)
        gx_sum, gh0_sum = Xt.grad.numpy().copy(), h0t.grad.numpy().copy()
        cell.zero_grad(); Xt.grad = None; h0t.grad = None
        torch.nn.L1Loss()(H, torch.tensor(target)).backward()
        g_l1 = grads_np(cell)
        save('g3_cell_' + name, S=S, X=X, h0=h0, target=target, H=H.detach().numpy(), params=p,
             grad_sum=g_sum, grad_sum_X=gx_sum, grad_sum_h0=gh0_sum,
             grad_l1=g_l1, grad_l1_X=Xt.grad.numpy(), grad_l1_h0=h0t.grad.numpy())
    # no-bias variant of the plain and time-gated cell
    for name, tg, sg in VARIANTS[:2]:
        torch.manual_seed(31)
        cell = gml.GGCRNNCell(G, F, K, K - 1, torch.tanh, tg, sg, 1, False)   # Kin != Kst too
        cell.addGSO(torch.tensor(S))
        p = sd_np(cell)
        H = cell(torch.tensor(X), torch.tensor(h0)).detach().numpy()
        check(orc.ggcrnn_cell(p, S, X, h0, tg, sg), H, 'G3 nobias ' + name)
        save('g3_cell_%s_nobias' % name, S=S, X=X, h0=h0, H=H, params=p)


# --------------------------------------------------------------------------- G5
def g5_models():
    rng = np.random.default_rng(15)
    # config 1: SBM N=50, T=8, G=1, F_h=20, taps 2   (BASELINE.json configs[0])
    S50, _ = sbm_gso(50, 5, 0.8, 0.2, 5)
    B, T = 6, 8
    x = rng.standard_normal((B, T, 1, 50))
    h0 = np.zeros((B, 20, 50))
    for mlp in ('multipMlp', 'oneMlp'):
        for name, tg, sg in (('none', False, None), ('time', True, None)):
            torch.manual_seed(50)
            dims = [1] if mlp == 'multipMlp' else [50]
            m = archit.GatedGCRNNforRegression(1, 20, 2, 2, torch.tanh, torch.nn.ReLU, dims, S50[0], True,
                                               time_gating=tg, spatial_gating=sg, mlpType=mlp)
            y = m(torch.tensor(x), torch.tensor(h0)).detach().numpy()
            p = sd_np(m)
            check(orc.gated_gcrnn_regression(p, S50, x, h0, tg, sg, mlp), y, 'G5 reg %s %s' % (mlp, name))
            save('g5_reg_%s_%s' % (mlp, name), S=S50, x=x, h0=h0, y=y, params=p)
    # seismic graph Adj.p: directed 59 nodes (driver normalisation epicenterEstimation.py:619)
    with open(os.path.join(REF, 'Adj.p'), 'rb') as f:
        A = np.asarray(pickle.load(f), dtype=np.float64)
    lam = np.max(np.abs(np.linalg.eigvals(A)))
    S59 = (A / lam).reshape(1, 59, 59)
    np.save(os.path.join(HERE, 'adj59.npy'), A)
    for (T, K, tag) in ((20, 4, 'T20K4'), (200, 3, 'T200K3')):
        Bq = 4
        xq = rng.standard_normal((Bq, T, 1, 59))
        h0q = np.zeros((Bq, 20, 59))
        for name, tg, sg in (('none', False, None), ('time', True, None)):
            torch.manual_seed(59)
            m = archit.GatedGCRNNforClassification(1, 20, K, K, torch.tanh, torch.nn.ReLU, [11], S59[0], True,
                                                   time_gating=tg, spatial_gating=sg)
            if T >= 100:
                # With G=1 the reference init draws weight_B ~ U(+-1/sqrt(K)) on a 20x20 state map
                # (graphML.py:2231-2233): a chaotic recurrence in which 200 steps amplify any
                # rounding-order difference to O(0.1). Shrink it so the long-sequence golden pins
                # arithmetic, not chaos.
                with torch.no_grad():
                    m.stateGCRNN.weight_B.mul_(0.25)
            y = m(torch.tensor(xq), torch.tensor(h0q)).detach().numpy()
            Hlast = m.stateGCRNN(torch.tensor(xq), torch.tensor(h0q)).detach().numpy()[:, -1]
            p = sd_np(m)
            check(orc.gated_gcrnn_classification(p, S59, xq, h0q, tg, sg), y, 'G5 cls %s %s' % (tag, name))
            save('g5_cls_%s_%s' % (tag, name), S=S59, x=xq, h0=h0q, y=y, h_last=Hlast, params=p)


# --------------------------------------------------------------------------- G6
def g6_training_trace():
    """20 Adam steps, L1 loss, RMSE-like metric -- the inner loop of train_rnn.py:247-288 on fixed tensors."""
    S, W = sbm_gso(50, 5, 0.8, 0.2, 6)
    rng = np.random.default_rng(16)
    B, T, N = 20, 5, 50
    A = S[0]
    # KStepPrediction recipe (dataTools.py:1275-1302): x_{t+1} = x_t A + w_t
    xs = np.zeros((B, T + 1, N))
    xs[:, 0] = rng.random((B, N))
    for t in range(T):
        xs[:, t + 1] = xs[:, t] @ A + 0.1 * rng.standard_normal((B, N))
    x = xs[:, :T].reshape(B, T, 1, N)
    y = xs[:, 1:].reshape(B, T, 1, N)
    h0 = np.zeros((B, 20, N))
    for name, tg in (('GCRNNMLP', False), ('TimeGCRNNMLP', True)):
        torch.manual_seed(60)
        m = archit.GatedGCRNNforRegression(1, 20, 3, 3, torch.tanh, torch.nn.ReLU, [1], A, True,
                                           time_gating=tg, spatial_gating=None, mlpType='multipMlp')
        p0 = sd_np(m)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
        losses, metrics = [], []
        for it in range(20):
            m.zero_grad()
            yhat = m(torch.tensor(x), torch.tensor(h0))
            loss = misc.batchTimeL1Loss(yhat, torch.tensor(y))
            loss.backward()
            opt.step()
            losses.append(loss.item())
            metrics.append(misc.batchTimeMSELoss(yhat.detach(), torch.tensor(y)).item())
        # oracle check of the loss/metric on the initial parameters
        y0 = orc.gated_gcrnn_regression(p0, S, x, h0, tg, None, 'multipMlp')
        assert abs(orc.batch_time_l1_loss(y0, y) - losses[0]) < TOL
        assert abs(orc.batch_time_mse_loss(y0, y) - metrics[0]) < 1e-10
        save('g6_trace_' + name, S=S, x=x, y=y, h0=h0, params0=p0, params20=sd_np(m),
             loss=np.array(losses), metric=np.array(metrics))


# --------------------------------------------------------------------------- G7
def g7_csr():
    """Index fixtures: CSR(S^T), CSR(S) from S != 0, and the attention mask |S+I| > 1e-9."""
    with open(os.path.join(REF, 'Adj.p'), 'rb') as f:
        A = np.asarray(pickle.load(f), dtype=np.float64)
    out = {}
    for tag, M in (('dir30', directed_gso(30, 0.15, 1)[0]), ('adj59', A), ('sbm50', sbm_gso(50, 5, 0.8, 0.2, 5)[0][0])):
        for nm, mat in (('S', M), ('ST', M.T.copy())):
            rp, col, val = orc.csr_from_dense(mat)
            out['%s/%s_rowptr' % (tag, nm)] = rp
            out['%s/%s_col' % (tag, nm)] = col
            out['%s/%s_val' % (tag, nm)] = val
        mask = np.abs(M + np.eye(M.shape[0])) > 1e-9            # graphML.py:577, 611-613
        rp, col, _ = orc.csr_from_dense(mask.astype(np.float64))
        out['%s/mask_rowptr' % tag] = rp
        out['%s/mask_col' % tag] = col
        out['%s/dense' % tag] = M
    path = os.path.join(HERE, 'g7_csr.npz')
    np.savez_compressed(path, **out)
    print('wrote g7_csr.npz')


# --------------------------------------------------------------------------- G8
def g8_midsize():
    """N=1000 sparse SBM, B=2,T=4,G=F=64,K=5: store COO + seeds + sampled outputs (small fixture)."""
    S, W = sbm_gso(1000, 5, 0.04, 0.0025, 0)
    N, B, T, G, F, K = 1000, 2, 4, 64, 64, 5
    rng = np.random.default_rng(18)
    X = rng.standard_normal((B, T, G, N))
    h0 = np.zeros((B, F, N))
    torch.manual_seed(80)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    H = cell(torch.tensor(X), torch.tensor(h0)).detach().numpy()
    p = sd_np(cell)
    idx = rng.choice(H.size, size=H.size // 100, replace=False)
    rows, cols = np.nonzero(S[0])
    save('g8_mid', coo_row=rows.astype(np.int32), coo_col=cols.astype(np.int32), coo_val=S[0][rows, cols],
         x_seed=np.array([18]), shape=np.array([N, B, T, G, F, K]), params=p,
         sample_idx=idx.astype(np.int64), sample_val=H.reshape(-1)[idx],
         checksum=np.array([H.sum(), np.abs(H).sum(), (H ** 2).sum()]),
         H_b0_t3_f0=H[0, 3, 0], H_b1_t0=H[1, 0, :, :8])


def bf16_round(a):
    """Nearest bf16-representable value (round to nearest even on the fp32 bit pattern), returned as float64."""
    u = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def g9_fused_bptt(variants=(('none', False, None), ('time', True, None), ('node', False, 'node'))):
    """Reference autograd gradients at a shape the fused bf16 kernels support (N=200, F=G=32, K=3, T=4, B=3): every
    operand is bf16-representable (S fp32-representable) so that the bf16 kernels and the fp64 reference see the SAME
    numbers; non-zero h0, directed weighted S; losses H.sum() and L1 against a random target (the drivers' loss,
    miscTools.py:112-119). Un-gated, time-gated and node-gated cells (edge-gated: g9_fused_bptt_edge); gradients of every parameter incl. the gate
    sub-networks."""
    N, T, G, F, K, B = 200, 4, 32, 32, 3, 3
    rng = np.random.default_rng(19)
    M = (rng.random((N, N)) < 0.05) * rng.uniform(0.2, 1.5, (N, N)) * rng.choice([-1.0, 1.0], (N, N), p=[0.2, 0.8])
    np.fill_diagonal(M, 0.0)
    S = (M / np.max(np.abs(np.linalg.eigvals(M)))).astype(np.float32).astype(np.float64).reshape(1, N, N)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.5 * rng.standard_normal((B, F, N)))
    target = bf16_round(rng.standard_normal((B, T, F, N)))
    rows, cols = np.nonzero(S[0])
    for name, tg, sg in variants:
        torch.manual_seed(90)
        cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, sg, 1, True)
        cell.addGSO(torch.tensor(S))
        with torch.no_grad():
            for q in cell.parameters():
                q.copy_(torch.tensor(bf16_round(q.detach().numpy())))
        p = sd_np(cell)
        h0t = torch.tensor(h0, requires_grad=True)
        Xt = torch.tensor(X, requires_grad=True)            # (round 3: the reference's gradient w.r.t. the input sequence too)
        H = cell(Xt, h0t)
        check(orc.ggcrnn_cell(p, S, X, h0, tg, sg), H.detach().numpy(), 'G9 ' + name)
        cell.zero_grad()
        H.sum().backward(retain_graph=True)
        g_sum, gh0_sum, gX_sum = grads_np(cell), h0t.grad.numpy().copy(), Xt.grad.numpy().copy()
        cell.zero_grad(); h0t.grad = None; Xt.grad = None
        torch.nn.L1Loss()(H, torch.tensor(target)).backward()
        g_l1 = grads_np(cell)
        f32 = lambda d: {k: (v.astype(np.float32) if v is not None else None) for k, v in d.items()}
        save('g9_fused_' + name, coo_row=rows.astype(np.int16), coo_col=cols.astype(np.int16), coo_val=S[0][rows, cols].astype(np.float32),
             shape=np.array([N, T, G, F, K, B]), X=X.astype(np.float32), h0=h0.astype(np.float32), target=target.astype(np.float32),
             H=H.detach().numpy().astype(np.float32), params=f32(p), grad_sum=f32(g_sum), grad_sum_h0=gh0_sum.astype(np.float32),
             grad_l1=f32(g_l1), grad_l1_h0=h0t.grad.numpy().astype(np.float32),
             grad_sum_X=gX_sum.astype(np.float32), grad_l1_X=Xt.grad.numpy().astype(np.float32))


def g9_fused_bptt_edge():
    """G9 for the edge-gated cells (round 2: fused attention kernels): same graph, operands and losses, the reference's
    GraphAttentional gates (graphML.py:2409-2416) under autograd."""
    g9_fused_bptt((('edge', False, 'edge'), ('time_edge', True, 'edge')))


def g11_fused_f32():
    """G11 (round 3): reference autograd at a shape the fp32-accurate fused kernels support -- N = 200, F = G = 32, K = 3, T = 4,
    B = 3 on a UNIFORM-weight undirected graph (the drivers' S = W / lambda_max, kStepPredGRNNs.py:768), every operand
    fp32-representable (NOT bf16-rounded): un-gated cell, fp64 reference, gradients of every parameter and of h0 for the losses
    H.sum() and L1 (miscTools.py:112-119). The fp32 kernels are compared at <= 1e-5 (H) / <= 2e-5 of each gradient's max."""
    N, T, G, F, K, B = 200, 4, 32, 32, 3, 3
    rng = np.random.default_rng(23)
    U = np.triu(rng.random((N, N)) < 0.05, 1)
    W = (U + U.T).astype(np.float64)
    lam = np.max(np.linalg.eigvalsh(W))
    w32 = np.float32(1.0 / lam)
    S = (W * np.float64(w32)).reshape(1, N, N)                 # ONE fp32-representable weight on every edge
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = f32(rng.standard_normal((B, T, G, N)))
    h0 = f32(0.5 * rng.standard_normal((B, F, N)))
    target = f32(rng.standard_normal((B, T, F, N)))
    rows, cols = np.nonzero(S[0])
    torch.manual_seed(91)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    with torch.no_grad():
        for q in cell.parameters():
            q.copy_(torch.tensor(f32(q.detach().numpy())))
    p = sd_np(cell)
    h0t = torch.tensor(h0, requires_grad=True)
    H = cell(torch.tensor(X), h0t)
    check(orc.ggcrnn_cell(p, S, X, h0, False, None), H.detach().numpy(), 'G11')
    cell.zero_grad()
    H.sum().backward(retain_graph=True)
    g_sum, gh0_sum = grads_np(cell), h0t.grad.numpy().copy()
    cell.zero_grad(); h0t.grad = None
    torch.nn.L1Loss()(H, torch.tensor(target)).backward()
    g_l1 = grads_np(cell)
    save('g11_fused_f32', coo_row=rows.astype(np.int16), coo_col=cols.astype(np.int16), coo_val=S[0][rows, cols].astype(np.float32),
         shape=np.array([N, T, G, F, K, B]), X=X.astype(np.float32), h0=h0.astype(np.float32), target=target.astype(np.float32),
         H=H.detach().numpy(), params={k: v.astype(np.float32) for k, v in p.items()}, grad_sum=g_sum, grad_sum_h0=gh0_sum,
         grad_l1=g_l1, grad_l1_h0=h0t.grad.numpy())


def g12_fused_f32_time():
    """G12 (round 4): G11's graph, shapes and operand recipe for the TIME-GATED cell (the reference's default, graphML.py:2196): fp64
    reference states and autograd gradients of EVERY parameter -- the cell's taps and bias, both gate sub-cells (GFL_in / GFL_forget) and
    both read-outs (MLP_in / MLP_forget) -- for the losses H.sum() and L1 (miscTools.py:112-119); non-zero h0 so that the gate cells' state
    taps get a gradient. The fp32-accurate fused training (ops.fused_cell_train_x3_gated) is compared at <= 1e-5 (H) / <= 2e-5 of each
    gradient's max."""
    N, T, G, F, K, B = 200, 4, 32, 32, 3, 3
    rng = np.random.default_rng(29)
    U = np.triu(rng.random((N, N)) < 0.05, 1)
    W = (U + U.T).astype(np.float64)
    lam = np.max(np.linalg.eigvalsh(W))
    w32 = np.float32(1.0 / lam)
    S = (W * np.float64(w32)).reshape(1, N, N)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = f32(rng.standard_normal((B, T, G, N)))
    h0 = f32(0.5 * rng.standard_normal((B, F, N)))
    target = f32(rng.standard_normal((B, T, F, N)))
    rows, cols = np.nonzero(S[0])
    torch.manual_seed(92)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, True, None, 1, True)
    cell.addGSO(torch.tensor(S))
    with torch.no_grad():
        for name, q in cell.named_parameters():
            if name.startswith('MLP_'):
                q.mul_(6.0)                              # read-outs of F N = 6400 inputs: default init leaves the gates within 1e-2 of 0.5
        for q in cell.parameters():
            q.copy_(torch.tensor(f32(q.detach().numpy())))
    p = sd_np(cell)
    H = cell(torch.tensor(X), torch.tensor(h0))
    check(orc.ggcrnn_cell(p, S, X, h0, True, None), H.detach().numpy(), 'G12')
    cell.zero_grad()
    H.sum().backward(retain_graph=True)
    g_sum = grads_np(cell)
    cell.zero_grad()
    torch.nn.L1Loss()(H, torch.tensor(target)).backward()
    g_l1 = grads_np(cell)
    used = {k: v for k, v in g_sum.items() if not k.startswith(('GFL_out', 'MLP_out'))}      # (created but never used by the reference's forward)
    assert len(used) == 13 and all(v is not None and np.abs(v).max() > 0 for v in used.values()), list(used)
    save('g12_fused_f32_time', coo_row=rows.astype(np.int16), coo_col=cols.astype(np.int16), coo_val=S[0][rows, cols].astype(np.float32),
         shape=np.array([N, T, G, F, K, B]), X=X.astype(np.float32), h0=h0.astype(np.float32), target=target.astype(np.float32),
         H=H.detach().numpy(), params={k: v.astype(np.float32) for k, v in p.items()}, grad_sum=g_sum, grad_l1=g_l1)


def g13_fused_f32_node():
    """G13 (round 5): G11's graph, shapes and operand recipe for the NODE-gated cell (Utils/graphML.py:2379-2407) and the time + node gated one:
    fp64 reference states (and autograd gradients of every used parameter for H.sum(): kept for the training path) with fp32-representable
    operands and non-zero h0. The node-gated forward on the fp32-accurate fused kernels (ops.fused_node_cell_forward_x3) is compared at <= 1e-5."""
    N, T, G, F, K, B = 200, 4, 32, 32, 3, 3
    rng = np.random.default_rng(31)
    U = np.triu(rng.random((N, N)) < 0.05, 1)
    W = (U + U.T).astype(np.float64)
    lam = np.max(np.linalg.eigvalsh(W))
    w32 = np.float32(1.0 / lam)
    S = (W * np.float64(w32)).reshape(1, N, N)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = f32(rng.standard_normal((B, T, G, N)))
    h0 = f32(0.5 * rng.standard_normal((B, F, N)))
    rows, cols = np.nonzero(S[0])
    for name, tg in (('g13_fused_f32_node', False), ('g13_fused_f32_time_node', True)):
        torch.manual_seed(93)
        cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'node', 1, True)
        cell.addGSO(torch.tensor(S))
        with torch.no_grad():
            for pname, q in cell.named_parameters():
                if pname.startswith('MLP_'):
                    q.mul_(6.0)
                if pname.startswith('GFL_node_'):
                    q.mul_(3.0)                          # (the F -> 1 gate filters: default init leaves the node gates within a few 1e-2 of 0.5)
            for q in cell.parameters():
                q.copy_(torch.tensor(f32(q.detach().numpy())))
        p = sd_np(cell)
        H = cell(torch.tensor(X), torch.tensor(h0))
        check(orc.ggcrnn_cell(p, S, X, h0, tg, 'node'), H.detach().numpy(), name)
        cell.zero_grad()
        H.sum().backward()
        g_sum = {k: v for k, v in grads_np(cell).items() if v is not None}
        save(name, coo_row=rows.astype(np.int16), coo_col=cols.astype(np.int16), coo_val=S[0][rows, cols].astype(np.float32),
             shape=np.array([N, T, G, F, K, B]), X=X.astype(np.float32), h0=h0.astype(np.float32),
             H=H.detach().numpy(), params={k: v.astype(np.float32) for k, v in p.items()}, grad_sum=g_sum)


def g14_fused_f32_edge():
    """G14 (round 5): G13's graph, shapes and operand recipe for the EDGE-gated cell (Utils/graphML.py:2409-2416; graphAttention :521-627) and the
    time + edge gated one: fp64 reference states with fp32-representable operands and non-zero h0. The edge-gated forward with both filters on
    the fp32-accurate fused kernels (ops.fused_edge_cell_forward_x3) is compared at <= 1e-5."""
    N, T, G, F, K, B = 200, 4, 32, 32, 3, 3
    rng = np.random.default_rng(37)
    U = np.triu(rng.random((N, N)) < 0.05, 1)
    W = (U + U.T).astype(np.float64)
    lam = np.max(np.linalg.eigvalsh(W))
    w32 = np.float32(1.0 / lam)
    S = (W * np.float64(w32)).reshape(1, N, N)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = f32(rng.standard_normal((B, T, G, N)))
    h0 = f32(0.5 * rng.standard_normal((B, F, N)))
    rows, cols = np.nonzero(S[0])
    for name, tg in (('g14_fused_f32_edge', False), ('g14_fused_f32_time_edge', True)):
        torch.manual_seed(94)
        cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'edge', 1, True)
        cell.addGSO(torch.tensor(S))
        with torch.no_grad():
            for pname, q in cell.named_parameters():
                if pname.startswith('MLP_'):
                    q.mul_(6.0)
                if pname.endswith('attention.mixer'):
                    q.mul_(4.0)                          # (default init leaves the attention coefficients near uniform over a neighbourhood)
            for q in cell.parameters():
                q.copy_(torch.tensor(f32(q.detach().numpy())))
        p = sd_np(cell)
        with torch.no_grad():
            H = cell(torch.tensor(X), torch.tensor(h0))
        check(orc.ggcrnn_cell(p, S, X, h0, tg, 'edge'), H.numpy(), name)
        save(name, coo_row=rows.astype(np.int16), coo_col=cols.astype(np.int16), coo_val=S[0][rows, cols].astype(np.float32),
             shape=np.array([N, T, G, F, K, B]), X=X.astype(np.float32), h0=h0.astype(np.float32),
             H=H.numpy(), params={k: v.astype(np.float32) for k, v in p.items()})


def g10_kstep_data():
    """The reference's KStepPrediction dataset (Utils/dataTools.py:1259-1317) on a reference SBM graph
    (Utils/graphTools.py createGraph 'SBM'), with the numpy global generator seeded: stores the graph, the noise arrays the
    reference drew (replayed with the same calls in the same order after re-seeding) and the signals / labels of every split.
    `gensim` (dataTools.py:1001, used by an unrelated text dataset) is absent here: an empty stand-in module lets the import
    through -- a generator-side shim, nothing of it is stored."""
    import types
    sys.modules.setdefault('gensim', types.ModuleType('gensim'))
    import Utils.dataTools as rdata
    import Utils.graphTools as rgraph
    N, K, horizon, nTrain, nValid, nTest = 20, 2, 7, 6, 2, 2
    sigS, sigT = 0.1, 0.1
    np.random.seed(1234)
    G = rgraph.Graph('SBM', N, {'nCommunities': 2, 'probIntra': 0.8, 'probInter': 0.2})
    np.random.seed(4321)
    data = rdata.KStepPrediction(K, G, nTrain, nValid, nTest, horizon, sigmaSpatial=sigS, sigmaTemporal=sigT)
    # replay of the reference's draws (dataTools.py:1284-1297), same calls, same order
    np.random.seed(4321)
    nTotal = nTrain + nValid + nTest
    x0 = np.random.rand(nTotal, N)
    temp = np.random.multivariate_normal(np.zeros(horizon), sigT ** 2 * np.eye(horizon) + 0.0 * np.ones((horizon, horizon)), (nTotal, N))
    temp = np.transpose(temp, (2, 0, 1))                                  # horizon x nTotal x N
    spat = np.stack([np.random.multivariate_normal(np.zeros(N), sigS ** 2 * np.eye(N) + 0.0 * np.ones((N, N)), nTotal)
                     for _ in range(horizon)])                              # horizon x nTotal x N
    out = {}
    for split in ('train', 'valid', 'test'):
        xs, ys = data.getSamples(split)
        out[split + '_signals'] = np.asarray(xs)
        out[split + '_labels'] = np.asarray(ys)
    # self-check of the replay: x_1 = x_0 A + noise must be what the reference stored as the second step of the signals
    EW = np.linalg.eigvalsh(G.W)
    A = G.W / np.max(EW)
    allsig = np.concatenate([out['train_signals'], out['valid_signals'], out['test_signals']], axis=0)
    assert np.max(np.abs(allsig[:, :N] - x0)) <= 1e-14
    assert np.max(np.abs(allsig[:, N:2 * N] - (x0 @ A + spat[0] + temp[0]))) <= 1e-12
    save('g10_kstep_data', W=np.asarray(G.W, dtype=np.float64), x0=x0, spatial=spat, temporal=temp,
         shape=np.array([N, K, horizon, nTrain, nValid, nTest]), sigma=np.array([sigS, sigT]), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1:                       # regenerate selected fixtures only: make_golden.py g9_fused_bptt ...
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    g1_lsigf()
    g2_graphfilter()
    g3_g4_cells()
    g5_models()
    g6_training_trace()
    g7_csr()
    g8_midsize()
    g9_fused_bptt()
    g9_fused_bptt_edge()
    g10_kstep_data()
    g11_fused_f32()
    g12_fused_f32_time()
    g13_fused_f32_node()
    g14_fused_f32_edge()
    print('all oracle checks passed at tol', TOL)
