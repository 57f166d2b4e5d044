"""BASELINE configs[4]: N = 1e5 nodes, edge density 1e-3 (nnz ~ 1e7), K = 3 taps, T = 16, G = F = 32, B = 8 -- the streaming
CSR SpMM path (GGCRNNCell._forward_horner -> gcrnn_taps_* + gcrnn_spmm_ex). The reference cannot run this size at all
(a dense 1e5 x 1e5 GSO, graphML.py:117,123), so the checker is the oracle's CSR restatement of the same arithmetic
(oracle.cell_step_rows_csr), itself pinned to the dense oracle on small graphs (first test, CPU)."""
import numpy as np
import pytest
import torch

from oracle import gcrnn_oracle as orc

sp = pytest.importorskip('scipy.sparse')


def _csr_T(rowptr, col, val, N):
    """scipy CSR of S^T from the CSR of S."""
    return sp.csr_matrix((val, col, rowptr), shape=(N, N)).T.tocsr()


def test_csr_oracle_matches_dense_oracle():
    rng = np.random.default_rng(3)
    N, B, G, F, K = 40, 3, 2, 5, 3
    S = (rng.random((N, N)) < 0.15) * rng.uniform(0.1, 1.0, (N, N))          # directed, weighted
    S = S / np.max(np.abs(np.linalg.eigvals(S)))
    P = sp.csr_matrix(S).T.tocsr()
    params = {'weight_A': rng.uniform(-.4, .4, (F, 1, K, G)), 'weight_B': rng.uniform(-.4, .4, (F, 1, K, F)),
              'bias': rng.uniform(-.4, .4, (F, 1))}
    x = rng.standard_normal((B, 1, G, N))
    h0 = 0.5 * rng.standard_normal((B, F, N))
    ref = orc.ggcrnn_cell(params, S.reshape(1, N, N), x, h0)[:, 0]                     # B x F x N
    rows = np.array([0, 7, 7, 13, 39])
    got = orc.cell_step_rows_csr(params, P, x[:, 0], h0, rows)
    assert np.max(np.abs(got - ref[:, :, rows])) <= 1e-12
    y = orc.lsigf_rows_csr(params['weight_A'], P, x[:, 0], None, np.arange(N))
    assert np.max(np.abs(y - orc.lsigf(params['weight_A'], S.reshape(1, N, N), x[:, 0]))) <= 1e-12


@pytest.fixture(scope='module')
def cfg5():
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd.graph import erdos_renyi_csr, operator_from_csr
    N, K, T, G, F, B = 100000, 3, 16, 32, 32, 8
    dev = torch.device('cuda:0')
    rowptr, col, val = erdos_renyi_csr(N, 1e-3, seed=0)
    graph = operator_from_csr(rowptr, col, val, N, device=dev)
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(graph)
    params = {k: v.detach().double().numpy().copy() for k, v in cell.state_dict().items()}
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(B, T, G, N, generator=gen)
    return dict(N=N, K=K, T=T, G=G, F=F, B=B, dev=dev, P=_csr_T(rowptr, col, val, N), cell=cell, params=params, X=X,
                nnz=int(col.size))


def _teacher_forced_errors(c, H, X, params, steps, rows):
    """max |H_t[rows] - oracle step from (x_t, the path's own h_{t-1})| for every t in steps (fp64 checker). Sequences are
    independent: the first and the last step are checked for every sequence of the batch, the steps in between for two of
    them (the single-threaded sparse products of the checker dominate the test's run time)."""
    errs = []
    for t in steps:
        seqs = np.arange(c['B']) if t in (0, c['T'] - 1) else np.array([0, 5])
        hprev = H[seqs, t - 1].double().numpy() if t > 0 else np.zeros((seqs.size, c['F'], c['N']))
        ref = orc.cell_step_rows_csr(params, c['P'], X[seqs, t].double().numpy(), hprev, rows)
        errs.append(float(np.max(np.abs(H[seqs, t][:, :, rows].double().numpy() - ref))))
    return errs


@pytest.mark.gpu
def test_cfg5_fp32_matches_csr_oracle_on_sampled_rows_of_every_step(cfg5):
    c = cfg5
    cell = c['cell'].to(c['dev'])
    X = c['X'].to(c['dev'])
    h0 = torch.zeros(c['B'], c['F'], c['N'], device=c['dev'])
    with torch.no_grad():
        assert cell._use_horner(X, h0)
        H = cell(X, h0)
        H2 = cell(X, h0)
    assert torch.equal(H, H2), 'the streaming path is not bit-deterministic'
    H = H.cpu()
    rows = np.sort(np.random.default_rng(11).choice(c['N'], c['N'] // 100, replace=False))      # 1 % of the nodes
    errs = _teacher_forced_errors(c, H, c['X'], c['params'], range(c['T']), rows)
    assert max(errs) <= 1e-5, errs                                # north_star: 1e-5 fp32 against the fp64 checker
    assert float(H.abs().max()) <= 1.0 and float(H[:, -1].abs().mean()) > 1e-3


@pytest.mark.gpu
def test_cfg5_bf16_matches_csr_oracle_on_sampled_rows(cfg5):
    c = cfg5
    cell = c['cell'].to(c['dev']).to(torch.bfloat16)
    try:
        Xb = c['X'].to(torch.bfloat16)
        h0 = torch.zeros(c['B'], c['F'], c['N'], device=c['dev'], dtype=torch.bfloat16)
        with torch.no_grad():
            assert cell._use_horner(Xb.to(c['dev']), h0)
            H = cell(Xb.to(c['dev']), h0)
            H2 = cell(Xb.to(c['dev']), h0)
        assert torch.equal(H, H2)
        H = H.cpu()
        params = {k: v.detach().double().cpu().numpy() for k, v in cell.state_dict().items()}      # the bf16-rounded parameters
        rows = np.sort(np.random.default_rng(12).choice(c['N'], c['N'] // 100, replace=False))
        errs = _teacher_forced_errors(c, H, Xb, params, (0, 1, 7, c['T'] - 1), rows)
        # bf16 rows: every tap, every hop result and the state are rounded to 8 significant bits (fp32 sums in between):
        # builder-chosen tolerance as for the fused bf16 kernel, not inherited from the reference
        assert max(errs) <= 3e-2, errs
    finally:
        c['cell'].float()


@pytest.mark.gpu
def test_spmm_stream_variants_agree_and_epilogue():
    """Every (piece width, unroll) variant of the streaming SpMM computes the same sums (fp64: to rounding; the order of the
    partial sums differs between variants) and the tanh/bias epilogue equals the composed expression; ragged rows (empty,
    1 entry, > 128 entries = several staging trips) included."""
    from gated_gcrnns_amd import ops
    from gated_gcrnns_amd.graph import operator_from_csr
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(7)
    N = 700
    deg = rng.integers(0, 40, size=N)
    deg[3] = 0; deg[4] = 1; deg[5] = 300; deg[N - 1] = 0
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    col = np.concatenate([np.sort(rng.choice(N, size=d, replace=False)) for d in deg]).astype(np.int32)
    val = rng.uniform(-1, 1, col.size)
    g = operator_from_csr(rowptr, col, val, N, device=dev)
    dense = np.zeros((N, N)); dense[np.repeat(np.arange(N), deg), col] = val                   # S; forward shift = S^T rows
    for dt, L, tol in ((torch.float64, 24, 1e-12), (torch.float32, 72, 1e-5), (torch.bfloat16, 256, 2e-2), (torch.float32, 1028, 1e-5)):
        X = torch.randn(2, N, L, dtype=torch.float64)
        Y0 = torch.randn(2, N, L, dtype=torch.float64)
        Xd, Y0d = X.to(dev, dt), Y0.to(dev, dt)
        ref = torch.einsum('mn,imc->inc', torch.tensor(dense), Xd.double().cpu()) + Y0d.double().cpu()
        F = 8
        bias = torch.randn(F, dtype=torch.float64)
        bd = bias.to(dev, torch.float32 if dt == torch.bfloat16 else dt)
        ref_t = torch.tanh(ref + 2.0 * bias.repeat(L // F if L % F == 0 else 1)[:L]) if L % F == 0 else None
        outs = []
        for pl in (4, 8, 16, 32, 64):
            for u in (2, 4, 8):
                if (u == 2 and pl > 8):
                    continue
                Y = Y0d.clone()
                ops.spmm_raw(g.fwd[0], Xd, out=Y, accumulate=True, tune=dict(piece_lanes=pl, unroll=u, rows_per_wave=3))
                assert float((Y.double().cpu() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max())), (dt, pl, u)
                outs.append(Y)
        if ref_t is not None:
            Y = Y0d.clone()
            ops.spmm_raw(g.fwd[0], Xd, out=Y, accumulate=True, bias=bd, bias_scale=2.0, tanh=True)
            assert float((Y.double().cpu() - ref_t).abs().max()) <= tol, dt
        Y = torch.empty_like(Xd)
        ops.spmm_raw(g.fwd[0], Xd, out=Y)                                                      # auto variant, no accumulate
        assert float((Y.double().cpu() - (ref - Y0d.double().cpu())).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
def test_taps_bf16_matches_fp64_contraction():
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for F, G, K in ((32, 32, 3), (64, 64, 5), (64, 32, 2), (32, 0, 3)):
        R = 1000 + 7                                                                           # not a multiple of the 16-row tiles
        zh = torch.randn(1, R, 1, F).to(torch.bfloat16)
        zx = torch.randn(1, R, 1, G).to(torch.bfloat16) if G else None
        wB = (0.2 * torch.randn(F, 1, K, F)).to(torch.bfloat16)
        wA = (0.2 * torch.randn(F, 1, K - 1 if K > 2 else K, max(G, 1))).to(torch.bfloat16)    # Kin < Kst: zero taps beyond Kin
        u0, rest = ops.taps_bf16(zh.to(dev), zx.to(dev) if G else None, wA.to(dev), wB.to(dev))
        got = torch.cat([u0.unsqueeze(0), rest[:K - 1]], 0).double().cpu()
        for k in range(K):
            ref = zh.double().view(R, F) @ wB[:, 0, k].double().t()
            if G and k < wA.shape[2]:
                ref = ref + zx.double().view(R, G) @ wA[:, 0, k].double().t()
            err = float((got[k].view(R, F) - ref).abs().max())
            assert err <= 2e-2 * max(1.0, float(ref.abs().max())), (F, G, K, k, err)           # one bf16 rounding of an fp32 sum
