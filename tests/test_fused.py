"""Fused flagship path (bf16 step kernel): host-side ELL plan on CPU, kernel parity on the GPU."""
import os
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import gcrnn_oracle as orc
from gated_gcrnns_amd.graph import GraphOperator


def random_graph(N, density, seed, isolated=0):
    rng = np.random.default_rng(seed)
    S = (rng.random((N, N)) < density) * rng.uniform(0.2, 1.0, (N, N))
    if isolated:
        S[:isolated] = 0.0
        S[:, :isolated] = 0.0
    lam = np.max(np.abs(np.linalg.eigvals(S)))
    return (S / lam).reshape(1, N, N)


@pytest.mark.parametrize('kernel', ['step', 'wgrad'])
@pytest.mark.parametrize('N,density,iso', [(200, 0.05, 7), (1000, 0.01, 0), (1024, 0.004, 30), (17, 0.5, 0)])
def test_ell_plan_reproduces_shift_exactly(N, density, iso, kernel):
    """CPU: the degree-sorted sliced ELL is a permuted, padded but otherwise exact copy of CSR(S^T) -- for the wave count
    of the step kernels and for that of the weight-gradient kernel."""
    from gated_gcrnns_amd import _lib
    waves = int(_lib.lib.gcrnn_fused_wgrad_waves() if kernel == 'wgrad' else _lib.lib.gcrnn_fused_step_waves())
    S = random_graph(N, density, 5, iso)
    plan = GraphOperator(S).fused_plan(kernel=kernel)
    order = plan['order'].numpy()
    toff, ecol, eval_ = plan['tile_off'].numpy(), plan['ell_col'].numpy(), plan['ell_val'].numpy()
    assert sorted(order.tolist()) == list(range(N))
    deg = np.count_nonzero(S[0].T, axis=1)
    assert np.all(np.diff(deg[order]) <= 0)                         # descending degree
    tn = plan['tile_nodes'].numpy()
    assert sorted(tn.tolist()) == list(range(plan['npad']))         # every row (incl. padding rows) in exactly one slot
    # wave w owns `per` consecutive storage tiles = degree-ranked tiles w, w + waves, ...: per-wave work is balanced
    per_tile = np.diff(toff)
    per_wave = per_tile.reshape(waves, -1).sum(axis=1)
    assert per_wave.max() - per_wave.min() <= per_tile.max()       # round-robin over descending depths: at most one tile's worth
    assert toff[0] == 0 and np.all(np.diff(toff) % 4 == 0) and toff[-1] == plan['entries']
    # rebuild the dense operator from the ELL (slot -> node through tile_nodes) and compare with P = S^T
    P = np.zeros((plan['npad'], plan['npad']))
    for t in range(plan['npad'] // 16):
        for e in range(toff[t], toff[t + 1]):
            for r in range(16):
                v = eval_[e * 16 + r]
                if v != 0.0:
                    P[tn[t * 16 + r], ecol[e * 16 + r]] += v
    ref = S[0].T.astype(np.float32).astype(np.float64)
    assert np.array_equal(P[:N, :N], ref)
    assert not P[N:].any() and not P[:, N:].any()
    # padding waste stays small once rows are degree-sorted
    nnz = np.count_nonzero(S)
    assert plan['entries'] * 16 <= 1.35 * nnz + 64 * 16 * 4


def bf16_round(a):
    return torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).double().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,B,T,iso', [(1000, 64, 5, 3, 4, 0), (200, 32, 3, 9, 5, 11), (1024, 64, 2, 8, 2, 0),
                                           (37, 32, 5, 2, 3, 0), (1000, 64, 3, 17, 3, 40), (500, 64, 4, 5, 3, 0),
                                           (300, 32, 2, 6, 3, 5)])
def test_fused_step_matches_oracle(N, F, K, B, T, iso):
    """bf16 kernel vs the fp64 oracle evaluated on the same bf16-rounded inputs and weights.
    Remaining difference = bf16 rounding of the stored states h_t (2^-9 relative per step)."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    G = F
    S = random_graph(N, min(0.5, 10.0 / N), 21, iso)
    rng = np.random.default_rng(7)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.5 * rng.standard_normal((B, F, N)))
    torch.manual_seed(3)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16)
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    # the kernel keeps S in fp32
    S32 = S.astype(np.float32).astype(np.float64)
    Href = orc.ggcrnn_cell(params, S32, X, h0)
    cell = cell.to(dev)
    with torch.no_grad():
        assert cell._use_fused(torch.zeros(1, 1, G, N, dtype=torch.bfloat16, device=dev),
                               torch.zeros(1, F, N, dtype=torch.bfloat16, device=dev))
        H = cell(torch.tensor(X, dtype=torch.bfloat16, device=dev), torch.tensor(h0, dtype=torch.bfloat16, device=dev))
    assert H.dtype == torch.bfloat16 and tuple(H.shape) == (B, T, F, N)
    err = np.abs(H.double().cpu().numpy() - Href)
    # one-step error: only the final bf16 store (|h| <= 1 -> <= 2^-9); later steps add propagated rounding
    assert err[:, 0].max() <= 2.0e-3, err[:, 0].max()
    assert err.max() <= 5.0e-3, err.max()
    assert err.mean() <= 1.0e-3, err.mean()


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,B,T', [(1000, 64, 5, 5, 4), (200, 32, 3, 9, 3), (1000, 64, 3, 70, 2), (200, 32, 5, 4, 3),
                                       (400, 64, 2, 4, 3), (400, 64, 4, 4, 2), (150, 32, 2, 4, 2)])
def test_fused_time_gated_matches_oracle(N, F, K, B, T):
    """Time gating on the fused path: gate pre-pass (one launch over all (t, b)) + gated recurrence vs the fp64 oracle
    on bf16-rounded operands. Non-zero h0 exercises the gates-read-h0 rule (graphML.py:2362, 2370)."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    G = F
    S = random_graph(N, min(0.5, 10.0 / N), 23)
    rng = np.random.default_rng(9)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.5 * rng.standard_normal((B, F, N)))
    torch.manual_seed(5)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, True, None, 1, True)
    cell.addGSO(torch.tensor(S))
    with torch.no_grad():                               # make the scalar gates vary: larger MLP weights than the default init
        cell.MLP_in[0].weight.mul_(8.0)
        cell.MLP_forget[0].weight.mul_(8.0)
    cell = cell.to(torch.bfloat16)
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    S32 = S.astype(np.float32).astype(np.float64)
    Href = orc.ggcrnn_cell(params, S32, X, h0, True, None)
    cell = cell.to(dev)
    with torch.no_grad():
        Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
        hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
        assert cell._use_fused(Xd, hd)
        H = cell(Xd, hd)
    err = np.abs(H.double().cpu().numpy() - Href)
    assert err[:, 0].max() <= 6.0e-3, err[:, 0].max()
    assert err.max() <= 5.0e-3 and err.mean() <= 1.0e-3, (err.max(), err.mean())


@pytest.mark.gpu
def test_fused_forward_hipgraph_replay_is_bit_identical():
    """The captured hipGraph (pack -> T launches -> unpack) replays to exactly the eager result, also on new inputs."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd.ops import FusedForwardGraph
    dev = torch.device('cuda:0')
    N, F, K, B, T = 300, 32, 3, 6, 4
    S = random_graph(N, 0.03, 4)
    torch.manual_seed(1)
    for tg in (False, True):
        cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, None, 1, True)
        cell.addGSO(torch.tensor(S))
        cell = cell.to(dev).to(torch.bfloat16)
        runner = FusedForwardGraph(cell, B, T)
        for seed in (0, 1):
            g = torch.Generator(device=dev); g.manual_seed(seed)
            X = torch.randn(B, T, F, N, device=dev, generator=g).to(torch.bfloat16)
            h0 = (0.3 * torch.randn(B, F, N, device=dev, generator=g)).to(torch.bfloat16)
            with torch.no_grad():
                ref = cell(X, h0)
            out = runner(X, h0)
            assert torch.equal(out, ref)
        # (r4) the graph reads the cached packed parameters: an in-place update (optimiser step, load_state_dict) must be followed
        with torch.no_grad():
            cell.weight_B.mul_(0.5)
            cell.bias.add_(0.125)
            ref2 = cell(X, h0)
        out2 = runner(X, h0)
        assert torch.equal(out2, ref2) and not torch.equal(ref2, ref)
        assert torch.equal(runner(), ref2)                       # ... and plain replays stay on the new capture


def _bwd_reference(S, params, X, h0, dH):
    """fp32 autograd on the composed HIP path (itself pinned to the reference's gradients by the G4 goldens)."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    F, _, K, G = params['weight_A'].shape
    cell = gml.GGCRNNCell(G, F, K, params['weight_B'].shape[2], torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell.load_state_dict({k: torch.tensor(v, dtype=torch.float32) for k, v in params.items()})
    cell = cell.to(dev)
    Xd = torch.tensor(X, dtype=torch.float32, device=dev)
    hd = torch.tensor(h0, dtype=torch.float32, device=dev, requires_grad=True)
    H = cell(Xd, hd)
    (H * torch.tensor(dH, dtype=torch.float32, device=dev)).sum().backward()
    return cell, hd.grad.detach().cpu().numpy(), H.detach()


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,B,T', [(1000, 64, 5, 4, 5), (200, 32, 3, 9, 4), (600, 64, 3, 3, 3), (520, 64, 2, 3, 3),
                                       (1000, 64, 4, 2, 3), (1000, 32, 5, 3, 3), (304, 32, 2, 3, 3)])
def test_fused_backward_data_chain(N, F, K, B, T):
    """BPTT data-gradient chain on the fused kernel (bf16) vs fp32 autograd: d loss / d h0 depends on every step."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd import ops, _lib
    dev = torch.device('cuda:0')
    G = F
    S = random_graph(N, min(0.5, 10.0 / N), 31)                       # directed: the adjoint graph differs from the forward one
    rng = np.random.default_rng(3)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.5 * rng.standard_normal((B, F, N)))
    dH = bf16_round(rng.standard_normal((B, T, F, N)))
    torch.manual_seed(7)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16)
    params = {k: v.detach().float().numpy() for k, v in cell.state_dict().items()}
    _, dh0_ref, _ = _bwd_reference(S, params, X, h0, dH)
    cell = cell.to(dev)
    with torch.no_grad():
        Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
        hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
        hs_all, plan, _Hu = ops.fused_cell_forward(Xd, hd, cell.weight_A, cell.weight_B, cell.bias, cell.graph, return_states=True)
        hs = hs_all[1:]                                              # slot 0 holds h0
        npad = plan['npad']
        dHd = torch.tensor(dH, dtype=torch.bfloat16, device=dev)
        dHs = torch.empty((T, B, npad, F), dtype=torch.bfloat16, device=dev)
        _lib.check(_lib.lib.gcrnn_pack_seq_major(_lib.BF16, ops._p(dHd), ops._p(dHs), B, T, F, N, npad, None, ops._stream()), 'pack')
        dpre, dh0s = ops.fused_backward_data(dHs, hs, cell.weight_B, cell.graph)
        dh0 = torch.empty((B, 1, F, N), dtype=torch.bfloat16, device=dev)
        _lib.check(_lib.lib.gcrnn_unpack_seq_major(_lib.BF16, ops._p(dh0s), ops._p(dh0), B, 1, F, N, npad, None, ops._stream()), 'unpack')
    got = dh0.float().cpu().numpy().reshape(B, F, N)
    scale = np.abs(dh0_ref).max()
    err = np.abs(got - dh0_ref)
    assert scale > 0 and err.max() <= 4e-2 * scale and err.mean() <= 4e-3 * scale, (err.max() / scale, err.mean() / scale)
    # weight gradients: all T*B items in one launch
    with torch.no_grad():
        H = torch.empty((B, T, F, N), dtype=torch.bfloat16, device=dev)
        _lib.check(_lib.lib.gcrnn_unpack_seq_major(_lib.BF16, ops._p(hs), ops._p(H), B, T, F, N, npad, None, ops._stream()), 'unpack')
        dW, dbs = ops.fused_backward_weight(dpre, Xd, H, hd, cell.graph, F, G, K, want_bias=True)
        dW, db = dW.cpu().numpy(), dbs.cpu().numpy()              # the kernel returns the bias gradient itself (2 sum dpre)
        assert np.allclose(db, 2.0 * dpre.float().sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-3, atol=1e-3 * np.abs(db).max())
    ref_cell = _bwd_reference(S, params, X, h0, dH)[0]
    gB = ref_cell.weight_B.grad[:, 0].cpu().numpy()          # [F][K][F]
    gA = ref_cell.weight_A.grad[:, 0].cpu().numpy()          # [F][K][G]
    gb = ref_cell.bias.grad.view(-1).cpu().numpy()
    for name, got_w, ref_w in (('weight_B', dW[:, :, :F], gB), ('weight_A', dW[:, :, F:], gA), ('bias', db, gb)):
        sc = np.abs(ref_w).max()
        e = np.abs(got_w - ref_w)
        assert e.max() <= 2e-2 * sc, (name, e.max() / sc, e.mean() / sc)


@pytest.mark.gpu
@pytest.mark.parametrize('master', [torch.float32, torch.bfloat16])
def test_fused_training_autograd_end_to_end(master):
    """cell(X, h0).backward() on bf16 activations runs forward + BPTT on the fused kernels (parameters bf16 or fp32
    master weights) and reproduces the fp32 autograd gradients of the composed path."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, F, K, B, T = 1000, 64, 5, 6, 6
    G = F
    S = random_graph(N, 0.01, 41)
    rng = np.random.default_rng(5)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.3 * rng.standard_normal((B, F, N)))
    target = bf16_round(rng.standard_normal((B, T, F, N)))
    torch.manual_seed(11)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).to(torch.float32)                  # bf16-representable values, fp32 storage
    params = {k: v.detach().float().numpy() for k, v in cell.state_dict().items()}
    # reference: fp32 composed path, loss = mean |H - target|
    ref = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    ref.addGSO(torch.tensor(S))
    ref.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    ref = ref.to(dev)
    Hr = ref(torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev))
    torch.nn.functional.l1_loss(Hr, torch.tensor(target, dtype=torch.float32, device=dev)).backward()
    # fused
    cell = cell.to(dev).to(master)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    assert cell._use_fused_training(Xd, hd)
    H = cell(Xd, hd)
    assert H.dtype == torch.bfloat16 and H.requires_grad
    loss = torch.nn.functional.l1_loss(H.float(), torch.tensor(target, dtype=torch.float32, device=dev))
    loss.backward()
    assert abs(float(loss) - float(torch.nn.functional.l1_loss(Hr, torch.tensor(target, dtype=torch.float32, device=dev)))) < 2e-3
    for name in ('weight_A', 'weight_B', 'bias'):
        g, gr = getattr(cell, name).grad.float().cpu().numpy(), getattr(ref, name).grad.cpu().numpy()
        assert g.shape == gr.shape
        sc = np.abs(gr).max()
        # L1 loss: dH = sign(H - target)/count flips where bf16 rounding moves H across the target: allow a few %
        assert np.abs(g - gr).max() <= 6e-2 * sc and np.abs(g - gr).mean() <= 1e-2 * sc, (name, np.abs(g - gr).max() / sc)


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,B,T', [(1500, 32, 3, 4, 3), (2100, 64, 2, 2, 2), (1100, 8, 4, 3, 3)])
def test_bf16_streaming_path_for_graphs_beyond_the_fused_kernel(N, F, K, B, T):
    """N > 1024: bf16 inference runs in Horner form on the bf16 accumulate-SpMM (gcrnn_spmm GCRNN_BF16: bf16 rows, fp32
    weights / accumulation) -- vs the fp64 oracle on the same bf16-rounded operands."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    G = F
    S = random_graph(N, 8.0 / N, 3)
    rng = np.random.default_rng(11)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.5 * rng.standard_normal((B, F, N)))
    torch.manual_seed(1)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    params = {k: bf16_round(v.detach().numpy()) for k, v in cell.state_dict().items()}
    cell = cell.to(dev).to(torch.bfloat16)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        assert not cell._use_fused(Xd, hd) and cell._use_horner(Xd, hd)
        H = cell(Xd, hd).double().cpu().numpy()
    ref = orc.ggcrnn_cell(params, S, X, h0)
    err = np.abs(H - ref)
    assert err.max() <= 5e-2 and err.mean() <= 4e-3, (err.max(), err.mean())


def _gated_pair(N, F, K, S, seed, dev, master=torch.float32):
    """A time-gated cell with bf16-representable parameters and its fp32 composed-path twin."""
    import gated_gcrnns_amd.Utils.graphML as gml
    torch.manual_seed(seed)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, True, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).to(torch.float32)
    ref = gml.GGCRNNCell(F, F, K, K, torch.tanh, True, None, 1, True)
    ref.addGSO(torch.tensor(S))
    ref.load_state_dict(cell.state_dict())
    return cell.to(dev).to(master), ref.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,B,T,master', [(1000, 64, 5, 6, 5, torch.float32), (200, 32, 3, 9, 4, torch.float32),
                                              (600, 64, 3, 4, 3, torch.bfloat16), (1000, 64, 2, 3, 3, torch.float32),
                                              (304, 32, 5, 4, 3, torch.float32), (200, 32, 2, 5, 3, torch.float32),
                                              (1000, 64, 4, 3, 3, torch.float32),      # every (K, F) the fused kernels are built for
                                              (200, 32, 3, 1, 1, torch.float32), (1024, 64, 2, 2, 1, torch.float32)])   # T = 1, B = 1, N = NPad
def test_fused_time_gated_training_matches_composed_autograd(N, F, K, B, T, master):
    """Time-gated cell, bf16 activations: forward, both gate sub-networks and the whole BPTT run on the fused kernels
    (ops.fused_cell_train) and reproduce the fp32 autograd gradients of the composed path for EVERY trained parameter --
    the cell's taps and bias, the gate cells' taps and biases, the gates' read-out layers (reference graphML.py:2357-2374,
    2420-2423). h0 != 0: the gates read (x_t, h0)."""
    dev = torch.device('cuda:0')
    S = random_graph(N, min(0.5, 10.0 / N), 43)
    rng = np.random.default_rng(9)
    X = bf16_round(rng.standard_normal((B, T, F, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    dH = bf16_round(rng.standard_normal((B, T, F, N)))
    cell, ref = _gated_pair(N, F, K, S, 17, dev, master)
    # make the gates informative: the reference init gives logits of O(10) magnitude spread, fine; nothing to rescale
    Hr = ref(torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev))
    dHd = torch.tensor(dH, dtype=torch.float32, device=dev)
    (Hr * dHd).sum().backward()
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    assert cell._use_fused_training(Xd, hd)
    H = cell(Xd, hd)
    assert H.dtype == torch.bfloat16 and H.requires_grad
    err = (H.detach().float() - Hr.detach()).abs()
    assert float(err.max()) <= 6e-3 and float(err.mean()) <= 1.2e-3, (float(err.max()), float(err.mean()))
    (H.float() * dHd).sum().backward()
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    assert any(n.startswith('GFL_in.') for n in names) and any(n.startswith('MLP_forget.') for n in names)
    got = dict(cell.named_parameters())
    for n, p in ref.named_parameters():
        if p.grad is None:                       # GFL_out / MLP_out: built, never used (graphML.py:2280-2290)
            assert got[n].grad is None, n
            continue
        g, gr = got[n].grad.float().cpu().numpy(), p.grad.cpu().numpy()
        assert g.shape == gr.shape, n
        sc = np.abs(gr).max()
        e = np.abs(g - gr)
        # bf16 activations: a few % on single entries; the mean bound only means something where there is something to average
        assert sc > 0 and e.max() <= 4e-2 * sc and (e.size < 16 or e.mean() <= 8e-3 * sc), (n, e.max() / sc, e.mean() / sc)


@pytest.mark.gpu
def test_fused_gated_cell_gate_gradients_match_composed_gates():
    """The cell alone with the gates as differentiable inputs (ops.fused_cell_train_with_gates): d loss / d gi, d gf of the
    fused BPTT against autograd through the composed fp32 path fed the same gate values."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    N, F, K, B, T = 1000, 64, 4, 5, 4
    S = random_graph(N, 0.01, 47)
    rng = np.random.default_rng(13)
    X = bf16_round(rng.standard_normal((B, T, F, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    dH = torch.tensor(bf16_round(rng.standard_normal((B, T, F, N))), dtype=torch.float32, device=dev)
    gi0 = torch.tensor(rng.uniform(0.1, 0.9, (T, B)), dtype=torch.float32, device=dev)
    gf0 = torch.tensor(rng.uniform(0.1, 0.9, (T, B)), dtype=torch.float32, device=dev)
    torch.manual_seed(19)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).to(torch.float32).to(dev)
    # composed fp32 reference with explicit scalar gates: h_t = tanh(gi (A x_t + b) + gf (B h_{t-1} + b))
    gi_r, gf_r = gi0.clone().requires_grad_(True), gf0.clone().requires_grad_(True)
    Xn = ops.pack_node_major(torch.tensor(X, dtype=torch.float32, device=dev))
    h = ops.pack_node_major(torch.tensor(h0, dtype=torch.float32, device=dev).reshape(B, 1, F, N))
    ya = ops.lsigf_node_major(Xn, cell.weight_A, cell.bias, cell.graph, 1.0)
    Hs = []
    for t in range(T):
        yb = ops.lsigf_node_major(h, cell.weight_B, cell.bias, cell.graph, 1.0)
        h = torch.tanh(gi_r[t].view(1, 1, B, 1) * ya[t:t + 1] + gf_r[t].view(1, 1, B, 1) * yb)
        Hs.append(h)
    Hr = ops.unpack_node_major(torch.cat(Hs, dim=0))
    (Hr * dH).sum().backward()
    ref_grads = {n: p.grad.clone() for n, p in cell.named_parameters()}
    cell.zero_grad()
    gi_f, gf_f = gi0.clone().requires_grad_(True), gf0.clone().requires_grad_(True)
    H = ops.fused_cell_train_with_gates(torch.tensor(X, dtype=torch.bfloat16, device=dev), torch.tensor(h0, dtype=torch.bfloat16, device=dev),
                                        cell.weight_A, cell.weight_B, cell.bias, cell.graph, gi_f, gf_f)
    (H.float() * dH).sum().backward()
    for name, g, gr in (('gi', gi_f.grad, gi_r.grad), ('gf', gf_f.grad, gf_r.grad)):
        sc = float(gr.abs().max())
        e = (g - gr).abs()
        assert float(e.max()) <= 3e-2 * sc, (name, float(e.max()) / sc)
    for n, p in cell.named_parameters():
        sc = float(ref_grads[n].abs().max())
        e = (p.grad - ref_grads[n]).abs()
        assert float(e.max()) <= 3e-2 * sc, (n, float(e.max()) / sc)


@pytest.mark.gpu
def test_fused_gate_prepass_splits_launches_beyond_32bit_offsets():
    """T*B*NPad*F*2 bytes > 2 GiB: the all-items gate pre-pass is split over whole time steps; gates equal those of
    single-step calls."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    N, F, K, B, T = 1000, 64, 2, 520, 32                      # 16640 items x 128 KiB per item row block
    S = random_graph(N, 0.01, 53)
    torch.manual_seed(23)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, True, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev).to(torch.bfloat16)
    X = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
    h0 = (0.3 * torch.randn(B, F, N, device=dev)).to(torch.bfloat16)
    g = cell._fused_gates()['in']
    with torch.no_grad():
        xs, hs_all = ops.fused_pack_inputs(X, h0, cell.graph)
        full = ops.fused_time_gate(xs, hs_all[:1], *g, cell.graph, N)
        for t in (0, 30, 31):
            one = ops.fused_time_gate(xs[t:t + 1], hs_all[:1], *g, cell.graph, N)
            assert torch.equal(one[0], full[t]), t
    assert float(full.std()) > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('head,tg', [('multipMlp', False), ('multipMlp', True), ('oneMlp', False), ('classification', True)])
def test_fused_cell_inside_the_models_bf16_activations_fp32_master_weights(head, tg):
    """GatedGCRNNforRegression / forClassification (reference architectures.py:1405-1859) with bf16 inputs and fp32 master
    weights: the cell trains on the fused kernels, the per-node head on its bf16 kernel, the dense heads in the parameters'
    dtype; outputs and every parameter gradient against the same model run in fp32 on the composed path."""
    import gated_gcrnns_amd.Modules.architectures as archit
    dev = torch.device('cuda:0')
    N, F, K, B, T = 200, 32, 3, 6, 4
    S = random_graph(N, 0.05, 61)
    rng = np.random.default_rng(15)
    X = bf16_round(rng.standard_normal((B, T, F, N)))
    h0 = np.zeros((B, F, N))

    def build():
        torch.manual_seed(29)
        if head == 'classification':
            m = archit.GatedGCRNNforClassification(F, F, K, K, torch.tanh, torch.nn.ReLU, [5], S[0], True, time_gating=tg)
        else:
            m = archit.GatedGCRNNforRegression(F, F, K, K, torch.tanh, torch.nn.ReLU, [1], S[0], True, time_gating=tg,
                                               spatial_gating=None, mlpType=head)
        return m.to(torch.bfloat16).to(torch.float32).to(dev)             # bf16-representable values, fp32 storage

    ref, mod = build(), build()
    yr = ref(torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev))
    w = torch.tensor(bf16_round(rng.standard_normal(tuple(yr.shape))), dtype=torch.float32, device=dev)
    (yr * w).sum().backward()
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    assert mod.stateGCRNN._use_fused_training(Xd, hd)
    y = mod(Xd, hd)
    assert tuple(y.shape) == tuple(yr.shape)
    sc = float(yr.abs().max())
    assert float((y.float() - yr).abs().max()) <= 3e-2 * sc, float((y.float() - yr).abs().max()) / sc
    (y.float() * w).sum().backward()
    got = dict(mod.named_parameters())
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert got[n].grad is None, n
            continue
        g, gr = got[n].grad.float(), p.grad
        s = float(gr.abs().max())
        if s == 0.0:                                   # h0 = 0: the gate cells' state taps get exactly no gradient, in both
            assert float(g.abs().max()) == 0.0, n
            continue
        e = (g - gr).abs()
        assert float(e.max()) <= 5e-2 * s and (e.numel() < 16 or float(e.mean()) <= 1e-2 * s), (n, float(e.max()) / s, float(e.mean()) / s)


@pytest.mark.gpu
@pytest.mark.parametrize('N,G,F,K,B,T,tg', [(1000, 1, 64, 5, 4, 4, False), (1000, 1, 64, 3, 3, 3, True), (200, 3, 32, 3, 5, 4, False),
                                            (520, 20, 64, 2, 3, 3, True), (304, 32, 64, 4, 3, 3, False), (200, 1, 32, 5, 4, 3, True)])
def test_fused_cell_with_few_input_features(N, G, F, K, B, T, tg):
    """The reference drivers feed ONE input feature per node (kStepPredGRNNs.py:220): G < 32 runs on the fused kernels with the
    x operand zero-padded to one 32-feature MFMA step. Forward against the fp64 oracle, training against fp32 autograd on
    the composed path, for the un-gated and the time-gated cell."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    S = random_graph(N, min(0.5, 10.0 / N), 71)
    rng = np.random.default_rng(17)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    dH = bf16_round(rng.standard_normal((B, T, F, N)))
    torch.manual_seed(31)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).to(torch.float32)
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    Href = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X, h0, tg, None)
    ref = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
    ref.addGSO(torch.tensor(S))
    ref.load_state_dict(cell.state_dict())
    ref = ref.to(dev)
    dHd = torch.tensor(dH, dtype=torch.float32, device=dev)
    (ref(torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev)) * dHd).sum().backward()
    cell = cell.to(dev)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        assert cell.to(torch.bfloat16)._use_fused(Xd, hd)
        Hi = cell(Xd, hd)                                        # inference kernels (bf16 parameters)
        cell = cell.to(torch.float32)
    err = np.abs(Hi.double().cpu().numpy() - Href)
    # (G = 1: the reference init U(+-1/sqrt(G K)) gives taps of +-0.45 -- pre-activations, and with them the rounding of the stored
    # states, are larger than at G = 64: measured max 1.5e-2 / mean 1.0e-3)
    assert err.max() <= 2.5e-2 and err.mean() <= 1.5e-3, (err.max(), err.mean())
    assert cell._use_fused_training(Xd, hd)
    H = cell(Xd, hd)
    (H.float() * dHd).sum().backward()
    got = dict(cell.named_parameters())
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert got[n].grad is None, n
            continue
        g, gr = got[n].grad.float(), p.grad
        assert g.shape == gr.shape, n
        s = float(gr.abs().max())
        e = (g - gr).abs()
        assert s > 0 and float(e.max()) <= 4e-2 * s and (e.numel() < 16 or float(e.mean()) <= 8e-3 * s), (n, float(e.max()) / s, float(e.mean()) / s)


@pytest.mark.gpu
def test_fused_forward_with_side_stream_pack_is_bit_identical(monkeypatch):
    """GCRNN_FUSED_OVERLAP=1: steps 1.. of the input are packed on a second stream beside the step kernels, every step launch
    waits for its event (gcrnn_pack_seq_major_steps, step_events of gcrnn_fused_forward_bf16). Same bits as the plain order."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, F, K, B, T = 1000, 64, 3, 40, 9
    S = random_graph(N, 0.01, 83)
    torch.manual_seed(37)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev).to(torch.bfloat16)
    X = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
    h0 = (0.3 * torch.randn(B, F, N, device=dev)).to(torch.bfloat16)
    with torch.no_grad():
        ref = cell(X, h0)
        monkeypatch.setenv('GCRNN_FUSED_OVERLAP', '1')
        for blocks in ('256', '7'):
            monkeypatch.setenv('GCRNN_PACK_BLOCKS', blocks)
            out = cell(X, h0)
            torch.cuda.synchronize()
            assert torch.equal(out, ref), blocks


@pytest.mark.gpu
@pytest.mark.parametrize('N,tg', [(1000, False), (204, True), (1000, True)])
def test_fused_last_state_only_matches_full_forward(N, tg):
    """last_only = True (the classification models' read-out, reference architectures.py:1841-1850): the fused forward skips
    the user-layout store of every step but the last (N % 8 == 0) or unpacks only the last state; same bits as H[:, -1]."""
    import gated_gcrnns_amd.Utils.graphML as gml
    import gated_gcrnns_amd.Modules.architectures as archit
    dev = torch.device('cuda:0')
    F, K, B, T = 64, 3, 5, 6
    S = random_graph(N, min(0.5, 10.0 / N), 91)
    torch.manual_seed(41)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev).to(torch.bfloat16)
    X = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)
    h0 = (0.3 * torch.randn(B, F, N, device=dev)).to(torch.bfloat16)
    with torch.no_grad():
        full = cell(X, h0)
        last = cell(X, h0, last_only=True)
    assert tuple(last.shape) == (B, 1, F, N) and torch.equal(last[:, 0], full[:, -1])
    torch.manual_seed(43)
    m = archit.GatedGCRNNforClassification(F, F, K, K, torch.tanh, torch.nn.ReLU, [7], S[0], True, time_gating=tg).to(dev).to(torch.bfloat16)
    with torch.no_grad():
        logits = m(X, h0)
        H = m.stateGCRNN(X, h0)
        ref = m.outputNN(H.select(1, -1).reshape(-1, F * N))
    assert tuple(logits.shape) == (B, 7) and torch.equal(logits, ref)


@pytest.mark.gpu
@pytest.mark.parametrize('tg', [False, True])
def test_fused_full_size_batch_independence_and_determinism(tg, monkeypatch):
    """BASELINE configs[1] at the bench's full size (N=1000, K=5, T=32, G=F=64, B=256; the oracle cannot run this in seconds):
    size-independent properties instead -- (1) every sequence's states are the same bits whether it runs in the batch of 256
    (one workgroup per CU walking the sequences) or in a batch of 8; (2) two runs agree bit for bit (no atomics on the path);
    (3) the first sequences match the fp64 oracle on a single step."""
    import bench
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, K, T, F, B = 1000, 5, 32, 64, 256
    S = bench.sbm_graph(N)
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16)
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    cell = cell.to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    X = torch.randn(B, T, F, N, device=dev, generator=gen).to(torch.bfloat16)
    h0 = (0.3 * torch.randn(B, F, N, device=dev, generator=gen)).to(torch.bfloat16)
    with torch.no_grad():
        H = cell(X, h0)
        H2 = cell(X, h0)
        monkeypatch.setenv('GCRNN_SEQ32_MIN_B', '1')      # (the batch of 8 on the kernel family the batch of 256 runs on: the wide sequence-resident kernel
        Hs = cell(X[100:108].contiguous(), h0[100:108].contiguous())      #  and the 16-feature kernels agree to bf16 rounding, not bit for bit: tests/test_wide.py)
        monkeypatch.delenv('GCRNN_SEQ32_MIN_B')
    assert torch.equal(H, H2)
    assert torch.equal(H[100:108], Hs)
    Href = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X[:2, :1].double().cpu().numpy(),
                           h0[:2].double().cpu().numpy(), tg, None)
    err = np.abs(H[:2, :1].double().cpu().numpy() - Href)
    assert err.max() <= 4e-3, err.max()


@pytest.mark.gpu
@pytest.mark.parametrize('tg', [False, True])
def test_fused_full_size_training_gradients_add_over_the_batch(tg, monkeypatch):
    """Full bench size (B=256, T=32, N=1000, K=5, G=F=64), fused BPTT: sequences are independent, so the parameter gradients of
    the batch equal the sum of those of its two halves (linearity over the batch; the weight-gradient kernel accumulates
    with float atomics, hence a tolerance instead of bit equality). Covers the grid walk over 8192 (t, b) items."""
    import bench
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, K, T, F, B = 1000, 5, 32, 64, 256
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(bench.sbm_graph(N)))
    cell = cell.to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    X = torch.randn(B, T, F, N, device=dev, generator=gen).to(torch.bfloat16)
    W = torch.randn(B, T, F, N, device=dev, generator=gen).to(torch.bfloat16)
    h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)

    def grads(sl):
        cell.zero_grad()
        assert cell._use_fused_training(X[sl], h0[sl])
        (cell(X[sl].contiguous(), h0[sl].contiguous()).float() * W[sl].float()).sum().backward()
        return {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}

    monkeypatch.setenv('GCRNN_SEQ32_MIN_B', '1')      # (un-gated: the half batches' forward on the kernel family the full batch's runs on -- the families agree to bf16 rounding only)
    full, a, b = grads(slice(0, B)), grads(slice(0, B // 2)), grads(slice(B // 2, B))
    assert set(full) == set(a) == set(b) and len(full) >= (11 if tg else 3)
    for n, g in full.items():
        s = a[n] + b[n]
        sc = float(g.abs().max())
        assert sc > 0 or float(s.abs().max()) == 0, n
        assert float((g - s).abs().max()) <= 2e-3 * sc + 1e-12, (n, float((g - s).abs().max()) / max(sc, 1e-30))


def _grad_scale_and_bounds(name, grads, mx, mean):
    """Scale and (max, mean) bounds for one parameter's gradient error. Attention mixers: their gradients are sums of
    softmax-backward terms that cancel within every support row (sum_n alpha (d alpha - R) = 0), so the bf16 rounding of z and of
    the incoming gradient shows there first, at the same ABSOLUTE level in both mixers -- measured on the larger of the two (as
    the scalar biases are measured on their sibling weights); attention parameters get a wider single-entry bound."""
    sc = float(np.abs(grads[name]).max())
    if name.endswith('.mixer'):
        sc = max(float(np.abs(v).max()) for k, v in grads.items() if k.endswith('.mixer'))
        return sc, max(mx, 1e-1), 2.5e-2
    if '_attention.' in name:
        return sc, max(mx, 8e-2), mean
    return sc, mx, mean


def _load_g9_gates():
    import json
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g9_gradient_gates.json')
    if os.environ.get('GCRNN_TOL_REPORT') or not os.path.exists(f):      # (a measuring run uses the class-level gates)
        return {}
    return json.load(open(f))


_G9_GATES = _load_g9_gates()


def _tol_report(line):
    """GCRNN_TOL_REPORT=<file>: append the measured error ratios (the gates sit at <= 2x the worst measured: tools/tolerance_probe.py)."""
    f = os.environ.get('GCRNN_TOL_REPORT')
    if f:
        with open(f, 'a') as fh:
            fh.write(line + '\n')


def _g9_cell(g, tg, sg, dev):
    """The fixture's cell (parameters by state_dict key, fp32 master weights holding bf16-representable values) and inputs."""
    import gated_gcrnns_amd.Utils.graphML as gml
    N, T, G, F, K, B = (int(v) for v in g['shape'])
    S = np.zeros((1, N, N))
    S[0, g['coo_row'].astype(np.int64), g['coo_col'].astype(np.int64)] = g['coo_val'].astype(np.float64)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, sg, 1, True)
    cell.addGSO(torch.tensor(S))
    cell.load_state_dict({k: torch.tensor(v) for k, v in g['params'].items()})
    return cell.float().to(dev), S


@pytest.mark.gpu
@pytest.mark.parametrize('name,tg,sg', [('none', False, None), ('time', True, None), ('node', False, 'node'), ('edge', False, 'edge'),
                                        ('time_edge', True, 'edge')])
@pytest.mark.parametrize('loss', ['sum', 'l1'])
def test_fused_bptt_matches_reference_autograd_fixture(golden, name, tg, sg, loss):
    """G9: the fused forward + BPTT against gradients the REFERENCE's autograd produced (tests/golden/make_golden.py
    g9_fused_bptt: Utils/graphML.py:2336-2427 under torch autograd, fp64) at a fused-supported shape with bf16-representable
    operands, non-zero h0 and a directed weighted S -- every trained parameter incl. the gate sub-networks, and h0 where the
    fused path produces it. Tolerances are those of bf16 activations, relative to each gradient's max."""
    g = golden('g9_fused_' + name)
    dev = torch.device('cuda:0')
    cell, S = _g9_cell(g, tg, sg, dev)
    X = torch.tensor(g['X'], dtype=torch.bfloat16, device=dev)
    h0 = torch.tensor(g['h0'], dtype=torch.bfloat16, device=dev, requires_grad=(sg is None))
    assert float((X.float().cpu() - torch.tensor(g['X'])).abs().max()) == 0.0          # operands are bf16-exact
    X.requires_grad_(sg is None)          # d loss / d X and d loss / d h0 on the fused path: un-gated (round 3) and time-gated (round 4: through the gate cells too), G == F
    if not cell._use_fused_training(X, h0):
        pytest.skip('no fused training kernels for the %s-gated cell' % name)
    H = cell(X, h0)
    assert H.dtype == torch.bfloat16
    err = (H.detach().float().cpu() - torch.tensor(g['H'])).abs()
    assert float(err.max()) <= 1.2e-2 and float(err.mean()) <= 1.5e-3, (float(err.max()), float(err.mean()))
    if loss == 'sum':
        H.float().sum().backward()
        want, want_h0 = g['grad_sum'], g['grad_sum_h0']
    else:
        torch.nn.functional.l1_loss(H.float(), torch.tensor(g['target'], device=dev)).backward()
        want, want_h0 = g['grad_l1'], g['grad_l1_h0']
    got = dict(cell.named_parameters())
    checked = 0
    for k, gr in want.items():
        gg = got[k].grad
        assert gg is not None, k
        gg = gg.float().cpu().numpy()
        e = np.abs(gg - gr)
        # L1: dH = sign(H - target) / count flips where the bf16 rounding of H crosses the target -> looser single-entry bound
        sc, mx, mn = _grad_scale_and_bounds(k, want, 6e-2 if loss == 'l1' else 4e-2, 1e-2)
        if loss == 'sum':
            # round 4: gates at <= 2x the worst measured ratio (GCRNN_TOL_REPORT run, profiles/r04_gradient_tolerances.txt): weights 6.5e-3 of
            # their max / 1.1e-3 mean, attention 6.2e-3 / 1.9e-3, (few-element) biases 1.8e-2. The L1 loss keeps 6e-2 (measured 3.6e-2; attention
            # 5.4e-2 against 8e-2): sign flips of dH where the bf16 rounding of H crosses the target.
            if k.endswith('bias'):
                mx, mn = 3.6e-2, 3.6e-2
            elif '_attention.' in k:
                mx, mn = 1.3e-2, 4e-3
            else:
                mx, mn = 1.3e-2, 2.2e-3
        # round 5: ONE gate per parameter at 2x what was measured for it (tests/golden/g9_gradient_gates.json, written by
        # tools/make_gradient_gates.py from a GCRNN_TOL_REPORT run on an MI355X; the kernels are bit-reproducible, so the measured ratios are
        # the same on every run); the class-level gates above stay as the ceiling and for parameters the table does not list
        pg = _G9_GATES.get('g9 %s %s %s' % (name, loss, k))
        if pg is not None:
            mx, mn = min(mx, max(2.0 * pg[0], 2e-4)), min(mn, max(2.0 * pg[1], 1e-4))
        _tol_report('g9 %s %s %s max %.3e mean %.3e (gate %.1e / %.1e)' % (name, loss, k, e.max() / sc, e.mean() / sc, mx, mn))
        assert sc > 0 and e.max() <= mx * sc and (e.size < 16 or e.mean() <= mn * sc), (k, e.max() / sc, e.mean() / sc)
        checked += 1
    assert checked >= 3
    for k, p in got.items():                          # parameters the reference leaves without gradient (unused output gate)
        if k not in want:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    if X.requires_grad:
        want_X = g['grad_sum_X' if loss == 'sum' else 'grad_l1_X']
        e = (X.grad.float().cpu() - torch.tensor(want_X)).abs()
        sc = float(np.abs(want_X).max())
        _tol_report('g9 %s %s dX max %.3e mean %.3e' % (name, loss, float(e.max()) / sc, float(e.mean()) / sc))
        # (measured: sum 3.0e-3 / 2.9e-4, L1 1.1e-1 / 2.4e-3 -- the L1 maximum is a single sign flip of dH)
        # (time-gated, round 4: the gate cells' input gradients add two more bf16 chains -- measured sum 7.3e-3 / 4.0e-4, L1 3.7e-2 / 1.1e-3)
        assert float(e.max()) <= (1.4e-1 if loss == 'l1' else (1.5e-2 if tg else 6e-3)) * sc and float(e.mean()) <= (5e-3 if loss == 'l1' else (8e-4 if tg else 6e-4)) * sc, ('dX', float(e.max()) / sc, float(e.mean()) / sc)
    if h0.requires_grad:
        e = (h0.grad.float().cpu() - torch.tensor(want_h0)).abs()
        sc = float(np.abs(want_h0).max())
        # d h0 has passed T bf16 dpre stores; with the L1 loss single entries also see sign flips of dH (measured 8 % of the max)
        _tol_report('g9 %s %s dh0 max %.3e mean %.3e' % (name, loss, float(e.max()) / sc, float(e.mean()) / sc))
        # (measured: sum 3.3e-3 / 2.3e-4, L1 8.1e-2 / 3.2e-3)
        assert float(e.max()) <= (1.4e-1 if loss == 'l1' else (1.5e-2 if tg else 6.6e-3)) * sc and float(e.mean()) <= (6.4e-3 if loss == 'l1' else (1e-3 if tg else 5e-4)) * sc, (float(e.max()) / sc, float(e.mean()) / sc)


@pytest.mark.gpu
@pytest.mark.parametrize('tg', [False, True])
def test_weight_gradients_are_bit_reproducible(tg):
    """Two runs of the same training step give the same bits for every parameter gradient: the weight-gradient kernels
    store per-workgroup partial sums that are added in a fixed order (no atomics) -- the reference (CPU) is run-to-run
    deterministic too. Fused bf16 path (un-gated / time-gated) and the composed fp32 path."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, F, K, B, T = 1000, 64, 5, 40, 6
    S = random_graph(N, 0.01, 77)
    rng = np.random.default_rng(3)
    X = torch.tensor(bf16_round(rng.standard_normal((B, T, F, N))), dtype=torch.bfloat16, device=dev)
    h0 = torch.zeros(B, F, N, dtype=torch.bfloat16, device=dev)
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)
    torch.manual_seed(5)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev)
    assert cell._use_fused_training(X, h0)

    def grads(c, x, h, tg_):
        c.zero_grad()
        torch.nn.functional.l1_loss(c(x, h).float(), tg_).backward()
        return {n: p.grad.clone() for n, p in c.named_parameters() if p.grad is not None}

    g1, g2 = grads(cell, X, h0, tgt), grads(cell, X, h0, tgt)
    assert g1.keys() == g2.keys() and len(g1) >= 3
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    # composed fp32 path (LSIGF autograd nodes: row-split partial sums of the tap GEMM, block partials of the bias sum)
    Bs = 4
    cellf = gml.GGCRNNCell(8, 12, 3, 3, torch.tanh, tg, None, 1, True)
    cellf.addGSO(torch.tensor(S))
    cellf = cellf.to(dev)
    Xf = torch.randn(Bs, T, 8, N, device=dev)
    hf = torch.zeros(Bs, 12, N, device=dev)
    tf = torch.randn(Bs, T, 12, N, device=dev)
    assert not cellf._use_fused_training(Xf, hf)
    g1, g2 = grads(cellf, Xf, hf, tf), grads(cellf, Xf, hf, tf)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,tg', [(1000, 64, 5, False), (1000, 64, 5, True), (600, 32, 3, False), (1008, 64, 2, False)])
def test_uniform_weight_graph_stream_matches_oracle_and_weighted_stream(N, F, K, tg, monkeypatch):
    """S = W / lambda_max of an UNWEIGHTED adjacency (the reference drivers' GSO, kStepPredGRNNs.py:768): the forward kernels
    drop the weight image and sum the gathered rows (acc = init + w * sum; padding entries aimed at zero rows). Checked against
    the fp64 oracle, and against the weighted stream on the same graph (not bit-identical: the sums associate differently)."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd.graph import GraphOperator
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(21)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
    B, T = 5, 3
    X = bf16_round(rng.standard_normal((B, T, F, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    torch.manual_seed(3)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    params = {k: bf16_round(v.detach().numpy()) for k, v in cell.state_dict().items()}
    S32 = S.astype(np.float32).astype(np.float64)
    ref = orc.ggcrnn_cell(params, S32, X, h0, tg, None)
    cell = cell.to(dev).to(torch.bfloat16)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    plan = cell.graph.fused_plan()
    assert (plan['uniform_w'] != 0.0) == (1024 - N >= 16)          # N = 1008: exactly 16 padding rows; fewer -> weighted image
    with torch.no_grad():
        assert cell._use_fused(Xd, hd)
        H = cell(Xd, hd).double().cpu().numpy()
    err = np.abs(H - ref)
    assert err.max() <= 5e-3 and err.mean() <= 1e-3, (err.max(), err.mean())
    # the same graph on the weighted stream
    monkeypatch.setenv('GCRNN_NO_UNIFORM', '1')
    g2 = GraphOperator(torch.tensor(S)).to(dev)
    assert g2.fused_plan()['uniform_w'] == 0.0
    cell.graph = g2
    for sub in ('GFL_in', 'GFL_forget'):
        if hasattr(cell, sub):
            getattr(cell, sub).graph = g2
    with torch.no_grad():
        H2 = cell(Xd, hd).double().cpu().numpy()
    assert np.abs(H2 - ref).max() <= 3e-2
    assert np.abs(H - H2).max() <= 1.6e-2                          # one bf16 ulp at |h| < 1 where a rounding flips


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T,zero_h0', [(1000, 64, 64, 5, 3, 5, False), (1000, 64, 1, 5, 2, 4, True), (600, 32, 32, 3, 5, 3, False),
                                                 (1008, 64, 32, 2, 700, 4, True)])
def test_fp32_accurate_time_gated_cell_matches_oracle_to_1e5(N, F, G, K, B, T, zero_h0, monkeypatch):
    """The time-gated cell at the north_star's 1e-5 on the fused kernels (ops.fused_cell_forward_x3_gated): both gates from the x3 step on
    (x_t, h0) + the Linear(F N -> 1) read-out, the recurrence as scaled x3 steps (gcrnn_fused_forward_x3_scaled) -- against the fp64 oracle
    (reference graphML.py:2357-2374, 2420-2423), the drivers' G = 1, a batch whose gate items need several slices (B T = 2800), and the
    composed path it replaces."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(37)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
    Bo = min(B, 3)                                               # the oracle checks the first sequences (dense fp64 hops)
    X = rng.standard_normal((B, T, G, N)).astype(np.float32)
    h0 = np.zeros((B, F, N), np.float32) if zero_h0 else (0.5 * rng.standard_normal((B, F, N))).astype(np.float32)
    torch.manual_seed(8)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, True, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.float()
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    ref = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X[:Bo].astype(np.float64), h0[:Bo].astype(np.float64), True, None)
    cell = cell.to(dev)
    Xd, hd = torch.tensor(X, device=dev), torch.tensor(h0, device=dev)
    with torch.no_grad():
        assert cell._use_fused_x3(Xd, hd, time_gated=True) and not cell._use_fused_x3(Xd, hd)
        H = cell(Xd, hd)
        Hl = cell(Xd, hd, last_only=True)
        monkeypatch.setenv('GCRNN_NO_X3_GATED', '1')
        assert not cell._use_fused_x3(Xd, hd, time_gated=True)
        Hc = cell(Xd[:Bo], hd[:Bo])                              # the composed path
    assert H.dtype == torch.float32 and tuple(H.shape) == (B, T, F, N)
    err = np.abs(H[:Bo].double().cpu().numpy() - ref).max()
    assert err <= 1e-5, err
    assert torch.equal(Hl, H[:, -1:])
    assert float((H[:Bo] - Hc).abs().max()) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T', [(1000, 64, 64, 5, 3, 6), (1000, 64, 1, 5, 2, 4), (600, 32, 32, 3, 5, 3), (1008, 64, 32, 2, 2, 3),
                                         (1000, 64, 64, 4, 2, 2)])
def test_fp32_accurate_fused_path_matches_oracle_to_1e5(N, F, G, K, B, T):
    """gcrnn_fused_forward_x3: the un-gated cell on the fused kernels with every fp32 operand carried as three bf16 planes and
    six partial products per tap product -- fp32 inputs, fp32 output, <= 1e-5 against the fp64 oracle (the north_star's
    tolerance), non-zero h0, the drivers' G = 1 (padded channels), and the last-state-only read-out."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(31)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
    X = rng.standard_normal((B, T, G, N)).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((B, F, N))).astype(np.float32)
    torch.manual_seed(4)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.float()
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    ref = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X.astype(np.float64), h0.astype(np.float64))
    cell = cell.to(dev)
    Xd, hd = torch.tensor(X, device=dev), torch.tensor(h0, device=dev)
    with torch.no_grad():
        assert cell._use_fused_x3(Xd, hd)
        H = cell(Xd, hd)
        Hl = cell(Xd, hd, last_only=True)
        H2 = cell(Xd, hd)
    assert H.dtype == torch.float32 and tuple(H.shape) == (B, T, F, N)
    err = np.abs(H.double().cpu().numpy() - ref).max()
    assert err <= 1e-5, err
    assert torch.equal(Hl, H[:, -1:]) and torch.equal(H, H2)               # same bits: last-only read-out, second run


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,tg', [(200, 32, 32, 3, False), (1000, 64, 64, 5, False), (1000, 64, 1, 5, True), (600, 64, 64, 2, False),
                                        (304, 32, 32, 5, True)])
def test_fused_node_gated_forward_matches_oracle(N, F, G, K, tg):
    """Node-gated (and time + node gated) cell on the fused kernels vs the fp64 oracle (graphML.py:2379-2407): gate cells on
    (x_t, h0), their F -> 1 graph filters taps-first on one-channel hops, A(S)x_t + b for all steps at once, per-node gates in the
    recurrent step's epilogue. Non-zero h0, directed weighted graph."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    S = random_graph(N, min(0.5, 10.0 / N), 47)
    rng = np.random.default_rng(12)
    B, T = 4, 3
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    torch.manual_seed(6)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'node', 1, True)
    cell.addGSO(torch.tensor(S))
    params = {k: bf16_round(v.detach().numpy()) for k, v in cell.state_dict().items()}
    ref = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X, h0, tg, 'node')
    cell = cell.to(dev).to(torch.bfloat16)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        assert cell._use_fused_node(Xd, hd)
        H = cell(Xd, hd)
    err = np.abs(H.double().cpu().numpy() - ref)
    # bf16 states, the x part A(S)x_t + b stored in bf16 between its pass and the recurrence: same tolerance class as the fused cell
    assert err.max() <= 1.2e-2 and err.mean() <= 1.5e-3, (err.max(), err.mean())


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T,tg', [(1000, 64, 64, 5, 4, 4, False), (200, 32, 32, 3, 6, 3, True), (600, 64, 1, 3, 3, 3, False),
                                            (1000, 64, 64, 2, 2, 2, True)])
def test_fused_node_gated_training_matches_composed_autograd(N, F, G, K, B, T, tg):
    """Node-gated (and time + node gated) cell, bf16 activations over fp32 master weights: forward, both node-gate sub-networks
    (gate cell + F -> 1 graph filter), the time gates and the whole BPTT on the fused kernels reproduce the fp32 autograd gradients
    of the composed path for EVERY trained parameter (reference graphML.py:2379-2407, 2420-2423). h0 != 0."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    S = random_graph(N, min(0.5, 10.0 / N), 53)
    rng = np.random.default_rng(14)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    dH = bf16_round(rng.standard_normal((B, T, F, N)))
    torch.manual_seed(19)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'node', 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).to(torch.float32)
    ref = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'node', 1, True)
    ref.addGSO(torch.tensor(S))
    ref.load_state_dict(cell.state_dict())
    cell, ref = cell.to(dev), ref.to(dev)
    Hr = ref(torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev))
    dHd = torch.tensor(dH, dtype=torch.float32, device=dev)
    (Hr * dHd).sum().backward()
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    assert cell._use_fused_training(Xd, hd)
    H = cell(Xd, hd)
    assert H.dtype == torch.bfloat16 and H.requires_grad
    err = (H.detach().float() - Hr.detach()).abs()
    assert float(err.max()) <= 1.2e-2 and float(err.mean()) <= 1.5e-3, (float(err.max()), float(err.mean()))
    (H.float() * dHd).sum().backward()
    got = dict(cell.named_parameters())
    checked = 0
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert got[n].grad is None or float(got[n].grad.abs().max()) == 0.0, n
            continue
        assert got[n].grad is not None, n
        g, gr = got[n].grad.float().cpu().numpy(), p.grad.cpu().numpy()
        sc = np.abs(gr).max()
        if gr.size == 1 and n.endswith('.bias'):
            # a scalar bias gradient is a signed sum over every (t, b, n) and may cancel to far below its terms: measure it on the
            # scale of its sibling weight's gradient (the same terms times values of magnitude <= 1)
            sib = dict(ref.named_parameters()).get(n[:-5] + '.weight')
            if sib is not None and sib.grad is not None:
                sc = max(sc, float(sib.grad.abs().max()))
        e = np.abs(g - gr)
        assert sc > 0 and e.max() <= 5e-2 * sc and (e.size < 16 or e.mean() <= 1e-2 * sc), (n, e.max() / sc, e.mean() / sc)
        checked += 1
    assert checked >= 11 + (6 if tg else 0)          # cell 3 + two gate cells 6 + two F -> 1 filters 4 (+ time gates)


@pytest.mark.gpu
def test_full_size_bf16_states_track_the_fp32_accurate_path():
    """BASELINE configs[1] at the bench's own size (N=1000, K=5, T=32, G=F=64, B=64 of the 256): the bf16 fused kernel against the
    fp32-accurate fused kernel on the same fp32 operands rounded to bf16 -- the latter is pinned to the fp64 oracle and to golden G8
    at <= 1e-5, so this bounds the bf16 path's drift over all 32 steps at full size, where the oracle itself does not finish in
    seconds. bf16 rounds every state to 8 significant bits; the recurrence is contractive enough that the error does not grow."""
    import gated_gcrnns_amd.Utils.graphML as gml
    import sys, os
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from bench import sbm_graph
    dev = torch.device('cuda:0')
    N, K, T, F, B = 1000, 5, 32, 64, 64
    S = sbm_graph(N)
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).float().to(dev)                 # bf16-representable parameters
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    Xb = torch.randn(B, T, F, N, device=dev, generator=gen).to(torch.bfloat16)
    h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        assert cell._use_fused_x3(Xb.float(), h0.float())
        H32 = cell(Xb.float(), h0.float())
        cb = cell.to(torch.bfloat16)
        assert cb._use_fused(Xb, h0)
        Hb = cb(Xb, h0).float()
    err = (Hb - H32).abs()
    per_step = err.amax(dim=(0, 2, 3))
    assert float(err.max()) <= 5e-3 and float(err.mean()) <= 1e-3, (float(err.max()), float(err.mean()))
    assert float(per_step[-1]) <= 3e-2 and float(per_step[-8:].mean()) <= 1.5 * float(per_step[:8].mean()) + 5e-3, per_step.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,tg,sym', [(200, 32, 32, 3, False, False), (1000, 64, 64, 5, False, True), (1000, 64, 1, 5, True, False),
                                            (600, 64, 64, 2, False, False), (304, 32, 32, 5, True, True)])
def test_fused_edge_gated_forward_matches_oracle(N, F, G, K, tg, sym):
    """Edge-gated (and time + edge gated) cell on the fused kernels vs the fp64 oracle (graphML.py:2409-2416; graphAttention
    521-627): the mixing matrix folded into the filter taps, softmax over the support rows of S + I, aggregation over the
    columns, ReLU, then the cell's tanh. Non-zero h0; directed weighted graphs and symmetric ones."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    S = random_graph(N, min(0.5, 10.0 / N), 61)
    if sym:
        S = 0.5 * (S + S.transpose(0, 2, 1))
    rng = np.random.default_rng(21)
    B, T = 4, 3
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    torch.manual_seed(8)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'edge', 1, True)
    cell.addGSO(torch.tensor(S))
    params = {k: bf16_round(v.detach().numpy()) for k, v in cell.state_dict().items()}
    ref = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X, h0, tg, 'edge')
    cell = cell.to(dev).to(torch.bfloat16)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        assert cell._use_fused_edge(Xd, hd)
        H = cell(Xd, hd)
        Hl = cell(Xd, hd, last_only=True)
    err = np.abs(H.double().cpu().numpy() - ref)
    # bf16 states; the filter outputs z (composite taps rounded to bf16) and the x branch are stored in bf16 between the passes
    assert err.max() <= 1.2e-2 and err.mean() <= 1.5e-3, (err.max(), err.mean())
    assert torch.equal(Hl[:, 0], H[:, -1])


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T,tg', [(1000, 64, 64, 5, 4, 4, False), (200, 32, 32, 3, 6, 3, True), (600, 64, 1, 3, 3, 3, False),
                                            (1000, 64, 64, 2, 2, 2, True)])
def test_fused_edge_gated_training_matches_composed_autograd(N, F, G, K, B, T, tg):
    """Edge-gated (and time + edge gated) cell, bf16 activations over fp32 master weights: forward, both attention layers (mixer and
    mixing matrix), the time gates and the whole BPTT on the fused kernels reproduce the fp32 autograd gradients of the composed
    path for EVERY trained parameter (reference graphML.py:2409-2416 with graphAttention 521-627 under autograd). h0 != 0."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    S = random_graph(N, min(0.5, 10.0 / N), 59)
    rng = np.random.default_rng(16)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    dH = bf16_round(rng.standard_normal((B, T, F, N)))
    torch.manual_seed(23)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'edge', 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).to(torch.float32)
    ref = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'edge', 1, True)
    ref.addGSO(torch.tensor(S))
    ref.load_state_dict(cell.state_dict())
    cell, ref = cell.to(dev), ref.to(dev)
    Hr = ref(torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev))
    dHd = torch.tensor(dH, dtype=torch.float32, device=dev)
    (Hr * dHd).sum().backward()
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    assert cell._use_fused_training(Xd, hd)
    H = cell(Xd, hd)
    assert H.dtype == torch.bfloat16 and H.requires_grad
    err = (H.detach().float() - Hr.detach()).abs()
    assert float(err.max()) <= 1.2e-2 and float(err.mean()) <= 1.5e-3, (float(err.max()), float(err.mean()))
    (H.float() * dHd).sum().backward()
    got = dict(cell.named_parameters())
    refg = {n: p.grad.cpu().numpy() for n, p in ref.named_parameters() if p.grad is not None}
    checked = 0
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert got[n].grad is None or float(got[n].grad.abs().max()) == 0.0, n
            continue
        assert got[n].grad is not None, n
        g, gr = got[n].grad.float().cpu().numpy(), p.grad.cpu().numpy()
        sc = np.abs(gr).max()
        if gr.size == 1 and n.endswith('.bias'):
            sib = dict(ref.named_parameters()).get(n[:-5] + '.weight')
            if sib is not None and sib.grad is not None:
                sc = max(sc, float(sib.grad.abs().max()))
        e = np.abs(g - gr)
        if not (gr.size == 1 and n.endswith('.bias')):
            sc, tmax, tmean = _grad_scale_and_bounds(n, refg, 5e-2, 1e-2)
        else:
            tmax, tmean = 5e-2, 1e-2
        assert sc > 0 and e.max() <= tmax * sc and (e.size < 16 or e.mean() <= tmean * sc), (n, e.max() / sc, e.mean() / sc)
        checked += 1
    assert checked >= 7


@pytest.mark.gpu
def test_fused_edge_gated_gradients_are_deterministic():
    """Two training steps of the edge-gated cell on the same inputs give bit-identical outputs and gradients (every sum of the
    attention kernels is a gather in a fixed order)."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, F, G, K, B, T = 400, 64, 64, 3, 5, 3
    S = random_graph(N, 10.0 / N, 71)
    rng = np.random.default_rng(18)
    Xd = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.bfloat16, device=dev)
    torch.manual_seed(29)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, True, 'edge', 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev)
    outs = []
    for _ in range(2):
        cell.zero_grad(set_to_none=True)
        assert cell._use_fused_training(Xd, hd)
        H = cell(Xd, hd)
        H.float().square().sum().backward()
        outs.append((H.detach().clone(), {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys() and len(outs[0][1]) >= 9
    for n in outs[0][1]:
        assert torch.equal(outs[0][1][n], outs[1][1][n]), n


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,items,gated', [(200, 64, 3, False), (200, 32, 4, True), (1000, 64, 2, True)])
def test_fused_edge_attention_kernels_match_dense_autograd(N, F, items, gated):
    """The attention kernels alone (forward and backward, C ABI through ops) against a dense fp64 restatement of graphAttention
    (graphML.py:585-627) under torch autograd on the SAME bf16-representable inputs: only the bf16 rounding of the kernels' outputs
    separates the two, so the bounds are tight."""
    from gated_gcrnns_amd import ops
    from gated_gcrnns_amd.graph import as_operator
    dev = torch.device('cuda:0')
    S = random_graph(N, min(0.5, 10.0 / N), 83)
    graph = as_operator(torch.tensor(S)).to(dev)
    npad = graph.fused_plan()['npad']
    rng = np.random.default_rng(31)
    z = bf16_round(rng.standard_normal((items, N, F)))
    dpre = bf16_round(rng.standard_normal((items, N, F)))
    a12 = (0.3 * rng.standard_normal((2, F))).astype(np.float32)
    g = rng.uniform(0.3, 1.0, items).astype(np.float32) if gated else None
    slope = 0.2
    # dense reference, fp64 autograd
    zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    at = torch.tensor(a12.astype(np.float64), requires_grad=True)
    Sp = torch.tensor(S[0].astype(np.float32).astype(np.float64)) + torch.eye(N, dtype=torch.float64)
    mask = (Sp.abs() > 1e-9)
    s1 = zt @ at[0]                                              # [items][N]: scores the receiving node n
    s2 = zt @ at[1]                                              # scores the row m
    e = torch.nn.functional.leaky_relu(s1[:, None, :] + s2[:, :, None], slope)      # [items][m][n]
    e = e.masked_fill(~mask, float('-inf'))
    alpha = torch.softmax(e, dim=2)
    alpha = torch.where(mask, alpha, torch.zeros_like(alpha))
    o = torch.einsum('imn,imf->inf', alpha * Sp, zt)
    r_ref = torch.relu(o)
    gt = torch.tensor(g.astype(np.float64)) if gated else torch.ones(items, dtype=torch.float64)
    (r_ref * torch.tensor(dpre) * gt[:, None, None]).sum().backward()
    # kernels
    zd = torch.zeros((items, npad, F), dtype=torch.bfloat16, device=dev)
    zd[:, :N] = torch.tensor(z, dtype=torch.bfloat16, device=dev)
    dd = torch.zeros_like(zd)
    dd[:, :N] = torch.tensor(dpre, dtype=torch.bfloat16, device=dev)
    a12d = torch.tensor(a12, device=dev)
    r = ops.fused_edge_attention(zd, a12d, graph, N=N, negative_slope=slope)
    err = (r[:, :N].double().cpu() - r_ref.detach()).abs()
    assert float(err.max()) <= 1.0 / 128 * float(r_ref.abs().max()) and float(r[:, N:].abs().max()) == 0.0
    gd = torch.tensor(g, device=dev) if gated else None
    dz, da_part, dgate = ops.fused_edge_attention_backward(dd, r, gd, zd, a12d, graph, N, negative_slope=slope)
    dz_ref = zt.grad
    e1 = (dz[:, :N].double().cpu() - dz_ref).abs()
    assert float(e1.max()) <= 1.5e-2 * float(dz_ref.abs().max()) and float(e1.mean()) <= 2e-3 * float(dz_ref.abs().max()), (float(e1.max()), float(e1.mean()))
    assert float(dz[:, N:].abs().max()) == 0.0
    da = da_part.double().sum(dim=0).cpu()
    e2 = (da - at.grad).abs()
    assert float(e2.max()) <= 5e-3 * float(at.grad.abs().max()), (float(e2.max()), float(at.grad.abs().max()))
    if gated:
        dg_ref = (r[:, :N].double().cpu() * torch.tensor(dpre)).sum(dim=(1, 2))
        assert float((dgate.double().cpu() - dg_ref).abs().max()) <= 1e-4 * float(dg_ref.abs().max()) + 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T', [(1000, 64, 64, 5, 5, 4), (1000, 64, 1, 3, 3, 3), (400, 32, 32, 3, 7, 3), (1000, 64, 64, 2, 2, 2),
                                         (1000, 64, 32, 4, 3, 3)])
def test_inline_pack_of_next_input_is_bit_identical(N, F, G, K, B, T, monkeypatch):
    """Uniform-weight graphs: every launch of the un-gated recurrence lays out x_{t+1} itself (LDS-DMA of the user-layout rows,
    transposed read-back after the epilogue) instead of a pack pass over X. Same bits as the packed path for the states, the
    user-layout output and the sequence-major inputs it wrote (rows >= N zero); training forward included."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(41)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
    torch.manual_seed(11)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev).to(torch.bfloat16)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    Xp, wA = ops.fused_pad_operands(X, cell.weight_A.detach())
    plan = cell.graph.fused_plan()
    assert plan['uniform_w'] != 0.0 and ops.fused_inline_pack_ok(plan, N, F, Xp.shape[2], K)
    with torch.no_grad():
        hs1, _, H1 = ops.fused_cell_forward(Xp, h0, wA, cell.weight_B, cell.bias, cell.graph, return_states=True)
        Hl1 = cell(X, h0, last_only=True)
        monkeypatch.setenv('GCRNN_NO_INLINE_PACK', '1')
        assert not ops.fused_inline_pack_ok(plan, N, F, Xp.shape[2], K)
        hs0, _, H0 = ops.fused_cell_forward(Xp, h0, wA, cell.weight_B, cell.bias, cell.graph, return_states=True)
        Hl0 = cell(X, h0, last_only=True)
    assert torch.equal(H0, H1) and torch.equal(hs0, hs1) and torch.equal(Hl0, Hl1)
    assert float(hs1[:, :, N:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('tg', [False, True])
def test_inline_pack_in_the_bptt_chain_is_bit_identical(tg, monkeypatch):
    """Training step on a uniform-weight graph: the forward launches lay out x_{t+1}, the BPTT launches dH_{t-2} (only the first /
    last two steps go through the pack kernel). Same bits for every gradient as with the pack passes."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, F, G, K, B, T = 1000, 64, 64, 5, 6, 5
    rng = np.random.default_rng(43)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
    torch.manual_seed(13)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)

    def step():
        cell.zero_grad(set_to_none=True)
        assert cell._use_fused_training(X, h0)
        H = cell(X, h0)
        torch.nn.functional.l1_loss(H.float(), tgt).backward()
        return H.detach().clone(), {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}

    H1, g1 = step()
    monkeypatch.setenv('GCRNN_NO_INLINE_PACK', '1')
    H0, g0 = step()
    assert torch.equal(H0, H1) and g0.keys() == g1.keys() and len(g1) >= 3
    for n in g1:
        assert torch.equal(g0[n], g1[n]), n


@pytest.mark.gpu
def test_inline_pack_full_size_repeated_forwards_match_the_packed_path():
    """30 forwards at the bench's full size (B = 256, T = 32: 250k inline-packed (item, step) tiles) against the packed path, bit for
    bit. Guards the store-data hazard found on gfx950 (a 16-byte buffer store with an SGPR soffset followed at once by a
    VALU write of its first data register), which corrupted about one tile in 3 million before the read-back was restructured."""
    import bench
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, K, T, F, B = 1000, 5, 32, 64, 256
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(bench.sbm_graph(N)))
    cell = cell.to(torch.bfloat16).to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    X = torch.randn(B, T, F, N, device=dev, generator=gen).to(torch.bfloat16)
    h0 = (0.3 * torch.randn(B, F, N, device=dev, generator=gen)).to(torch.bfloat16)
    with torch.no_grad():
        os.environ['GCRNN_NO_INLINE_PACK'] = '1'
        try:
            Href = cell(X, h0).clone()
        finally:
            del os.environ['GCRNN_NO_INLINE_PACK']
        for r in range(30):
            assert torch.equal(cell(X, h0), Href), r


@pytest.mark.gpu
def test_edge_gated_cell_with_hub_nodes_on_the_fused_kernels():
    """The attention kernels keep a node's edge records in registers (32 per node); hubs -- a row reaching 60 nodes, a column reached
    by 70 -- continue chunk by chunk (forward: slow in-degree loop; backward: d alpha of the extra records parked in the scratch).
    Fused inference and the whole fused BPTT against the composed fp32 path."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, F, G, K, B, T = 200, 32, 32, 3, 3, 3
    rng = np.random.default_rng(3)
    S = (rng.random((N, N)) < 0.04) * rng.uniform(0.2, 1.0, (N, N))
    S[7, :60] = rng.uniform(0.2, 1.0, 60)
    S[:70, 11] = rng.uniform(0.2, 1.0, 70)
    S = (S / np.max(np.abs(np.linalg.eigvals(S)))).reshape(1, N, N)
    torch.manual_seed(2)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, 'edge', 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).float()
    ref = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, 'edge', 1, True)
    ref.addGSO(torch.tensor(S))
    ref.load_state_dict(cell.state_dict())
    cell, ref = cell.to(dev), ref.to(dev)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    assert cell.graph.edge_plan()['max_out_degree'] > 32
    Hr = ref(X.float(), h0.float())                    # composed path
    Hr.square().mean().backward()
    with torch.no_grad():
        assert cell._use_fused_edge(X, h0)
        Hf = cell(X, h0)
    err = (Hf.float() - Hr.detach()).abs()
    assert float(err.max()) <= 4e-2 and float(err.mean()) <= 3e-3, (float(err.max()), float(err.mean()))
    assert cell._use_fused_training(X, h0)
    H = cell(X, h0)
    H.float().square().mean().backward()
    refg = {k: v.grad.cpu().numpy() for k, v in ref.named_parameters()}
    for n, p in cell.named_parameters():
        g, gr = p.grad.float().cpu().numpy(), refg[n]
        sc, tmax, tmean = _grad_scale_and_bounds(n, refg, 5e-2, 1e-2)
        e = np.abs(g - gr)
        assert sc > 0 and e.max() <= tmax * sc and (e.size < 16 or e.mean() <= tmean * sc), (n, e.max() / sc, e.mean() / sc)


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,tg,uniform', [(1000, 64, 64, 5, False, True), (1000, 64, 1, 3, True, True), (400, 32, 32, 3, False, False),
                                                (600, 64, 64, 2, True, False)])
def test_regression_head_fused_onto_the_state_store(N, F, G, K, tg, uniform):
    """the regression model's `multipMlp` head with one output (reference architectures.py:1616-1627: the same
    Linear(F -> 1) on every node) evaluated in the step kernel's epilogue, H never written in the user layout. Same numbers as
    cell -> head kernel on the bf16-rounded states (fp32 sums in another order), un-gated and time-gated, uniform and weighted graphs."""
    import gated_gcrnns_amd.Modules.architectures as archit
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(51)
    if uniform:
        W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
        W = np.triu(W, 1); W = W + W.T
        S = W / np.max(np.abs(np.linalg.eigvalsh(W)))
    else:
        S = random_graph(N, 10.0 / N, 91)[0]
    torch.manual_seed(17)
    m = archit.GatedGCRNNforRegression(G, F, K, K, torch.tanh, torch.nn.ReLU, [1], S, True, time_gating=tg, spatial_gating=None,
                                       mlpType='multipMlp').to(dev).float()
    B, T = 5, 4
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        y = m(X, h0)                                                 # fused head
        assert m.stateGCRNN.forward_with_head(X, h0, m.outputNN[0].weight, m.outputNN[0].bias) is not None
        H = m.stateGCRNN(X, h0)                                      # the cell alone, then the head by hand in fp64
        lin = m.outputNN[0]
        want = torch.einsum('of,btfn->bton', lin.weight.double(), H.double()) + lin.bias.double().view(1, 1, -1, 1)
    assert tuple(y.shape) == (B, T, 1, N)
    err = (y.double() - want).abs().max()
    assert float(err) <= 1.0 / 128 * float(want.abs().max()), float(err)            # y is returned in the input dtype (bf16)
    with torch.no_grad():
        yf = m.stateGCRNN.forward_with_head(X, h0, lin.weight, lin.bias)             # fp32 result of the fused epilogue
    assert float((yf.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('sg', ['edge', None])
def test_fused_paths_without_bias(sg):
    """Cells built with bias=False (the reference allows it, graphML.py:2218-2222) on the fused kernels: forward and training against
    the composed fp32 path."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, F, G, K, B, T = 400, 64, 64, 3, 4, 3
    S = random_graph(N, 10.0 / N, 97)
    rng = np.random.default_rng(23)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    torch.manual_seed(31)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, sg, 1, False)
    cell.addGSO(torch.tensor(S))
    assert cell.bias is None
    cell = cell.to(torch.bfloat16).to(torch.float32)
    ref = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, sg, 1, False)
    ref.addGSO(torch.tensor(S))
    ref.load_state_dict(cell.state_dict())
    cell, ref = cell.to(dev), ref.to(dev)
    Hr = ref(torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev))
    Hr.square().mean().backward()
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        assert cell._use_fused_edge(Xd, hd) if sg == 'edge' else cell._use_fused(Xd, hd)
        Hi = cell(Xd, hd)
    assert float((Hi.float() - Hr.detach()).abs().max()) <= 4e-2
    assert cell._use_fused_training(Xd, hd)
    H = cell(Xd, hd)
    H.float().square().mean().backward()
    for n, p in ref.named_parameters():
        g, gr = dict(cell.named_parameters())[n].grad.float().cpu().numpy(), p.grad.cpu().numpy()
        sc, tmax, tmean = _grad_scale_and_bounds(n, {k: v.grad.cpu().numpy() for k, v in ref.named_parameters()}, 5e-2, 1e-2)
        e = np.abs(g - gr)
        assert sc > 0 and e.max() <= tmax * sc and (e.size < 16 or e.mean() <= tmean * sc), (n, e.max() / sc, e.mean() / sc)


@pytest.mark.parametrize('N,density,iso,uniform', [(1000, 0.01, 0, True), (1000, 0.01, 0, False), (200, 0.05, 7, True), (1024, 0.004, 30, False),
                                                   (1001, 0.01, 0, True)])
def test_plan_bank_keys_make_write_backs_and_scatters_conflict_free(N, density, iso, uniform):
    """CPU: host side of the LDS placement (write-aware keys). (1) every node sits in exactly one slot and every LDS row is
    used once; (2) the 8 slots of every half tile carry 8 different write keys (row & 1) << 2 | swz -- the kernels' ds_write_b128
    write-backs of the state (8-lane groups, 32 banks) are then conflict-free; (3) a uniform-weight plan has a zero row for each of the 16
    gather keys; (4) the modelled gather cycles stay within 8 % of conflict-free; (5) spreading the tiles over the (node >> 1) & 15
    classes neither changes the ELL size nor a tile's depth bound; (6) the plan is a pure function of the graph."""
    S = random_graph(N, density, 11, iso)
    if uniform:
        S = (S != 0).astype(np.float64) * 0.03125
    g = GraphOperator(S)
    plan = g.fused_plan()
    npad = plan['npad']
    slots = plan['tile_slots'].numpy().astype(np.int64)
    nodes, addr = slots >> 16, slots & 0xffff
    assert sorted(nodes.tolist()) == list(range(npad))
    rows, swz = addr >> 6, (addr >> 4) & 3
    assert sorted(rows.tolist()) == list(range(npad)) and np.all(addr & 15 == 0)
    wkey = ((rows & 1) << 2) | swz
    assert all(len(set(h.tolist())) == 8 for h in wkey.reshape(-1, 8))
    if uniform and npad - N >= 16:
        assert plan['uniform_w'] != 0.0
        na = plan['node_addr'].numpy()[N:N + 16].astype(np.int64)
        key4 = (((na >> 4) & 3) << 2) | ((na >> 6) & 3)
        assert sorted(key4.tolist()) == list(range(16))
    else:
        assert plan['uniform_w'] == 0.0 or npad - N >= 16
    assert plan['gather_cycles'] <= 1.08 * 4 * plan['entries']
    # tiles: depth bound kept, ELL size equal to the plain degree-ranked composition
    import os
    os.environ['GCRNN_PLAN_NO_CLASS_SPREAD'] = '1'
    try:
        plain = GraphOperator(S).fused_plan()
    finally:
        del os.environ['GCRNN_PLAN_NO_CLASS_SPREAD']
    assert plain['entries'] == plan['entries']
    assert np.array_equal(plain['tile_off'].numpy(), plan['tile_off'].numpy())
    cls = (nodes.reshape(-1, 16) >> 1) & 15
    dup_new = sum(16 - len(set(r.tolist())) for r in cls)
    dup_old = sum(16 - len(set(r.tolist())) for r in ((plain['tile_slots'].numpy().astype(np.int64) >> 16).reshape(-1, 16) >> 1) & 15)
    assert dup_new <= dup_old
    again = GraphOperator(S).fused_plan()
    assert np.array_equal(again['tile_slots'].numpy(), plan['tile_slots'].numpy())
    assert np.array_equal(again['ell_col'].numpy(), plan['ell_col'].numpy())
    # (7) the same plan re-addressed for the bf16 hop image (32-byte rows): unique rows, gather keys a bijection of the fp32 image's
    # (the conflict-free entry schedule carries over; only the few rows that did not fit their class may differ), every
    # (row & 3, half swizzle) exactly twice per tile = the 2-way minimum of a 16-lane ds_write_b64 group, column words = neighbour
    # addresses in the order (entry 0, 2, 1, 3)
    p16 = g.fused_plan_img16()
    if plan['uniform_w'] == 0.0:
        assert p16 is None
        return
    a16 = p16['node_addr16'].numpy().astype(np.int64)
    assert sorted((a16 >> 5).tolist()) == list(range(npad)) and np.all(a16 & 15 == 0) and a16.max() < 32768
    na = plan['node_addr'].numpy().astype(np.int64)
    old_key, new_key = (na >> 4) & 15, (a16 >> 4) & 15
    same = sum(1 for k in range(16) if len(set(new_key[old_key == k].tolist())) <= 1)
    assert same >= 16 - 2 * max(1, p16['img16_moved']) and p16['img16_moved'] <= npad // 32
    s16 = p16['tile_slots'].numpy().astype(np.int64)
    assert np.array_equal(s16 >> 16, nodes) and np.array_equal(s16 & 0xffff, a16[nodes])
    wk16 = ((((s16 & 0xffff) >> 5) & 3) << 1) | (((s16 & 0xffff) >> 4) & 1)
    assert sum(max(np.bincount(t, minlength=8)) > 2 for t in wk16.reshape(-1, 16)) <= max(2, 2 * p16['img16_moved'])
    ec = plan['ell_col'].numpy().astype(np.int64).reshape(-1, 4, 16)
    c16 = p16['ell_col4'].numpy().view(np.uint16).astype(np.int64).reshape(-1, 16, 4)
    for w_, e_ in enumerate((0, 2, 1, 3)):
        assert np.array_equal(a16[ec[:, e_, :]], c16[:, :, w_])


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,T', [(1000, 64, 5, 32), (600, 32, 3, 8)])
def test_bf16_hop_image_with_matrix_core_sums_against_oracle_and_fp32_image(N, F, K, T, monkeypatch):
    """The un-gated step kernels on uniform-weight graphs keep the hop state as a bf16 image and sum the gathered neighbour rows on
    the matrix cores. Same fp64 oracle, same tolerances as the fp32 image over a full-length sequence; the extra
    rounding of the hop intermediates must not cost more than a third on top of the fp32 image's own error; gradients of a training
    step (forward AND the BPTT data chain on the bf16 image) against the fp32-image kernels."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(33)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
    B = 3
    X = bf16_round(rng.standard_normal((B, T, F, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    torch.manual_seed(5)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    params = {k: bf16_round(v.detach().numpy()) for k, v in cell.state_dict().items()}
    ref = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X, h0)
    cell = cell.to(dev).to(torch.bfloat16)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    assert cell.graph.fused_plan_img16() is not None and ops.fused_img16_plan(cell.graph, False, None) is not None
    with torch.no_grad():
        H16 = cell(Xd, hd).double().cpu().numpy()
    monkeypatch.setenv('GCRNN_NO_IMG16', '1')
    assert ops.fused_img16_plan(cell.graph, False, None) is None
    with torch.no_grad():
        H32 = cell(Xd, hd).double().cpu().numpy()
    monkeypatch.delenv('GCRNN_NO_IMG16')
    e16, e32 = np.abs(H16 - ref), np.abs(H32 - ref)
    print('bf16 image: max %.2e mean %.2e   fp32 image: max %.2e mean %.2e' % (e16.max(), e16.mean(), e32.max(), e32.mean()))
    assert e16[:, 0].max() <= 3.0e-3 and e16.max() <= 5e-3 and e16.mean() <= 1e-3, (e16.max(), e16.mean())
    assert e16.mean() <= 1.34 * e32.mean() + 1e-5, (e16.mean(), e32.mean())
    # training step: gradients with the bf16 image (forward + BPTT data chain) vs the fp32 image
    cellf = cell.float()
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)
    def grads():
        for p_ in cellf.parameters():
            p_.grad = None
        Hh = cellf(Xd, hd)
        ((Hh.float() - tgt) ** 2).mean().backward()
        return {n: p_.grad.detach().double().cpu().numpy().copy() for n, p_ in cellf.named_parameters() if p_.grad is not None}
    g16 = grads()
    monkeypatch.setenv('GCRNN_NO_IMG16', '1')
    g32 = grads()
    assert set(g16) == set(g32) and 'weight_A' in g16
    for n in g16:
        sc = np.abs(g32[n]).max()
        assert sc > 0 and np.abs(g16[n] - g32[n]).max() <= 3e-2 * sc, (n, np.abs(g16[n] - g32[n]).max() / sc)


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,tg,hub', [(200, 32, 5, False, 90), (1008, 64, 2, True, 0), (496, 64, 4, False, 150), (1000, 32, 3, True, 60),
                                          (136, 64, 5, False, 0)])
def test_bf16_hop_image_on_hub_graphs_small_graphs_and_sixteen_padding_rows(N, F, K, tg, hub, monkeypatch):
    """The matrix-core hop stream at the edges of its domain: a hub row / column of degree `hub` (deep first tile, every other tile
    shallow), few nodes (most of the image is padding rows), exactly 16 padding rows (N = 1008: the zero rows of the 16 gather keys
    are all the padding there is), time gates (gate pre-pass + gated steps). Against the fp64 oracle and the fp32-image kernels."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(1000 + N + hub)
    W = (rng.random((N, N)) < 8.0 / N).astype(np.float64)
    W = np.triu(W, 1)
    if hub:
        W[0, rng.choice(np.arange(1, N), size=min(hub, N - 1), replace=False)] = 1.0
    W = W + W.T
    S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
    B, T = 3, 4
    X = bf16_round(rng.standard_normal((B, T, F, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    torch.manual_seed(9)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    params = {k: bf16_round(v.detach().numpy()) for k, v in cell.state_dict().items()}
    ref = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X, h0, tg, None)
    cell = cell.to(dev).to(torch.bfloat16)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    p16 = cell.graph.fused_plan_img16()
    assert p16 is not None and ops.fused_img16_plan(cell.graph, tg, None) is p16
    with torch.no_grad():
        assert cell._use_fused(Xd, hd)
        H16 = cell(Xd, hd).double().cpu().numpy()
    monkeypatch.setenv('GCRNN_NO_IMG16', '1')
    with torch.no_grad():
        H32 = cell(Xd, hd).double().cpu().numpy()
    e16, e32 = np.abs(H16 - ref), np.abs(H32 - ref)
    assert e16.max() <= 5e-3 and e16.mean() <= 1e-3, (e16.max(), e16.mean())
    assert e16.mean() <= 1.5 * e32.mean() + 1e-5, (e16.mean(), e32.mean())
    assert np.abs(H16 - H32).max() <= 1.6e-2


def _uniform_cell(N, G, F, K, tg, seed, dev, dtype=torch.bfloat16):
    import gated_gcrnns_amd.Utils.graphML as gml
    rng = np.random.default_rng(seed)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    S = (W / np.max(np.abs(np.linalg.eigvalsh(W)))).reshape(1, N, N)
    torch.manual_seed(seed)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev)
    return (cell.to(dtype) if dtype is not None else cell), rng, S


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T', [(1000, 64, 64, 5, 5, 4), (1000, 64, 1, 3, 3, 3), (400, 32, 32, 3, 7, 3), (1000, 64, 64, 2, 2, 2),
                                         (1000, 64, 32, 4, 3, 3), (1008, 64, 64, 5, 260, 2)])
def test_sequence_resident_kernel_is_bit_identical_to_the_chunk_parallel_kernel(N, F, G, K, B, T, monkeypatch):
    """gcrnn_fused_seq.h: one workgroup per sequence keeps the operand [h | x] in registers and walks the F/16 chunks, evaluating
    each tap right before the hop that adds it. Same arithmetic in the same order as the chunk-parallel step kernel: same bits for
    the states, the user-layout output, the inline-packed inputs and the last-state-only path (with and without the inline pack;
    B = 260 > 256 workgroups exercises the sequence loop and the LDS-DMA'd weights of the next sequence's first chunk)."""
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    cell, rng, _ = _uniform_cell(N, G, F, K, False, 71, dev)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    Xp, wA = ops.fused_pad_operands(X, cell.weight_A.detach())
    monkeypatch.setenv('GCRNN_SEQ_MIN_B', '1'); monkeypatch.setenv('GCRNN_SEQ32', '0')      # (the 16-feature kernels: the wide one is pinned in tests/test_wide.py)
    for nopack in ('0', '1'):
        monkeypatch.setenv('GCRNN_NO_INLINE_PACK', nopack) if nopack == '1' else monkeypatch.delenv('GCRNN_NO_INLINE_PACK', raising=False)
        with torch.no_grad():
            monkeypatch.setenv('GCRNN_SEQ_KERNEL', '1')
            hs1, _, H1 = ops.fused_cell_forward(Xp, h0, wA, cell.weight_B, cell.bias, cell.graph, return_states=True)
            Hl1 = cell(X, h0, last_only=True)
            monkeypatch.setenv('GCRNN_SEQ_KERNEL', '0')
            hs0, _, H0 = ops.fused_cell_forward(Xp, h0, wA, cell.weight_B, cell.bias, cell.graph, return_states=True)
            Hl0 = cell(X, h0, last_only=True)
        assert torch.equal(H0, H1) and torch.equal(hs0, hs1) and torch.equal(Hl0, Hl1), nopack
        assert float(hs1[:, :, N:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('tg,N,F,K', [(False, 1000, 64, 5), (True, 1000, 64, 5), (False, 400, 32, 3), (False, 1000, 64, 2)])
def test_sequence_resident_bptt_chain_is_bit_identical(tg, N, F, K, monkeypatch):
    """Training step with the forward steps and the BPTT data chain on the sequence-resident kernel (MODE 2: operand dpre_t in
    registers, epilogue operands prefetched at the last hop, inline pack of dH, forget-gate scale and the <h, chain> partials of the
    time-gated cell): every gradient has the bits of the chunk-parallel kernels."""
    dev = torch.device('cuda:0')
    G, B, T = F, 6, 5
    cell, rng, _ = _uniform_cell(N, G, F, K, tg, 73, dev, dtype=None)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16).requires_grad_(not tg)      # (the fused time gates give h0 no gradient)
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)
    monkeypatch.setenv('GCRNN_SEQ_MIN_B', '1'); monkeypatch.setenv('GCRNN_SEQ32', '0')      # (the 16-feature kernels: the wide one is pinned in tests/test_wide.py)

    def step():
        cell.zero_grad(set_to_none=True)
        h0.grad = None
        assert cell._use_fused_training(X, h0)
        H = cell(X, h0)
        torch.nn.functional.l1_loss(H.float(), tgt).backward()
        g = {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}
        if not tg:
            g['h0'] = h0.grad.clone()
        return H.detach().clone(), g

    monkeypatch.setenv('GCRNN_SEQ_KERNEL', '1')
    H1, g1 = step()
    monkeypatch.setenv('GCRNN_SEQ_KERNEL', '0')
    H0, g0 = step()
    assert torch.equal(H0, H1) and g0.keys() == g1.keys() and len(g1) >= 4
    for n in g1:
        assert torch.equal(g0[n], g1[n]), n


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T,zero_h0', [(1000, 64, 64, 5, 70, 5, True), (1008, 64, 32, 3, 300, 3, False), (400, 32, 32, 4, 7, 40, True),
                                                 (1000, 64, 64, 2, 130, 2, False)])
def test_gate_prepass_that_lays_out_the_input_is_bit_identical(N, F, G, K, B, T, zero_h0, monkeypatch):
    """gcrnn_fused_gate_prepass_pack_bf16: the FIRST gate pre-pass of a time-gated cell lays out x_t of every step beyond the first
    round of workgroups itself (item i packs the operand of item i + gridDim) -- same gates, same states, same gradients as with
    the separate pack pass over X (GCRNN_NO_INLINE_PACK=1); items = B T below, at and above 256 workgroups, a ragged last round."""
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    cell, rng, _ = _uniform_cell(N, G, F, K, True, 89, dev)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.zeros((B, F, N), dtype=torch.bfloat16, device=dev) if zero_h0 else \
        torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)
    monkeypatch.setenv('GCRNN_SEQ_MIN_B', '1'); monkeypatch.setenv('GCRNN_SEQ32', '0')      # (the 16-feature kernels: the wide one is pinned in tests/test_wide.py)
    plan16 = ops.fused_img16_plan(cell.graph, True, None)
    steps = int(ops.lib.gcrnn_fused_gate_prepass_lays_out(B, T, N, F, G, K, plan16['entries'], cell.graph.fused_plan()['uniform_w'], 1))
    assert steps == -(-min(B * T, 256) // B)

    def run():
        with torch.no_grad():
            H = cell(X, h0)
        cell.zero_grad(set_to_none=True)
        assert cell._use_fused_training(X, h0)
        Ht = cell(X, h0)
        torch.nn.functional.l1_loss(Ht.float(), tgt).backward()
        return H, Ht.detach(), {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}

    H1, Ht1, g1 = run()
    monkeypatch.setenv('GCRNN_NO_INLINE_PACK', '1')
    H0, Ht0, g0 = run()
    assert torch.equal(H1, H0) and torch.equal(Ht1, Ht0) and g0.keys() == g1.keys() and len(g1) >= 8
    for n in g1:
        assert torch.equal(g0[n], g1[n]), n


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T,tg', [(1000, 64, 64, 5, 5, 4, False), (1008, 64, 32, 3, 260, 2, True), (400, 32, 32, 3, 7, 3, False),
                                            (1000, 64, 64, 2, 3, 3, True)])
def test_node_gated_recurrence_on_the_sequence_resident_kernel_is_bit_identical(N, F, G, K, B, T, tg, monkeypatch):
    """The node-gated steps h_t = tanh(ni (A(S)x_t + b) + nf (B(S)h_{t-1} + b)) (graphML.py:2420-2423) as ONE persistent launch of the
    sequence-resident kernel (MODE 5: node gates held per tile for all chunks of a step, Yx_t gathered while the hops run, Yh_t kept for
    the backward): same states, last-state read-out and gradients as the chunk-parallel kernel's per-step launches."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    _, rng, S = _uniform_cell(N, G, F, K, False, 97, dev)
    torch.manual_seed(23)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'node', 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(dev).to(torch.bfloat16)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)
    monkeypatch.setenv('GCRNN_SEQ_MIN_B', '1'); monkeypatch.setenv('GCRNN_SEQ32', '0')      # (the 16-feature kernels: the wide one is pinned in tests/test_wide.py)
    monkeypatch.setenv('GCRNN_NO_FUSED_TAPS', '1')       # (the fused tap dots sum in another order: compared below, with a tolerance)

    def run():
        with torch.no_grad():
            assert cell._use_fused_node(X, h0)
            H = cell(X, h0)
            Hl = cell(X, h0, last_only=True)
        cell.zero_grad(set_to_none=True)
        assert cell._use_fused_training(X, h0)
        Ht = cell(X, h0)
        torch.nn.functional.l1_loss(Ht.float(), tgt).backward()
        return H, Hl, Ht.detach(), {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}

    monkeypatch.setenv('GCRNN_SEQ_KERNEL', '1')
    H1, Hl1, Ht1, g1 = run()
    monkeypatch.setenv('GCRNN_SEQ_PERSIST', '0')
    H1s, _, _, _ = run()
    monkeypatch.delenv('GCRNN_SEQ_PERSIST')
    monkeypatch.setenv('GCRNN_SEQ_KERNEL', '0')
    H0, Hl0, Ht0, g0 = run()
    assert torch.equal(H1, H0) and torch.equal(H1s, H0) and torch.equal(Hl1, Hl0) and torch.equal(Hl1, H1[:, -1:]) and torch.equal(Ht1, Ht0)
    assert g0.keys() == g1.keys() and len(g1) >= 11
    for n in g1:
        assert torch.equal(g0[n], g1[n]), n
    # inference with the F -> 1 filters' tap dots fused into the gate cells' pre-passes (gcrnn_fused_gate_prepass_taps_bf16): the same
    # bf16 states times the same fp32 taps (three exact bf16 planes), summed on the matrix cores -- per-tap dot products equal to the
    # separate pass over the stored states up to the order of the fp32 sums, states equal up to a rare bf16 rounding flip
    from gated_gcrnns_amd import ops
    monkeypatch.setenv('GCRNN_SEQ_KERNEL', '1')
    monkeypatch.delenv('GCRNN_NO_FUSED_TAPS')
    with torch.no_grad():
        Hf = cell(X, h0)
        Xp, wAp = ops.fused_pad_operands(X, cell.GRNN_node_in.weight_A.detach()) if hasattr(cell, 'GRNN_node_in') else (None, None)
    d = (Hf.float() - H1.float()).abs()
    assert float(d.max()) <= 8e-3 and float((d > 0).float().mean()) <= 1e-3, (float(d.max()), float((d > 0).float().mean()))
    sub = cell.GRNN_node_in if hasattr(cell, 'GRNN_node_in') else None
    if sub is not None:
        xs = ops.to_sequence_major(Xp, cell.graph)
        h0s = ops.to_sequence_major(h0.unsqueeze(1), cell.graph)
        hz = ops.fused_h0_zero_flag(h0)
        wf = cell.GFL_node_in.weight if hasattr(cell.GFL_node_in, 'weight') else cell.GFL_node_in[0].weight
        with torch.no_grad():
            sf = ops.fused_node_gate_taps(xs, h0s, wAp, sub.weight_B.detach(), sub.bias.detach(), wf.detach(), cell.graph, N, hzero=hz)
            assert sf is not None
            zl = torch.zeros((1, F * N), dtype=torch.float32, device=dev)
            _, cs, _ = ops.fused_time_gate(xs, h0s, wAp, sub.weight_B.detach(), sub.bias.detach(), zl, None, cell.graph, N, store_states=True, hzero=hz)
            wk = wf.detach().float().reshape(wf.shape[2], F)
            ref = torch.einsum('tbnf,kf->tbkn', cs[:, :, :N].double(), wk.double()).reshape(T * B, wf.shape[2], 1, N)
        err = (sf.double() - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,C,K,B,T,adj', [(1000, 64, 64, 5, 5, 3, False), (1000, 64, 64, 5, 3, 2, True), (400, 32, 32, 3, 7, 2, False),
                                             (1000, 64, 32, 2, 4, 3, False), (1008, 32, 32, 4, 130, 2, False)])
def test_filter_output_pass_on_the_sequence_resident_kernel_is_bit_identical(N, F, C, K, B, T, adj, monkeypatch):
    """The all-items filter-output pass A(S) x_t + b (graphML.py:2420: the node- and edge-gated cells' x filter, the edge-gated cell's
    per-step state filter, the input gradient dX) on the sequence-resident kernel -- one workgroup per (t, b) item, the operand a
    state-like array (C == F) or [0 | x_t] with the zero half skipped -- gives the bits of the chunk-parallel kernel, and both agree
    with an fp64 evaluation of the filter on the bf16-rounded operands."""
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    cell, rng, S = _uniform_cell(N, C, F, K, False, 83, dev)
    w = torch.tensor(bf16_round(0.2 * rng.standard_normal((F, 1, K, C))), dtype=torch.float32, device=dev)
    bias = torch.tensor(0.1 * rng.standard_normal((F, 1)), dtype=torch.float32, device=dev)
    x = torch.tensor(bf16_round(rng.standard_normal((B, T, C, N))), dtype=torch.float32, device=dev).to(torch.bfloat16)
    xs = ops.to_sequence_major(x, cell.graph)
    outs = {}
    for seq in ('1', '0'):
        monkeypatch.setenv('GCRNN_SEQ_KERNEL', seq)
        monkeypatch.setenv('GCRNN_SEQ_MIN_B', '1'); monkeypatch.setenv('GCRNN_SEQ32', '0')      # (the 16-feature kernels: the wide one is pinned in tests/test_wide.py)
        outs[seq] = ops.fused_filter_output(xs, w, bias, cell.graph, K, N, adjoint=adj)
        torch.cuda.synchronize()
    assert torch.equal(outs['1'], outs['0'])
    assert float(outs['1'][:, :, N:].abs().max()) == 0.0
    # fp64 check of one item: sum_k S^k x w_k + b (S^T on the adjoint graph)
    Sd = np.asarray(S, dtype=np.float64).reshape(N, N)
    Sd = Sd.T if adj else Sd
    xi = x[B - 1, T - 1].double().cpu().numpy()                      # [C][N]
    acc = np.zeros((F, N))
    z = xi.copy()
    wd = w.double().cpu().numpy()
    for k in range(K):
        acc += wd[:, 0, k, :] @ z
        z = z @ Sd
    ref = acc + bias.double().cpu().numpy()
    got = outs['1'][T - 1, B - 1, :N].float().cpu().numpy().T
    scale = float(np.abs(ref).max())
    assert float(np.abs(got - ref).max()) <= 2.5e-2 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T', [(1000, 64, 64, 5, 5, 4), (400, 32, 32, 3, 7, 3), (1008, 64, 64, 5, 257, 2), (1000, 64, 1, 3, 3, 3)])
def test_native_layout_output_and_sequence_major_forward_are_bit_identical(N, F, G, K, B, T, monkeypatch):
    """the cell output as a VIEW of the sequence-major state image (module flag `native_layout`: the launches skip
    the user-layout copy of every h_t) and the forward on sequence-major arrays end to end (`forward_native`: no pack, no inline
    pack) give the bits of the ordinary user-layout forward, on both step kernels."""
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    cell, rng, _ = _uniform_cell(N, G, F, K, False, 81, dev)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    for seq in ('1', '0'):
        monkeypatch.setenv('GCRNN_SEQ_KERNEL', seq)
        monkeypatch.setenv('GCRNN_SEQ_MIN_B', '1'); monkeypatch.setenv('GCRNN_SEQ32', '0')      # (the 16-feature kernels: the wide one is pinned in tests/test_wide.py)
        with torch.no_grad():
            cell.native_layout = False
            H = cell(X, h0)
            Hl = cell(X, h0, last_only=True)
            cell.native_layout = True
            Hv = cell(X, h0)
            Hvl = cell(X, h0, last_only=True)
            cell.native_layout = False
            assert Hv.shape == H.shape and not Hv.is_contiguous() and torch.equal(Hv, H) and torch.equal(Hvl, Hl)
            Xp, _ = ops.fused_pad_operands(X, cell.weight_A.detach())
            xs = ops.to_sequence_major(Xp, cell.graph)
            h0s = ops.to_sequence_major(h0.unsqueeze(1), cell.graph)[0]
            hs = cell.forward_native(xs, h0s)
            assert torch.equal(hs.permute(1, 0, 3, 2)[:, :, :, :N], H) and float(hs[:, :, N:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('N,G,F,K,B,T,tg', [(1000, 1, 20, 5, 4, 4, False), (1000, 1, 20, 5, 3, 3, True), (200, 3, 20, 4, 5, 4, False),
                                            (520, 20, 40, 2, 3, 3, True), (304, 1, 8, 4, 3, 3, False), (1000, 64, 48, 5, 3, 3, False)])
def test_fused_cell_with_state_widths_between_the_kernels(N, G, F, K, B, T, tg):
    """The reference drivers' own state width is F1 = 20 with K1 = 5 taps (kStepPredGRNNs.py:220-222): F not in {32, 64} runs on the
    fused kernels with ZERO-PADDED state channels (GGCRNNCell._state_padded: padded tap / bias / read-out rows are zero, a padded
    channel stays tanh(0) = 0 at every step); F = 32 with K = 4 has its instantiations now. Forward against the fp64 oracle,
    training (every parameter incl. the time gates' sub-networks) against fp32 autograd on the composed path; the state_dict keeps
    the reference's keys and shapes, the caller's RNG stream is not touched by the shadow cell."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    S = random_graph(N, min(0.5, 10.0 / N), 73)
    rng = np.random.default_rng(19)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    dH = bf16_round(rng.standard_normal((B, T, F, N)))
    torch.manual_seed(33)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).to(torch.float32)
    keys = {k: tuple(v.shape) for k, v in cell.state_dict().items()}
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    Href = orc.ggcrnn_cell(params, S.astype(np.float32).astype(np.float64), X, h0, tg, None)
    ref = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
    ref.addGSO(torch.tensor(S))
    ref.load_state_dict(cell.state_dict())
    ref = ref.to(dev)
    dHd = torch.tensor(dH, dtype=torch.float32, device=dev)
    (ref(torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev)) * dHd).sum().backward()
    cell = cell.to(dev)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev)
    hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        cell = cell.to(torch.bfloat16)
        assert cell._state_padded(Xd, hd) is not None
        rng_state = torch.random.get_rng_state()
        Hi = cell(Xd, hd)                                        # inference kernels (bf16 parameters)
        assert torch.equal(torch.random.get_rng_state(), rng_state)
        Hl = cell(Xd, hd, last_only=True)
        cell = cell.to(torch.float32)
    assert Hi.shape == (B, T, F, N) and Hi.is_contiguous() and torch.equal(Hl, Hi[:, -1:])
    err = np.abs(Hi.double().cpu().numpy() - Href)
    assert err.max() <= 2.5e-2 and err.mean() <= 1.5e-3, (err.max(), err.mean())      # (G = 1 cases: see test_fused_cell_with_few_input_features)
    H = cell(Xd, hd)
    (H.float() * dHd).sum().backward()
    got = dict(cell.named_parameters())
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert got[n].grad is None or float(got[n].grad.abs().max()) == 0.0, n
            continue
        g, gr = got[n].grad.float(), p.grad
        assert g.shape == gr.shape, n
        s = float(gr.abs().max())
        e = (g - gr).abs()
        assert s > 0 and float(e.max()) <= 4e-2 * s and (e.numel() < 16 or float(e.mean()) <= 8e-3 * s), (n, float(e.max()) / s, float(e.mean()) / s)
    assert {k: tuple(v.shape) for k, v in cell.state_dict().items()} == keys          # the shadow cell is not a sub-module


@pytest.mark.gpu
def test_few_input_channels_are_padded_by_the_pack_not_by_a_copy_of_X():
    """G = 1 in inference: the pack kernel writes the kernels' 32 input channels itself (gcrnn_pack_seq_major_padded); the bits are
    those of the old route (zero-padded copy of X in the user layout, then pack + inline pack)."""
    from gated_gcrnns_amd import ops
    dev = torch.device('cuda:0')
    for (N, G, F, K, B, T) in [(1000, 1, 64, 5, 5, 4), (600, 3, 32, 3, 4, 3), (1000, 20, 64, 4, 3, 2)]:
        cell, rng, _ = _uniform_cell(N, G, F, K, False, 91, dev)
        X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
        h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
        with torch.no_grad():
            H1 = cell(X, h0)
            Xp, wA = ops.fused_pad_operands(X, cell.weight_A.detach())
            H0 = ops.fused_cell_forward(Xp, h0, wA, cell.weight_B, cell.bias, cell.graph)
            xs = ops.fused_pack_inputs(X, h0, cell.graph, channels=Xp.shape[2])[0]
            assert torch.equal(xs, ops.to_sequence_major(Xp, cell.graph))
        assert torch.equal(H0, H1)


@pytest.mark.gpu
@pytest.mark.parametrize('loss', ['sum', 'l1'])
def test_fp32_accurate_fused_training_matches_reference_autograd_fixture(golden, loss):
    """G11: the north_star's 1e-5 mode for TRAINING on fused kernels -- x3 forward, x3 data chain (three bf16
    planes per operand, six partial products, fp32 hops) and the exact-fp32 weight gradient -- against the REFERENCE's autograd in
    fp64 (tests/golden/make_golden.py g11_fused_f32: Utils/graphML.py:2336-2427, fp32-representable operands, uniform-weight
    graph). H <= 1e-5 abs; every gradient <= 2e-5 of its max (L1: dH = sign(H - target) / count flips where H's fp32 error crosses the
    target -- none here, the margins are > 1e-5)."""
    g = golden('g11_fused_f32')
    dev = torch.device('cuda:0')
    cell, S = _g9_cell(g, False, None, dev)
    X = torch.tensor(g['X'], dtype=torch.float32, device=dev)
    h0 = torch.tensor(g['h0'], dtype=torch.float32, device=dev, requires_grad=True)
    assert cell._use_fused_x3_training(X, h0)
    H = cell(X, h0)
    assert H.dtype == torch.float32
    assert float((H.detach().double().cpu() - torch.tensor(g['H'])).abs().max()) <= 1e-5
    if loss == 'sum':
        H.sum().backward()
        want, want_h0 = g['grad_sum'], g['grad_sum_h0']
    else:
        torch.nn.functional.l1_loss(H, torch.tensor(g['target'], device=dev)).backward()
        want, want_h0 = g['grad_l1'], g['grad_l1_h0']
    got = dict(cell.named_parameters())
    for k, gr in want.items():
        e = np.abs(got[k].grad.double().cpu().numpy() - gr)
        sc = float(np.abs(gr).max())
        assert sc > 0 and e.max() <= 2e-5 * sc, (k, e.max() / sc)
    e = (h0.grad.double().cpu() - torch.tensor(want_h0)).abs()
    assert float(e.max()) <= 2e-5 * float(np.abs(want_h0).max()), float(e.max()) / float(np.abs(want_h0).max())


@pytest.mark.gpu
@pytest.mark.parametrize('loss', ['sum', 'l1'])
def test_fp32_accurate_fused_time_gated_training_matches_reference_autograd_fixture(golden, loss):
    """G12: the north_star's 1e-5 mode for what the drivers train -- the TIME-GATED cell (the reference's default,
    Utils/graphML.py:2196) -- on fused kernels: gate cells and scaled recurrence on the x3 step kernel, gated x3 data chain (the forget
    gate's gradient read off it), d gi off one x3 filter pass, exact-fp32 weight gradients with the gates as operand weights, the gate
    cells' BPTT over the items; against the REFERENCE's autograd in fp64 (tests/golden/make_golden.py g12_fused_f32_time:
    Utils/graphML.py:2336-2427 under torch autograd, fp32-representable operands). H <= 1e-5 abs; every one of the 13 gradients <= 2e-5
    of its max."""
    g = golden('g12_fused_f32_time')
    dev = torch.device('cuda:0')
    cell, S = _g9_cell(g, True, None, dev)
    X = torch.tensor(g['X'], dtype=torch.float32, device=dev)
    h0 = torch.tensor(g['h0'], dtype=torch.float32, device=dev)
    assert cell._use_fused_x3_training(X, h0, time_gated=True) and not cell._use_fused_x3_training(X, h0)
    H = cell(X, h0)
    assert H.dtype == torch.float32 and type(H.grad_fn).__name__ == '_FusedTimeCellX3Backward'
    assert float((H.detach().double().cpu() - torch.tensor(g['H'])).abs().max()) <= 1e-5
    if loss == 'sum':
        H.sum().backward()
        want = g['grad_sum']
    else:
        torch.nn.functional.l1_loss(H, torch.tensor(g['target'], device=dev)).backward()
        want = g['grad_l1']
    got = dict(cell.named_parameters())
    assert len(want) == 13
    report = {}
    for k, gr in want.items():
        e = np.abs(got[k].grad.double().cpu().numpy() - gr)
        sc = float(np.abs(gr).max())
        report[k] = e.max() / sc
        assert sc > 0 and e.max() <= 2e-5 * sc, (k, e.max() / sc)
    _tol_report('x3 time-gated training vs G12 (%s): worst gradient error / max = %.2e (%s)' % (loss, max(report.values()), max(report, key=report.get)))


@pytest.mark.gpu
@pytest.mark.parametrize('name,tg', [('g13_fused_f32_node', False), ('g13_fused_f32_time_node', True)])
def test_fp32_accurate_node_gated_forward_matches_reference_fixture(golden, name, tg):
    """G13: the NODE-gated cell (Utils/graphML.py:2379-2407, 2420-2423; with and without the time gates) at the
    north_star's 1e-5 on the fp32-accurate fused kernels -- gate cells as one-step x3 cells on the planes of X, their F -> 1 filters on the
    fp32 filter kernels, A(S) x_t and B(S) h_{t-1} as x3 filter passes, the per-node gating and tanh in fp32 (ops.fused_node_cell_forward_x3)
    -- against the REFERENCE's fp64 states (tests/golden/make_golden.py g13_fused_f32_node: fp32-representable operands, non-zero h0)."""
    g = golden(name)
    dev = torch.device('cuda:0')
    cell, S = _g9_cell(g, tg, 'node', dev)
    X = torch.tensor(g['X'], dtype=torch.float32, device=dev)
    h0 = torch.tensor(g['h0'], dtype=torch.float32, device=dev)
    calls = []
    from gated_gcrnns_amd import ops
    orig = ops.fused_node_cell_forward_x3
    ops.fused_node_cell_forward_x3 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            assert cell._use_fused_x3_node(X, h0)
            H = cell(X, h0)
            Hl = cell(X, h0, last_only=True)
    finally:
        ops.fused_node_cell_forward_x3 = orig
    assert len(calls) == 2 and H.dtype == torch.float32
    err = float((H.double().cpu() - torch.tensor(g['H'])).abs().max())
    _tol_report('x3 node-gated forward vs %s: max |H - reference| = %.2e' % (name, err))
    assert err <= 1e-5, err
    assert torch.equal(H[:, -1:], Hl)


@pytest.mark.gpu
@pytest.mark.parametrize('name,tg', [('g14_fused_f32_edge', False), ('g14_fused_f32_time_edge', True)])
def test_fp32_accurate_edge_gated_forward_matches_reference_fixture(golden, name, tg):
    """G14: the EDGE-gated cell (Utils/graphML.py:2409-2416, 2420-2423, graphAttention :521-627; with and without the time gates) at the
    north_star's 1e-5 with both filters as x3 filter passes, the attentions on the fp32 CSR edge-softmax kernels and the time gates' cells
    as one-step x3 cells (ops.fused_edge_cell_forward_x3) -- against the REFERENCE's fp64 states (tests/golden/make_golden.py
    g14_fused_f32_edge: fp32-representable operands, non-zero h0, mixers scaled so that the attention is far from uniform)."""
    g = golden(name)
    dev = torch.device('cuda:0')
    cell, S = _g9_cell(g, tg, 'edge', dev)
    X = torch.tensor(g['X'], dtype=torch.float32, device=dev)
    h0 = torch.tensor(g['h0'], dtype=torch.float32, device=dev)
    calls = []
    from gated_gcrnns_amd import ops
    orig = ops.fused_edge_cell_forward_x3
    ops.fused_edge_cell_forward_x3 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            assert cell._use_fused_x3_edge(X, h0)
            H = cell(X, h0)
    finally:
        ops.fused_edge_cell_forward_x3 = orig
    assert len(calls) == 1 and H.dtype == torch.float32
    err = float((H.double().cpu() - torch.tensor(g['H'])).abs().max())
    _tol_report('x3 edge-gated forward vs %s: max |H - reference| = %.2e' % (name, err))
    assert err <= 1e-5, err


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,B,T,tg,hz', [(1000, 64, 5, 3, 3, False, True), (1000, 64, 5, 2, 3, True, False), (600, 32, 3, 4, 3, False, False)])
def test_fp32_accurate_edge_gated_forward_matches_the_oracle_at_bench_sizes(N, F, K, B, T, tg, hz):
    """The same at the bench's node count against the fp64 oracle on fp32-representable operands (uniform-weight graph, G = F)."""
    dev = torch.device('cuda:0')
    import gated_gcrnns_amd.Utils.graphML as gml
    rng = np.random.default_rng(59)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    w32 = np.float32(1.0 / np.max(np.abs(np.linalg.eigvalsh(W))))
    S = (W * np.float64(w32)).reshape(1, N, N)
    torch.manual_seed(59)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, 'edge', 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.float()
    with torch.no_grad():
        for n_, q in cell.named_parameters():
            if n_.endswith('attention.mixer'):
                q.mul_(4.0)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = f32(rng.standard_normal((B, T, F, N)))
    h0 = np.zeros((B, F, N)) if hz else f32(0.4 * rng.standard_normal((B, F, N)))
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    Href = orc.ggcrnn_cell(params, S, X, h0, tg, 'edge')
    cell = cell.to(dev)
    Xd, hd = torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev)
    with torch.no_grad():
        assert cell._use_fused_x3_edge(Xd, hd)
        H = cell(Xd, hd)
        os.environ['GCRNN_NO_X3_EDGE'] = '1'
        try:
            assert not cell._use_fused_x3_edge(Xd, hd)
            Hc = cell(Xd, hd)                    # the composed fp32 path (LSIGF + attention kernels)
        finally:
            del os.environ['GCRNN_NO_X3_EDGE']
    err = float(np.abs(H.double().cpu().numpy() - Href).max())
    errc = float(np.abs(Hc.double().cpu().numpy() - Href).max())
    _tol_report('x3 edge-gated forward vs oracle N=%d F=%d K=%d tg=%s: %.2e (composed fp32 path: %.2e)' % (N, F, K, tg, err, errc))
    assert err <= 1e-5 and errc <= 1e-5, (err, errc)


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,K,B,T,tg,hz', [(1000, 64, 5, 3, 4, False, True), (1000, 64, 5, 2, 3, True, False), (600, 32, 3, 4, 3, False, False)])
def test_fp32_accurate_node_gated_forward_matches_the_oracle_at_bench_sizes(N, F, K, B, T, tg, hz):
    """The same at the bench's node count against the fp64 oracle on fp32-representable operands (uniform-weight graph, G = F)."""
    dev = torch.device('cuda:0')
    import gated_gcrnns_amd.Utils.graphML as gml
    rng = np.random.default_rng(57)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    w32 = np.float32(1.0 / np.max(np.abs(np.linalg.eigvalsh(W))))
    S = (W * np.float64(w32)).reshape(1, N, N)
    torch.manual_seed(57)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, 'node', 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.float()
    with torch.no_grad():
        for n_, q in cell.named_parameters():
            if n_.startswith('GFL_node_'):
                q.mul_(3.0)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = f32(rng.standard_normal((B, T, F, N)))
    h0 = np.zeros((B, F, N)) if hz else f32(0.4 * rng.standard_normal((B, F, N)))
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    Href = orc.ggcrnn_cell(params, S, X, h0, tg, 'node')
    cell = cell.to(dev)
    Xd, hd = torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(h0, dtype=torch.float32, device=dev)
    with torch.no_grad():
        assert cell._use_fused_x3_node(Xd, hd)
        H = cell(Xd, hd)
    err = float(np.abs(H.double().cpu().numpy() - Href).max())
    assert err <= 1e-5, err


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T,hz', [(1000, 64, 64, 5, 3, 4, True), (1000, 64, 1, 3, 2, 3, False), (600, 32, 32, 4, 4, 3, True),
                                            (1000, 64, 32, 3, 2, 3, False), (1000, 64, 64, 2, 2, 3, False), (1000, 64, 64, 5, 2, 1, False)])
def test_fp32_accurate_fused_time_gated_training_matches_composed_autograd(N, F, G, K, B, T, hz, monkeypatch):
    """The same at the bench's sizes against the composed fp32 path (exact fp32 kernels, golden-pinned since round 1), zero h0 (every
    training loop of the reference, train_rnn.py:256) and non-zero h0, G = 1 input feature (padded channels), G = 32 < F = 64 (the d gi
    filter pass on zero-padded tap columns), K = 2 and a single time step included: every gradient <= 2e-5 of its max, H <= 1e-5; two
    runs give the same bits (no atomics)."""
    dev = torch.device('cuda:0')
    cell, rng, _ = _uniform_cell(N, G, F, K, True, 99, dev, dtype=None)
    with torch.no_grad():
        cell.MLP_in[0].weight.mul_(6.0)
        cell.MLP_forget[0].weight.mul_(6.0)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev)
    h0 = torch.zeros(B, F, N, device=dev) if hz else torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev)
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)

    def step():
        cell.zero_grad(set_to_none=True)
        H = cell(X, h0)
        torch.nn.functional.l1_loss(H, tgt).backward()
        return H.detach().clone(), {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}

    assert cell._use_fused_x3_training(X, h0, time_gated=True)
    H1, g1 = step()
    H2, g2 = step()
    assert torch.equal(H1, H2) and all(torch.equal(g1[k], g2[k]) for k in g1)
    monkeypatch.setenv('GCRNN_NO_X3_TRAINING', '1')
    assert not cell._use_fused_x3_training(X, h0, time_gated=True)
    H0, g0 = step()
    assert float((H0 - H1).abs().max()) <= 1e-5 and g0.keys() == g1.keys() and len(g1) == 13
    for k in g1:
        sc = float(g0[k].abs().max())
        if sc == 0.0:      # (zero h0: the gate cells' state taps see a zero operand)
            assert float(g1[k].abs().max()) == 0.0, k
            continue
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-5 * sc, (k, float((g0[k] - g1[k]).abs().max()) / sc)


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T,kind,tg', [(1000, 64, 64, 5, 3, 4, 'sym', False), (1000, 64, 64, 3, 2, 3, 'rw', False), (400, 32, 32, 4, 3, 3, 'sym', True),
                                                (1000, 64, 1, 5, 2, 3, 'rw', True)])
def test_fp32_accurate_fused_kernels_on_rank1_weighted_graphs(N, F, G, K, B, T, kind, tg, monkeypatch):
    """the 1e-5 mode on RANK-1-weighted graphs -- the normalised adjacencies of the reference (Utils/graphTools.py:64
    normalizeAdjacency D^-1/2 A D^-1/2; 'rw': D^-1 A, a directed weighting), scaled by the largest eigenvalue as the drivers do
    (kStepPredGRNNs.py:768). The x3 kernels run on the plan of the 0/1 pattern with the factor table of graph.fused_plan_x3 (Horner carried
    in t / b): forward <= 1e-5 against the fp64 oracle on the dense S, un-gated and time-gated; training (forward, x3 chain, exact-fp32
    weight gradients -- the adjoint plan takes the swapped factors) <= 2e-5 of each gradient's max against the composed fp32 path."""
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(41)
    W = (rng.random((N, N)) < 10.0 / N).astype(np.float64)
    W = np.triu(W, 1); W = W + W.T
    d = W.sum(axis=1); d[d == 0] = 1.0
    S = W / np.sqrt(d)[:, None] / np.sqrt(d)[None, :] if kind == 'sym' else W / d[:, None]
    S = (S / np.max(np.abs(np.linalg.eigvals(S)))).astype(np.float32).astype(np.float64).reshape(1, N, N)
    torch.manual_seed(41)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.float()
    if tg:
        with torch.no_grad():
            cell.MLP_in[0].weight.mul_(6.0)
            cell.MLP_forget[0].weight.mul_(6.0)
    X = rng.standard_normal((B, T, G, N)).astype(np.float32)
    h0 = (0.4 * rng.standard_normal((B, F, N))).astype(np.float32)
    params = {k: v.detach().double().numpy() for k, v in cell.state_dict().items()}
    ref = orc.ggcrnn_cell(params, S, X.astype(np.float64), h0.astype(np.float64), tg, None)
    cell = cell.to(dev)
    assert cell.graph.fused_plan().get('uniform_w', 0.0) == 0.0 and cell.graph.fused_plan_x3().get('rank1_x3') is not None
    Xd, hd = torch.tensor(X, device=dev), torch.tensor(h0, device=dev)
    with torch.no_grad():
        assert cell._use_fused_x3(Xd, hd, time_gated=tg)
        H = cell(Xd, hd)
    err = np.abs(H.double().cpu().numpy() - ref).max()
    assert err <= 1e-5, err
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)

    def step():
        cell.zero_grad(set_to_none=True)
        Ht = cell(Xd, hd)
        torch.nn.functional.l1_loss(Ht, tgt).backward()
        return Ht.detach().clone(), {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}

    assert cell._use_fused_x3_training(Xd, hd, time_gated=tg)
    H1, g1 = step()
    assert torch.equal(H1, H)
    monkeypatch.setenv('GCRNN_NO_X3_TRAINING', '1')
    assert not cell._use_fused_x3_training(Xd, hd, time_gated=tg)
    H0, g0 = step()
    assert g0.keys() == g1.keys() and len(g1) == (13 if tg else 3)
    worst = 0.0
    for k in g1:
        sc = float(g0[k].abs().max())
        e = float((g0[k] - g1[k]).abs().max()) / sc
        worst = max(worst, e)
        assert e <= 2e-5, (k, e)
    _tol_report('x3 on a rank-1 graph (%s, time-gated %s): H err %.2e, worst gradient error / max %.2e' % (kind, tg, err, worst))
    monkeypatch.setenv('GCRNN_NO_RANK1', '1')
    cell.graph.__dict__.pop('_fused_plan_x3', None); cell.graph.__dict__.pop('_fused_plan_x3_adj', None)
    with torch.no_grad():
        assert not cell._use_fused_x3(Xd, hd, time_gated=tg)      # without the factorisation: the composed path
    cell.graph.__dict__.pop('_fused_plan_x3', None); cell.graph.__dict__.pop('_fused_plan_x3_adj', None)


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T', [(1000, 64, 64, 5, 3, 4), (1000, 64, 1, 3, 2, 3), (600, 32, 32, 4, 4, 3)])
def test_fp32_accurate_fused_training_matches_composed_autograd(N, F, G, K, B, T, monkeypatch):
    """The same at the bench's sizes against the composed fp32 path (exact fp32 kernels, golden-pinned since round 1): every
    gradient <= 2e-5 of its max, H <= 1e-5; two runs give the same bits (no atomics)."""
    dev = torch.device('cuda:0')
    cell, rng, _ = _uniform_cell(N, G, F, K, False, 95, dev, dtype=None)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev)
    h0 = torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev, requires_grad=True)
    tgt = torch.tensor(rng.standard_normal((B, T, F, N)), dtype=torch.float32, device=dev)

    def step():
        cell.zero_grad(set_to_none=True)
        h0.grad = None
        H = cell(X, h0)
        torch.nn.functional.l1_loss(H, tgt).backward()
        g = {n: p.grad.clone() for n, p in cell.named_parameters() if p.grad is not None}
        g['h0'] = h0.grad.clone()
        return H.detach().clone(), g

    assert cell._use_fused_x3_training(X, h0)
    H1, g1 = step()
    H2, g2 = step()
    assert torch.equal(H1, H2) and all(torch.equal(g1[k], g2[k]) for k in g1)
    monkeypatch.setenv('GCRNN_NO_X3_TRAINING', '1')
    assert not cell._use_fused_x3_training(X, h0)
    H0, g0 = step()
    assert float((H0 - H1).abs().max()) <= 1e-5 and g0.keys() == g1.keys() and len(g1) >= 4
    for k in g1:
        sc = float(g0[k].abs().max())
        assert sc > 0 and float((g0[k] - g1[k]).abs().max()) <= 2e-5 * sc, (k, float((g0[k] - g1[k]).abs().max()) / sc)


@pytest.mark.gpu
@pytest.mark.parametrize('N,F,G,K,B,T,hz', [(1000, 64, 64, 5, 5, 4, False), (1000, 64, 64, 5, 4, 3, True), (400, 32, 32, 3, 7, 3, False),
                                            (1000, 64, 1, 3, 3, 3, True), (1000, 64, 64, 2, 2, 2, False)])
def test_sequence_resident_time_gated_forward_is_bit_identical(N, F, G, K, B, T, hz, monkeypatch):
    """The time-gated cell on the sequence-resident kernel: both gate pre-passes (MODE 1: one workgroup per (t, b) item, the state half
    of the operand skipped when h0 is all zeros) and the gated recurrence (GATED: h-chain, scale by gf / gi, x-chain, scale by gi)
    give the bits of the chunk-parallel kernels, for zero and non-zero h0."""
    dev = torch.device('cuda:0')
    cell, rng, _ = _uniform_cell(N, G, F, K, True, 97, dev)
    X = torch.tensor(rng.standard_normal((B, T, G, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16) if hz else \
        torch.tensor(0.3 * rng.standard_normal((B, F, N)), dtype=torch.float32, device=dev).to(torch.bfloat16)
    monkeypatch.setenv('GCRNN_SEQ_MIN_B', '1'); monkeypatch.setenv('GCRNN_SEQ32', '0')      # (the 16-feature kernels: the wide one is pinned in tests/test_wide.py)
    with torch.no_grad():
        monkeypatch.setenv('GCRNN_SEQ_KERNEL', '1')
        H1, Hl1 = cell(X, h0), cell(X, h0, last_only=True)
        monkeypatch.setenv('GCRNN_SEQ_PERSIST', '0')
        H2 = cell(X, h0)
        monkeypatch.delenv('GCRNN_SEQ_PERSIST')
        monkeypatch.setenv('GCRNN_SEQ_KERNEL', '0')
        H0, Hl0 = cell(X, h0), cell(X, h0, last_only=True)
    assert torch.equal(H0, H1) and torch.equal(H0, H2) and torch.equal(Hl0, Hl1)


@pytest.mark.gpu
@pytest.mark.parametrize('tg', [False, True])
def test_sequence_resident_kernel_full_size_matches_the_chunk_parallel_kernel(tg, monkeypatch):
    """The bench's full size (B = 256 sequences = one per CU, T = 32 steps inside ONE launch, N = 1000, K = 5, G = F = 64): the
    persistent sequence-resident kernel -- each workgroup reads its own h_t back as the next step's operand 31 times -- gives the
    bits of 32 launches of the chunk-parallel kernel, un-gated (with the inline pack) and time-gated (pre-passes + gated steps)."""
    import bench
    import gated_gcrnns_amd.Utils.graphML as gml
    dev = torch.device('cuda:0')
    N, K, T, F, B = 1000, 5, 32, 64, 256
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(F, F, K, K, torch.tanh, tg, None, 1, True)
    cell.addGSO(torch.tensor(bench.sbm_graph(N)))
    cell = cell.to(torch.bfloat16).to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    X = torch.randn(B, T, F, N, device=dev, generator=gen).to(torch.bfloat16)
    h0 = (0.3 * torch.randn(B, F, N, device=dev, generator=gen)).to(torch.bfloat16)
    monkeypatch.setenv('GCRNN_SEQ32', '0')      # (the 16-feature sequence-resident kernel: the wide one is pinned to the oracle in tests/test_wide.py)
    with torch.no_grad():
        monkeypatch.setenv('GCRNN_SEQ_KERNEL', '0')
        Href = cell(X, h0).clone()
        monkeypatch.setenv('GCRNN_SEQ_KERNEL', '1')
        for r in range(3):
            assert torch.equal(cell(X, h0), Href), r
