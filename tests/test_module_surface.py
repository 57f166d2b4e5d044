"""CPU: the nn.Module mirror keeps the reference's surface -- constructor signatures, parameter names and
shapes, seeded initialisation order (same seed => bit-identical parameters as the reference goldens)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import gated_gcrnns_amd.Utils.graphML as gml
import gated_gcrnns_amd.Modules.architectures as archit

VARIANTS = [('none', False, None), ('time', True, None), ('node', False, 'node'),
            ('edge', False, 'edge'), ('time_node', True, 'node'), ('time_edge', True, 'edge')]


@pytest.fixture(autouse=True)
def f64_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)          # the drivers' setting (kStepPredGRNNs.py:44)
    yield
    torch.set_default_dtype(old)


@pytest.mark.parametrize('name,tg,sg', VARIANTS)
def test_cell_state_dict_and_seeded_init_match_reference(name, tg, sg):
    g = load_golden('g3_cell_' + name)
    torch.manual_seed(30)
    cell = gml.GGCRNNCell(2, 5, 3, 3, torch.tanh, tg, sg, 1, True)
    cell.addGSO(torch.tensor(g['S']))
    sd = cell.state_dict()
    assert sorted(sd) == sorted(g['params'])
    for k in sd:
        assert np.array_equal(sd[k].numpy(), g['params'][k]), k
    cell.load_state_dict({k: torch.tensor(v) for k, v in g['params'].items()})      # checkpoint compatible


def test_graphfilter_init_matches_reference():
    g = load_golden('g2_graphfilter')
    torch.manual_seed(2)
    gf = gml.GraphFilter(2, 5, 3)
    gf.addGSO(torch.tensor(g['S']))
    for k, v in gf.state_dict().items():
        assert np.array_equal(v.numpy(), g['params'][k])


@pytest.mark.parametrize('mlp,dims', [('multipMlp', [1]), ('oneMlp', [50])])
def test_regression_model_keys_and_init(mlp, dims):
    g = load_golden('g5_reg_%s_time' % mlp)
    torch.manual_seed(50)
    m = archit.GatedGCRNNforRegression(1, 20, 2, 2, torch.tanh, torch.nn.ReLU, dims, g['S'][0], True,
                                       time_gating=True, spatial_gating=None, mlpType=mlp)
    sd = m.state_dict()
    assert sorted(sd) == sorted(g['params'])
    for k in sd:
        assert np.array_equal(sd[k].numpy(), g['params'][k]), k


def test_classification_model_keys():
    g = load_golden('g5_cls_T20K4_none')
    torch.manual_seed(59)
    m = archit.GatedGCRNNforClassification(1, 20, 4, 4, torch.tanh, torch.nn.ReLU, [11], g['S'][0], True,
                                           time_gating=False, spatial_gating=None)
    sd = m.state_dict()
    assert sorted(sd) == sorted(g['params'])
    for k in sd:
        assert np.array_equal(sd[k].numpy(), g['params'][k]), k


def test_readding_gso_reinitialises_gates_like_reference():
    """Appendix B.5b: gate sub-modules are created in addGSO, so calling it again redraws them."""
    S = torch.eye(8).reshape(1, 8, 8)
    cell = gml.GGCRNNCell(1, 2, 2, 2, torch.tanh, True, None, 1, True)
    cell.addGSO(S)
    w0 = cell.GFL_in.weight_A.detach().clone()
    a0 = cell.weight_A.detach().clone()
    cell.addGSO(S)
    assert not torch.equal(cell.GFL_in.weight_A, w0)
    assert torch.equal(cell.weight_A, a0)


def test_gnn_heads_are_rejected_not_silently_ignored():
    with pytest.raises(NotImplementedError):
        archit.GatedGCRNNforRegression(1, 4, 2, 2, torch.tanh, torch.nn.ReLU, [1], np.eye(5), True,
                                       dimNodeSignals=[4, 2], nFilterTaps=[2])


def test_edge_plan_lists_are_consistent_with_the_dense_support():
    """Host logic (CPU): the packed attention support of graph.edge_plan -- rows, columns, the column -> row position map and the
    degree orders the attention kernels walk -- against the dense mask |S + I| > 1e-9 of reference graphAttention
    (Utils/graphML.py:577, 611-613), incl. a cancelled self-loop, an isolated node and a hub."""
    import numpy as np
    import torch
    from gated_gcrnns_amd.graph import GraphOperator
    rng = np.random.default_rng(4)
    N = 60
    S = (rng.random((N, N)) < 0.1) * rng.uniform(0.2, 1.0, (N, N))
    S[3, 3] = -1.0                       # S + I cancels this self-loop
    S[5, :] = 0.0; S[:, 5] = 0.0         # isolated node: only its self-loop remains
    S[9, :40] = 0.5                      # hub row
    g = GraphOperator(S.reshape(1, N, N))
    ep = g.edge_plan()
    Sp = S + np.eye(N)
    mask = np.abs(Sp) > 1e-9
    rowptr = ep['rowptr'].numpy(); r_edge = ep['r_edge'].numpy(); trp = ep['t_rowptr'].numpy(); t_edge = ep['t_edge'].numpy()
    t_pos = ep['t_pos'].numpy()
    assert ep['nnz'] == int(mask.sum()) and rowptr[-1] == ep['nnz'] and trp[-1] == ep['nnz']
    dense = np.zeros((N, N), dtype=np.float32)
    for m in range(N):
        for j in range(rowptr[m], rowptr[m + 1]):
            dense[m, r_edge[j, 0]] = r_edge[j, 1:2].copy().view(np.float32)[0]
    assert np.array_equal(dense != 0, mask) and np.max(np.abs(dense - Sp.astype(np.float32))) == 0.0
    rows_of = np.repeat(np.arange(N), np.diff(rowptr))
    for n in range(N):
        for q in range(trp[n], trp[n + 1]):
            m = t_edge[q, 0]
            assert mask[m, n] and rows_of[t_pos[q]] == m and r_edge[t_pos[q], 0] == n and t_edge[q, 1] == r_edge[t_pos[q], 1]
    outdeg, indeg = np.diff(rowptr), np.diff(trp)
    assert ep['max_out_degree'] == outdeg.max() >= 40
    for order, deg in ((ep['r_order'].numpy(), outdeg), (ep['t_order'].numpy(), indeg)):
        assert sorted(order.tolist()) == list(range(N)) and np.all(np.diff(deg[order]) <= 0)
    assert not mask[3, 3] and mask[5, 5] and outdeg[5] == 1
