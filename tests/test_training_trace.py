"""GPU: the training-loop counterpart reproduces the reference's first 20 Adam steps
(golden G6: per-step L1 loss and RMSE-like metric on fixed graph / data / initial parameters)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('flat', [False, True])
@pytest.mark.parametrize('name,tg', [('GCRNNMLP', False), ('TimeGCRNNMLP', True)])
def test_g6_twenty_adam_steps_match_reference(name, tg, flat):
    import gated_gcrnns_amd.Modules.architectures as archit
    from gated_gcrnns_amd.Modules.train_rnn import train_step
    from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss, batchTimeMSELoss
    g = load_golden('g6_trace_' + name)
    dev = torch.device('cuda:0')
    m = archit.GatedGCRNNforRegression(1, 20, 3, 3, torch.tanh, torch.nn.ReLU, [1], g['S'][0], True,
                                       time_gating=tg, spatial_gating=None, mlpType='multipMlp').double()
    m.load_state_dict({k: torch.tensor(v) for k, v in g['params0'].items()})
    m = m.to(dev)
    if flat:       # the flat-buffer Adam kernel + gradients as views of the flat all-reduce buffer (N3)
        from gated_gcrnns_amd.optim import FlatAdam
        opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    else:
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))      # kStepPredGRNNs.py:158-161
    x = torch.tensor(g['x'], device=dev)
    y = torch.tensor(g['y'], device=dev)
    losses, metrics = [], []
    for it in range(20):
        loss, yHat = train_step(m, batchTimeL1Loss, opt, x, y, 20)
        losses.append(float(loss))
        metrics.append(float(batchTimeMSELoss(yHat, y)))
    assert np.max(np.abs(np.array(losses) - g['loss'])) <= 1e-9
    assert np.max(np.abs(np.array(metrics) - g['metric'])) <= 1e-8
    sd = m.state_dict()
    for k, v in g['params20'].items():
        assert np.max(np.abs(sd[k].cpu().numpy() - v)) <= 1e-8, k


def test_harness_runs_and_checkpoints(tmp_path):
    import gated_gcrnns_amd.Modules.architectures as archit
    from gated_gcrnns_amd.Modules.train_rnn import MultipleModels, TrainableModel
    from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss, batchTimeMSELoss
    g = load_golden('g6_trace_GCRNNMLP')
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = archit.GatedGCRNNforRegression(1, 20, 3, 3, torch.tanh, torch.nn.ReLU, [1], g['S'][0], True,
                                       time_gating=False, spatial_gating=None, mlpType='multipMlp').to(dev)
    tm = TrainableModel(m, batchTimeL1Loss, torch.optim.Adam(m.parameters(), lr=1e-3), 'GCRNNMLP', str(tmp_path))
    x = torch.tensor(g['x'][:, :, 0, :], dtype=torch.float32)          # nTrain x T x N
    y = torch.tensor(g['y'][:, :, 0, :], dtype=torch.float32)
    out = MultipleModels({'GCRNNMLP': tm}, x, y, x[:6], y[:6], nEpochs=2, batchSize=8, seqLen=5, stateFeat=20,
                         evaluate=batchTimeMSELoss, validationInterval=2, rng=np.random.default_rng(0))
    assert len(out['lossTrain']['GCRNNMLP']) == 2 * 3                    # 20 samples -> batches 8, 8, 4
    assert out['lossTrain']['GCRNNMLP'][-1] < out['lossTrain']['GCRNNMLP'][0]
    assert (tmp_path / 'savedModels' / 'GCRNNMLPArchitBest.ckpt').exists()
    assert (tmp_path / 'savedModels' / 'GCRNNMLPArchitLast.ckpt').exists()
    tm.load('Last')


def test_graphed_training_step_with_flat_adam_matches_eager():
    """zero_grad -> forward -> L1 loss -> BPTT -> FlatAdam captured as ONE hipGraph (device step counter, gradients accumulate into
    views of the flat buffer) reproduces the eager steps bit for bit over several replays."""
    import gated_gcrnns_amd.Modules.architectures as archit
    from gated_gcrnns_amd.Modules.train_rnn import train_step, GraphedTrainStep
    from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss
    from gated_gcrnns_amd.optim import FlatAdam
    g = load_golden('g6_trace_GCRNNMLP')
    dev = torch.device('cuda:0')
    x = torch.tensor(g['x'], device=dev)
    y = torch.tensor(g['y'], device=dev)
    ms = []
    for _ in range(2):
        m = archit.GatedGCRNNforRegression(1, 20, 3, 3, torch.tanh, torch.nn.ReLU, [1], g['S'][0], True,
                                           time_gating=False, spatial_gating=None, mlpType='multipMlp').double()
        m.load_state_dict({k: torch.tensor(v) for k, v in g['params0'].items()})
        ms.append(m.to(dev))
    opt_e, opt_g = FlatAdam(ms[0].parameters(), lr=1e-3), FlatAdam(ms[1].parameters(), lr=1e-3)
    stepper = GraphedTrainStep(ms[1], batchTimeL1Loss, opt_g, x, y, 20)            # 3 eager warm-up steps inside
    for _ in range(3):
        train_step(ms[0], batchTimeL1Loss, opt_e, x, y, 20)
    for it in range(5):
        le, _ = train_step(ms[0], batchTimeL1Loss, opt_e, x, y, 20)
        lg, _ = stepper(x, y)
        assert float(le) == float(lg), it
    for p, q in zip(ms[0].parameters(), ms[1].parameters()):
        assert torch.equal(p, q)
    assert int(opt_g.step_dev.item()) == 8


@pytest.mark.gpu
def test_graphed_training_step_as_two_graphs_around_the_flat_allreduce():
    """The multi-rank form of GraphedTrainStep (sync = the optimiser's flat gradient buffer): [zero_grad, forward, loss, BPTT] and [FlatAdam step]
    captured as TWO hipGraphs, the one flat all-reduce issued between their replays (no process group here: the collective itself is skipped,
    the scaling and the stream order are not). Losses and parameters equal the ONE-graph form bit for bit over several replays."""
    import gated_gcrnns_amd.Modules.architectures as archit
    from gated_gcrnns_amd.Modules.train_rnn import GraphedTrainStep
    from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss
    from gated_gcrnns_amd.optim import FlatAdam
    g = load_golden('g6_trace_GCRNNMLP')
    dev = torch.device('cuda:0')
    x = torch.tensor(g['x'], device=dev)
    y = torch.tensor(g['y'], device=dev)
    ms = []
    for _ in range(2):
        m = archit.GatedGCRNNforRegression(1, 20, 3, 3, torch.tanh, torch.nn.ReLU, [1], g['S'][0], True,
                                           time_gating=False, spatial_gating=None, mlpType='multipMlp').double()
        m.load_state_dict({k: torch.tensor(v) for k, v in g['params0'].items()})
        ms.append(m.to(dev))
    opt_1, opt_2 = FlatAdam(ms[0].parameters(), lr=1e-3), FlatAdam(ms[1].parameters(), lr=1e-3)
    one = GraphedTrainStep(ms[0], batchTimeL1Loss, opt_1, x, y, 20)
    two = GraphedTrainStep(ms[1], batchTimeL1Loss, opt_2, x, y, 20, sync=opt_2.sync)
    assert one.graph_step is None and two.graph_step is not None
    for it in range(5):
        l1, _ = one(x, y)
        l2, _ = two(x, y)
        assert float(l1) == float(l2), it
    for p, q in zip(ms[0].parameters(), ms[1].parameters()):
        assert torch.equal(p, q)
