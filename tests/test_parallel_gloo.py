"""CPU, world_size 2 over gloo: the batch-sharded data-parallel path (flat gradient all-reduce) reproduces the
single-process large-batch gradient and keeps replicas identical. The per-rank compute here is a small torch
model standing in for the HIP cell (no GPU in this container); what is under test is the N>1 logic itself:
sharding, flat buffer layout incl. parameters without gradient, scaling, and the harness's batch partition."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gated_gcrnns_amd.parallel import FlatGradAllReduce, shard_range, shard_batch
from gated_gcrnns_amd.Modules.train_rnn import batch_partition


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 5)
        self.unused = torch.nn.Linear(3, 2)       # like GFL_out / MLP_out: saved, never used, no gradient
        self.b = torch.nn.Linear(5, 1)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    model = Tiny().double()
    x = torch.randn(10, 6, dtype=torch.float64)
    y = torch.randn(10, 1, dtype=torch.float64)
    # reference: the whole batch in one process
    ref = Tiny().double()
    ref.load_state_dict(model.state_dict())
    torch.nn.functional.l1_loss(ref(x), y).backward()
    xs, ys = shard_batch(rank, world, x, y)
    sync = FlatGradAllReduce(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    torch.nn.functional.l1_loss(model(xs), ys).backward()
    flat = sync.all_reduce_()                    # equal shards (5 + 5): weight 1/world
    ok = True
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            ok &= bool(p.grad is not None and float(p.grad.abs().max()) == 0.0)
        else:
            ok &= bool(torch.allclose(p.grad, q.grad, atol=1e-12))
    opt.step()
    # replicas stay bit-identical after the step
    mine = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ok &= bool(all(torch.equal(g, gathered[0]) for g in gathered))
    ok &= flat.numel() == sum(p.numel() for p in model.parameters()) and flat.dtype == torch.float64      # fp64 parameters: fp64 buffer (fp32 otherwise)
    ok &= all(p.grad.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 8 for p in model.parameters())   # .grad are views of it
    ret[rank] = ok
    dist.destroy_process_group()


def test_flat_allreduce_world2_matches_large_batch():
    world = 2
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 100, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('nTrain,bs,expect', [(100, 20, [20] * 5), (95, 20, [20, 20, 20, 20, 15]), (7, 20, [7]),
                                              (41, 20, [20, 20, 1])])
def test_batch_partition_matches_reference_rule(nTrain, bs, expect):
    sizes, index = batch_partition(nTrain, bs)
    assert sizes == expect and index[-1] == nTrain and len(index) == len(sizes) + 1


def adam_worker(rank, world, port, ret):
    """FlatAdam (parameters and gradients as views of flat buffers) + the flat all-reduce with unequal shards (weights
    local / global) reproduces torch.optim.Adam on the whole batch, step after step."""
    from gated_gcrnns_amd.optim import FlatAdam
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    model = Tiny().double()
    ref = Tiny().double()
    ref.load_state_dict(model.state_dict())
    x = torch.randn(7, 6, dtype=torch.float64)            # 7 = 4 + 3: unequal shards
    y = torch.randn(7, 1, dtype=torch.float64)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    opt = FlatAdam(model.parameters(), lr=1e-2)
    lo, hi = shard_range(7, rank, world)
    ok = True
    for it in range(4):
        ropt.zero_grad()
        torch.nn.functional.l1_loss(ref(x), y).backward()
        ropt.step()
        opt.zero_grad()
        torch.nn.functional.l1_loss(model(x[lo:hi]), y[lo:hi]).backward()
        opt.sync.all_reduce_((hi - lo) / 7.0)
        opt.step()
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            ok &= bool(torch.allclose(p, q, atol=1e-12, rtol=0))
    ok &= all(p.grad.data_ptr() >= opt.sync.flat.data_ptr() for p in model.parameters())     # still views of the flat buffer
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_adam_world2_matches_torch_adam_on_the_whole_batch():
    world = 2
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(adam_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment starts its own two ranks as child processes and relays ONE
    JSON line (here with --dry-run --backend gloo: launcher, rendezvous and the MAX-over-ranks reduction, no GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '0', '--dry-run',
                        '--backend', 'gloo'], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 2 * out['config']['batch_per_gpu'] and out['steps'] == 2
    # a mismatched external launch is refused with the exact command instead of an assert
    env2 = dict(env, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True, text=True,
                       env=env2, timeout=600)
    assert r.returncode != 0 and 'torch.distributed.run' in r.stderr


class TinyGCRNN(torch.nn.Module):
    """CPU stand-in with the surface MultipleModels touches (archit.stateGCRNN.weight_A for device / dtype, archit(x, h0))."""

    def __init__(self, N):
        super().__init__()
        self.stateGCRNN = torch.nn.Module()
        self.stateGCRNN.weight_A = torch.nn.Parameter(torch.randn(N, N, dtype=torch.float64) * 0.1)

    def forward(self, x, h0):                       # x: B x T x 1 x N
        return torch.tanh(x @ self.stateGCRNN.weight_A) + 0.0 * h0.sum()


def harness_worker(rank, world, port, ret):
    """MultipleModels with nTrain % batchSize < world: the last global batch has ONE sample, so rank 1's shard is empty. It
    must still join every collective (zeros, weight 0) and end with the same parameters as rank 0 -- and as a single process
    that trains on the whole batches (the empty slice used to die in `.view(0, T, -1)` while rank 0 hung)."""
    from gated_gcrnns_amd.Modules.train_rnn import MultipleModels, TrainableModel
    from gated_gcrnns_amd.optim import FlatAdam
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    N, T, nTrain, bs = 6, 3, 5, 4                     # batches of 4 and 1: the second one leaves rank 1 without a sample
    g = torch.Generator().manual_seed(3)
    xT = torch.randn(nTrain, T, N, dtype=torch.float64, generator=g)
    yT = torch.randn(nTrain, T, N, dtype=torch.float64, generator=g)

    def build():
        torch.manual_seed(5)
        m = TinyGCRNN(N)
        return m, TrainableModel(m, torch.nn.functional.l1_loss, FlatAdam(m.parameters(), lr=1e-2), 'TinyGCRNN', '/tmp')

    # the optimiser (and with it the flat gradient buffer) is built BEFORE the process group exists: the world size must be read late
    mdl, tm = build()
    dist.init_process_group('gloo', rank=rank, world_size=world)
    out = MultipleModels({'TinyGCRNN': tm}, xT, yT, None, None, nEpochs=2, batchSize=bs, seqLen=T, stateFeat=2,
                         evaluate=lambda a, b: (a - b).abs().mean(), validationInterval=0, rank=rank, world=world,
                         rng=np.random.RandomState(11))
    mine = mdl.stateGCRNN.weight_A.detach().reshape(-1).clone()
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ok = all(torch.equal(t, gathered[0]) for t in gathered)
    dist.destroy_process_group()
    ref, rtm = build()                                # the same two epochs in one process
    MultipleModels({'TinyGCRNN': rtm}, xT, yT, None, None, nEpochs=2, batchSize=bs, seqLen=T, stateFeat=2,
                   evaluate=lambda a, b: (a - b).abs().mean(), validationInterval=0, rng=np.random.RandomState(11))
    ok &= bool(torch.allclose(mine, ref.stateGCRNN.weight_A.detach().reshape(-1), atol=1e-12, rtol=0))
    ret[rank] = bool(ok)


def test_multiple_models_survives_an_empty_shard():
    world = 2
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(harness_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_zero_grad_drops_gradients_that_live_outside_the_flat_buffer():
    """after archit.zero_grad() (grads -> None) and a backward, the gradients are fresh tensors outside the flat buffer;
    sync.zero_grad() must zero, not copy them back in."""
    torch.manual_seed(0)
    m = Tiny().double()
    sync = FlatGradAllReduce(m.parameters())
    m.zero_grad(set_to_none=True)
    m(torch.randn(4, 6, dtype=torch.float64)).sum().backward()
    assert m.a.weight.grad is not None and m.a.weight.grad.data_ptr() != sync.views[0].data_ptr()
    sync.zero_grad()
    assert float(sync.flat.abs().max()) == 0.0
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
    assert sync.world == 1
    with pytest.raises(RuntimeError):
        # a shard weight with an initialised group of ONE rank is a bug (nothing would be reduced)
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(free_port())
        dist.init_process_group('gloo', rank=0, world_size=1)
        try:
            sync.all_reduce_(0.5)
        finally:
            dist.destroy_process_group()


def graphed_worker(rank, world, port, ret):
    """train_rnn.GraphedTrainStep with a flat-gradient sync: segment 1 = zero_grad / forward / loss / BPTT, the ONE flat
    all-reduce with this rank's shard weight, segment 2 = the optimiser step. On a GPU the two segments are two captured hipGraphs with the
    collective between them; here (CPU, gloo) they run eagerly through the SAME class, so what is reduced, where and with which weight is
    the code under test: unequal shards (4 + 3) reproduce torch.optim.Adam on the whole batch, step after step, and replicas stay identical."""
    from gated_gcrnns_amd.Modules.train_rnn import GraphedTrainStep
    from gated_gcrnns_amd.optim import FlatAdam
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    N, T, Bg = 6, 3, 7
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(Bg, T, 1, N, dtype=torch.float64, generator=gen)
    y = torch.randn(Bg, T, 1, N, dtype=torch.float64, generator=gen)
    torch.manual_seed(5)
    model, ref = TinyGCRNN(N), TinyGCRNN(N)
    ref.load_state_dict(model.state_dict())
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    opt = FlatAdam(model.parameters(), lr=1e-2)
    lo, hi = shard_range(Bg, rank, world)
    l1 = torch.nn.functional.l1_loss
    step = GraphedTrainStep(model, l1, opt, x[lo:hi], y[lo:hi], 2, sync=opt.sync, weight=(hi - lo) / float(Bg))
    assert not step.captured and step.graph is None          # CPU: the two segments run eagerly
    ok = True
    for it in range(4):
        ropt.zero_grad()
        l1(ref(x, torch.zeros(Bg, 2, N, dtype=torch.float64)), y).backward()
        ropt.step()
        loss, yhat = step(x[lo:hi] + 0.0, y[lo:hi] + 0.0)     # (fresh tensors: the step copies them into its static buffers)
        ok &= bool(torch.allclose(model.stateGCRNN.weight_A, ref.stateGCRNN.weight_A, atol=1e-12, rtol=0))
        ok &= tuple(yhat.shape) == (hi - lo, T, 1, N)
    mine = model.stateGCRNN.weight_A.detach().reshape(-1).clone()
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ok &= all(torch.equal(t, gathered[0]) for t in gathered)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_graphed_train_step_world2_allreduce_between_the_two_segments():
    world = 2
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(graphed_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)
