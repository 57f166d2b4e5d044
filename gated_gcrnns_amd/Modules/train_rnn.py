"""Training harness for GCRNN models on MI355X (counterpart of the reference's Modules/train_rnn.py:18-541,
SURVEY.md section 8a row H1). Reproduces: the batch partition (reference :121-135), per-epoch permutation (:193),
the B x T x N -> B x T x 1 x N input form (:218, :239), h0 = zeros(B, F, N) (:256), loss -> backward -> optimiser
step (:270-276), the per-batch metric (:288), validation every `validationInterval` steps with best/last
checkpoints named <name>Archit<label>.ckpt (reference model.py:107-118). Adds what the reference lacks:
batch-sharded data parallelism with one flat gradient all-reduce per step (parallel.FlatGradAllReduce).
"""
import os
import time

import numpy as np
import torch

from .. import ops
from ..parallel import FlatGradAllReduce, shard_range


def batch_partition(nTrain, batchSize):
    """Batch sizes and index boundaries exactly as the reference computes them (train_rnn.py:121-140)."""
    if nTrain < batchSize:
        sizes = [nTrain]
    elif nTrain % batchSize != 0:
        nBatches = int(np.ceil(nTrain / batchSize))
        sizes = [batchSize] * nBatches
        while sum(sizes) != nTrain:
            sizes[-1] -= 1
    else:
        sizes = [batchSize] * (nTrain // batchSize)
    return sizes, np.cumsum([0] + sizes).tolist()


def train_step(archit, loss_fn, optim, x, y, stateFeat, sync=None, weight=None):
    """One optimiser step on a batch x, y: B x T x 1 x N (already on the device). Returns (loss, yHat).
    weight: this rank's share of the global batch (local / global); the flat all-reduce sums the ranks' mean-loss
    gradients with these weights = the gradient of the global-batch mean. A rank whose shard is empty (global batch
    smaller than the world) contributes zeros and still joins the collective."""
    B, N = x.shape[0], x.shape[3]
    if sync is None and hasattr(optim, 'sync'):
        sync_z = optim.sync                      # optim.FlatAdam: the gradients live in its flat buffer
    else:
        sync_z = sync
    if sync_z is not None:
        sync_z.zero_grad()                       # one memset of the flat gradient buffer
    else:
        archit.zero_grad()
    loss, yHat = None, None
    if B > 0:
        h0 = torch.zeros(B, stateFeat, N, dtype=x.dtype, device=x.device)
        yHat = archit(x, h0)
        loss = loss_fn(yHat, y)
        loss.backward()
    if sync is not None:
        sync.all_reduce_(weight)
    optim.step()
    if loss is None:
        return torch.zeros((), dtype=x.dtype, device=x.device), None
    return loss.detach(), yHat.detach()


class TrainableModel(object):
    """archit + loss + optimiser + name, with the reference's checkpoint naming (model.py:107-130)."""

    def __init__(self, archit, loss, optim, name, saveDir, order=None):
        self.archit, self.loss, self.optim, self.name, self.saveDir, self.order = archit, loss, optim, name, saveDir, order

    def save(self, label=''):
        d = os.path.join(self.saveDir, 'savedModels')
        os.makedirs(d, exist_ok=True)
        torch.save(self.archit.state_dict(), os.path.join(d, self.name + 'Archit' + label + '.ckpt'))
        torch.save(self.optim.state_dict(), os.path.join(d, self.name + 'Optim' + label + '.ckpt'))

    def load(self, label=''):
        d = os.path.join(self.saveDir, 'savedModels')
        self.archit.load_state_dict(torch.load(os.path.join(d, self.name + 'Archit' + label + '.ckpt')))
        self.optim.load_state_dict(torch.load(os.path.join(d, self.name + 'Optim' + label + '.ckpt')))


def MultipleModels(modelsDict, xTrain, yTrain, xValid, yValid, nEpochs, batchSize, seqLen, stateFeat,
                   evaluate, validationInterval=5, rank=0, world=1, doPrint=False, rng=None, dataType=None):
    """Train every model of `modelsDict` (name -> TrainableModel, names containing 'GCRNN') on the same batches.

    xTrain / yTrain: nTrain x seqLen x N tensors (host or device); evaluate(yHat, y) is the dataset metric
    (batchTimeMSELoss for k-step prediction). With world > 1 each rank takes its shard of every batch and the
    gradients are averaged by one flat all-reduce. dataType: dtype of the batches on the device (default: the parameters'
    dtype); torch.bfloat16 with fp32 parameters = bf16 activations over fp32 master weights (the fused kernels).
    Returns dicts of per-step loss / metric / seconds per model.
    """
    if rng is None:
        # every rank must draw the SAME epoch permutations: with world > 1 the default generator is seeded identically on
        # all ranks (rank 0's seed would need a broadcast; a fixed seed needs none); a single process keeps numpy's global
        # state like the reference (train_rnn.py:193)
        rng = np.random.RandomState(20231) if world > 1 else np.random
    nTrain = xTrain.shape[0]
    sizes, index = batch_partition(nTrain, batchSize)
    dev = next(iter(modelsDict.values())).archit.stateGCRNN.weight_A.device
    dt = dataType if dataType is not None else next(iter(modelsDict.values())).archit.stateGCRNN.weight_A.dtype
    syncs = {k: ((m.optim.sync if hasattr(m.optim, 'sync') else FlatGradAllReduce(m.archit.parameters())) if world > 1 else None)
             for k, m in modelsDict.items()}
    lossTrain = {k: [] for k in modelsDict}
    evalTrain = {k: [] for k in modelsDict}
    evalValid = {k: [] for k in modelsDict}
    timeTrain = {k: [] for k in modelsDict}
    best = {}
    for epoch in range(nEpochs):
        perm = [int(i) for i in rng.permutation(nTrain)]
        for b in range(len(sizes)):
            idx = perm[index[b]:index[b + 1]]
            nGlobal = len(idx)
            lo, hi = shard_range(nGlobal, rank, world)
            idx = idx[lo:hi]
            share = len(idx) / float(nGlobal)                                     # this rank's weight in the flat all-reduce
            # (explicit last dimension: an EMPTY shard -- a global batch smaller than the world -- cannot infer a -1)
            xb = xTrain[idx].reshape(len(idx), seqLen, xTrain[0].numel() // seqLen).to(dev, dt)
            yb = yTrain[idx].reshape(len(idx), seqLen, yTrain[0].numel() // seqLen).to(dev, dt)
            for key, m in modelsDict.items():
                assert 'GCRNN' in key or 'gcrnn' in key or 'GCRnn' in key        # reference dispatches on the name
                xo = xb[:, :, m.order] if m.order is not None else xb
                xo, yo = xo.unsqueeze(2), yb.unsqueeze(2)                         # B x T x 1 x N
                torch.cuda.synchronize() if dev.type == 'cuda' else None
                t0 = time.perf_counter()
                loss, yHat = train_step(m.archit, m.loss, m.optim, xo, yo, stateFeat, syncs[key], share if world > 1 else None)
                torch.cuda.synchronize() if dev.type == 'cuda' else None
                timeTrain[key].append(time.perf_counter() - t0)
                lossTrain[key].append(float(loss))
                evalTrain[key].append(float(evaluate(yHat.to(yo.dtype), yo)) if yHat is not None else float('nan'))
            step = epoch * len(sizes) + b
            if validationInterval and step % validationInterval == 0 and xValid is not None:
                xv0 = xValid.view(xValid.shape[0], seqLen, -1).to(dev, dt)
                yv = yValid.view(yValid.shape[0], seqLen, -1).to(dev, dt).unsqueeze(2)
                for key, m in modelsDict.items():
                    xv = (xv0[:, :, m.order] if m.order is not None else xv0).unsqueeze(2)      # reference train_rnn.py:349
                    with torch.no_grad():
                        h0 = torch.zeros(xv.shape[0], stateFeat, xv.shape[3], dtype=dt, device=dev)
                        score = float(evaluate(m.archit(xv, h0).to(yv.dtype), yv))
                    evalValid[key].append(score)
                    if key not in best or score < best[key]:
                        best[key] = score
                        if rank == 0:
                            m.save(label='Best')
                    if doPrint and rank == 0:
                        print('[E %d B %d] %s valid %.4f' % (epoch + 1, b + 1, key, score))
        if rank == 0:
            for m in modelsDict.values():
                m.save(label='Last')
    return dict(lossTrain=lossTrain, evalTrain=evalTrain, evalValid=evalValid, timeTrain=timeTrain, bestScore=best)


class GraphedTrainStep(object):
    """One optimiser step (zero_grad -> forward -> loss -> BPTT -> [flat gradient all-reduce] -> Adam) captured as hipGraphs and replayed.

    The reference's training configurations (N = 50..80 nodes, T = 5..20, batch 100; kStepPredGRNNs.py:110-127) are
    launch-bound on a GPU: a step is several hundred tiny kernels. Capturing the whole step removes the per-launch host
    cost. Inputs are copied into static buffers; the optimiser is optim.FlatAdam (one kernel over the flat buffers, device step
    counter: capturable as it is) or torch.optim.Adam created with capturable=True.

    One process (`sync=None`): ONE graph holds the whole step. Batch-sharded over several ranks (`sync` = the parallel.FlatGradAllReduce whose
    views are the parameters' `.grad`, `optim.sync` for FlatAdam): TWO graphs -- [zero_grad, forward, loss, BPTT] and [optimiser step] -- with the
    ONE flat all-reduce between them, issued eagerly on the replay stream (the collective's position is the reference's
    `loss.backward()` -> `optim.step()`, Modules/train_rnn.py:273-276; a collective inside a captured graph would tie the graph to one
    communicator state). `weight` = local batch / global batch of this rank (default 1 / world). On a CPU device (the gloo tests) nothing can
    be captured: the same two segments run eagerly, so the N > 1 logic -- what is reduced, when, with which weight -- is the code under test.

        step = GraphedTrainStep(archit, loss_fn, optim, x_example, y_example, stateFeat)
        loss, yHat = step(x, y)        # x, y: B x T x 1 x N on the device, same shapes as the examples
    """

    def __init__(self, archit, loss_fn, optim, x, y, stateFeat, sync=None, weight=None):
        self.x = x.clone()
        self.y = y.clone()
        self.archit, self.loss_fn, self.optim = archit, loss_fn, optim
        self.sync, self.weight = sync, weight
        B, N = x.shape[0], x.shape[3]
        self.h0 = torch.zeros(B, stateFeat, N, dtype=x.dtype, device=x.device)
        self.flat = hasattr(self.optim, 'sync')          # optim.FlatAdam: gradients are views of one flat buffer, step counter on the device
        assert sync is None or self.flat and sync is self.optim.sync or not self.flat, 'FlatAdam reduces its own flat buffer: pass sync=optim.sync'
        self.captured = x.device.type == 'cuda'
        if not self.captured:
            self.graph = self.graph_step = None
            return
        s = torch.cuda.Stream(device=x.device)
        s.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(s):
            for _ in range(3):                                   # warm-up: allocator, lazily created optimiser state, the communicator
                self._eager()
        torch.cuda.current_stream(x.device).wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        if not self.flat:
            self.optim.zero_grad(set_to_none=True)
        # Captured on the warm-up's OWN side stream (as ops.FusedForwardGraph does). Captured on torch.cuda.graph's default stream -- another one
        # than the stream the warm-up created the autograd accumulators on -- every capture AFTER THE FIRST of a process replayed a step whose
        # forward was not the eager one: same parameters, loss 0.2059 against the eager 0.1611, the same wrong value on every replay, for the
        # time-gated N = 80 model at B = 256, F = 64 (round 5: tools/experiments/train_sweep_debug3.py; tools/train_steps_sweep.py caught it as
        # a loss trajectory one step behind). With one stream for warm-up and capture every run replays the eager step bit for bit.
        with torch.cuda.graph(self.graph, stream=s):
            self._segment_backward(first_capture=not self.flat)
            if sync is None:
                self.optim.step()
        self.graph_step = None
        if sync is not None:
            self.graph_step = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_step, stream=s):
                self.optim.step()

    def _segment_backward(self, first_capture=False):
        if self.flat:
            self.optim.zero_grad()                   # one memset node; backward accumulates into the views in place
        elif self.sync is not None:
            self.sync.zero_grad()
        elif not first_capture:
            self.optim.zero_grad(set_to_none=True)
        self.yHat = self.archit(self.x, self.h0)
        self.loss = self.loss_fn(self.yHat, self.y)
        self.loss.backward()

    def _eager(self):
        self._segment_backward()
        if self.captured:
            # (warm-up only) The gradients are persistent views of the flat buffer: every backward ends in AccumulateGrad nodes that ADD in place,
            # on the stream their accumulators were created on. The optimiser step behind them waits for the device here, whatever stream
            # autograd chose (PyTorch warns "AccumulateGrad node's stream does not match" on this pattern).
            torch.cuda.synchronize(self.x.device)
        if self.sync is not None:
            self.sync.all_reduce_(self.weight)
        self.optim.step()
        return self.loss

    def __call__(self, x, y):
        self.x.copy_(x)
        self.y.copy_(y)
        if not self.captured:
            self._eager()
        else:
            self.graph.replay()
            if self.graph_step is not None:
                self.sync.all_reduce_(self.weight)       # between the two graphs, on the replay stream (stream order: behind graph 1, in front of graph 2)
                self.graph_step.replay()
        ops.parameters_changed()      # (the replayed optimiser step wrote the parameters without moving their version counters)
        return self.loss.detach(), self.yHat.detach()
