"""Synthetic graph + k-step-prediction data for the GCRNN drivers (counterparts of the reference recipes:
SBM graph Utils/graphTools.py:581-634, KStepPrediction Utils/dataTools.py:1200-1399). Host-side, numpy;
only what the GCRNN hot path's callers need."""
import numpy as np
import torch


def is_connected(W):
    N = W.shape[0]
    seen = np.zeros(N, bool)
    seen[0] = True
    frontier = np.array([0])
    while frontier.size:
        nxt = np.nonzero((np.abs(W[frontier]).sum(0) > 0) & ~seen)[0]
        seen[nxt] = True
        frontier = nxt
    return bool(seen.all())


def sbm_adjacency(N, nCommunities, probIntra, probInter, rng):
    """Undirected stochastic block model, communities of floor/ceil(N/C) consecutive nodes, redrawn until connected."""
    sizes = [N // nCommunities] * nCommunities
    c = 0
    while sum(sizes) < N:
        sizes[c] += 1
        c += 1
    labels = np.repeat(np.arange(nCommunities), sizes)
    P = np.where(labels[:, None] == labels[None, :], probIntra, probInter)
    while True:
        W = np.triu((rng.random((N, N)) < P).astype(np.float64), 1)
        W = W + W.T
        if is_connected(W):
            return W


def normalised_gso(W):
    """S = W / lambda_max (kStepPredGRNNs.py:768); |.| for directed graphs (epicenterEstimation.py:619)."""
    lam = np.max(np.abs(np.linalg.eigvalsh(W))) if np.allclose(W, W.T) else np.max(np.abs(np.linalg.eigvals(W)))
    return W / lam


class KStepPrediction(object):
    """x_0 ~ U[0,1)^N, x_{t+1} = x_t A + spatial noise + temporal noise with A = W / lambda_max; a sample is the
    first `seqLen` steps, its label the same sequence K steps ahead (reference dataTools.py:1275-1302).

    samples[split]['signals' | 'labels']: n x (seqLen * N), as in the reference; getSamples returns torch tensors.
    """

    def __init__(self, W, K, nTrain, nValid, nTest, horizon, sigmaSpatial=0.1, sigmaTemporal=0.1,
                 rhoSpatial=0.0, rhoTemporal=0.0, rng=None, dataType=torch.float64, noise=None):
        """noise: optional (x0 [nTotal][N], spatial [horizon][nTotal][N], temporal [horizon][nTotal][N]) used instead of
        fresh draws -- the arrays the reference drew make this class reproduce the reference's samples exactly
        (tests/golden/g10_kstep_data.npz)."""
        rng = rng if rng is not None else np.random.default_rng()
        N = W.shape[0]
        self.N, self.K, self.horizon, self.seqLen = N, K, horizon, horizon - K
        self.nTrain, self.nValid, self.nTest = nTrain, nValid, nTest
        A = normalised_gso(W)
        nTotal = nTrain + nValid + nTest
        if noise is not None:
            x_t, spat_in, tempNoise = (np.asarray(a, dtype=np.float64) for a in noise)
            assert x_t.shape == (nTotal, N) and spat_in.shape == (horizon, nTotal, N) and tempNoise.shape == (horizon, nTotal, N)
        else:
            covT = sigmaTemporal ** 2 * np.eye(horizon) + rhoTemporal ** 2 * np.ones((horizon, horizon))
            tempNoise = rng.multivariate_normal(np.zeros(horizon), covT, (nTotal, N)).transpose(2, 0, 1)
            x_t = rng.random((nTotal, N))
        xs = [x_t]
        for t in range(horizon):
            if noise is not None:
                spatial = spat_in[t]
            elif rhoSpatial == 0.0:
                spatial = sigmaSpatial * rng.standard_normal((nTotal, N))
            else:
                covS = sigmaSpatial ** 2 * np.eye(N) + rhoSpatial ** 2 * np.ones((N, N))
                spatial = rng.multivariate_normal(np.zeros(N), covS, nTotal)
            x_t = x_t @ A + spatial + tempNoise[t]
            xs.append(x_t)
        x = np.concatenate(xs, axis=1)
        labels = x[:, K * N:horizon * N]
        signals = x[:, 0:(horizon * N - K * N)]
        self.samples = {}
        lo = 0
        for split, n in (('train', nTrain), ('valid', nValid), ('test', nTest)):
            self.samples[split] = {'signals': torch.tensor(signals[lo:lo + n], dtype=dataType),
                                   'labels': torch.tensor(labels[lo:lo + n], dtype=dataType)}
            lo += n

    def getSamples(self, split, idx=None):
        s = self.samples[split]
        if idx is None:
            return s['signals'], s['labels']
        return s['signals'][idx], s['labels'][idx]

    @staticmethod
    def evaluate(yHat, y):
        from .miscTools import batchTimeMSELoss
        return batchTimeMSELoss(yHat, y)


def kstep_prediction_on_device(S, K, n, horizon, device, dtype=torch.float32, sigmaSpatial=0.1, sigmaTemporal=0.1,
                               generator=None, noise=None):
    """The KStepPrediction recipe (reference dataTools.py:1282-1302: x_0 ~ U[0,1)^N, x_{t+1} = x_t A + spatial noise +
    temporal noise) generated ON the device with the same CSR SpMM the filters use -- no host pass, no dense A, so it also
    serves graphs whose dense GSO would not fit (BASELINE configs[1], [4]).

    S: the already normalised GSO (E x N x N tensor / array with E = 1, or a GraphOperator). Returns
    (signals, labels), both n x seqLen x N with seqLen = horizon - K, labels = the sequence K steps ahead.
    noise: optional (x0 [N][n], spatial [horizon][N][n], temporal [horizon][N][n]) for reproducible tests."""
    from .. import ops
    from ..graph import as_operator, GraphOperator
    graph = S if isinstance(S, GraphOperator) else as_operator(torch.as_tensor(np.asarray(S) if not isinstance(S, torch.Tensor) else S))
    graph = graph.to(device)
    N = graph.N
    if noise is None:
        x = torch.rand((1, N, n), device=device, dtype=dtype, generator=generator)
        spatial = sigmaSpatial * torch.randn((horizon, 1, N, n), device=device, dtype=dtype, generator=generator)
        temporal = sigmaTemporal * torch.randn((horizon, 1, N, n), device=device, dtype=dtype, generator=generator)
    else:
        x = noise[0].to(device=device, dtype=dtype).reshape(1, N, n).contiguous()
        spatial = noise[1].to(device=device, dtype=dtype).reshape(horizon, 1, N, n)
        temporal = noise[2].to(device=device, dtype=dtype).reshape(horizon, 1, N, n)
    xs = torch.empty((horizon + 1, N, n), device=device, dtype=dtype)
    xs[0] = x[0]
    csr = graph.fwd[0]                                   # row-vector shift x A on node-major data = CSR(A^T) SpMM
    for t in range(horizon):
        nxt = (spatial[t] + temporal[t]).contiguous()
        ops.spmm_raw(csr, x.contiguous(), out=nxt, accumulate=True)          # nxt += x A
        xs[t + 1] = nxt[0]
        x = nxt
    seq = xs.permute(2, 0, 1)                            # n x (horizon + 1) x N
    return seq[:, 0:horizon - K].contiguous(), seq[:, K:horizon].contiguous()
