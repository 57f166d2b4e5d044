"""GatedGCRNN architectures on the MI355X-native cell (mirror of the reference's
Modules/architectures.py:1405-1859 for the hot path).

Same positional constructor signatures, attribute names and state_dict keys
(`stateGCRNN.*`, `outputNN.<i>.*`). Only the MLP output heads are provided (the
drivers' 'multipMlp' / 'oneMlp'); the reference's optional Selection/Aggregation-GNN
heads belong to model families outside the hot path.
"""
import numpy as np
import torch
import torch.nn as nn

from ..Utils import graphML as gml
from .. import ops


def _as_gso_tensor(GSO):
    assert len(GSO.shape) == 2 or len(GSO.shape) == 3
    if len(GSO.shape) == 2:
        assert GSO.shape[0] == GSO.shape[1]
        GSO = GSO.reshape([1, GSO.shape[0], GSO.shape[1]])       # 1 x N x N (reference :1501-1506)
    else:
        assert GSO.shape[1] == GSO.shape[2]
    return torch.tensor(np.asarray(GSO)) if not isinstance(GSO, torch.Tensor) else GSO


def _apply_mlp_rows(mlp, x2d):
    """nn.Sequential of Linear / activation modules applied to rows, Linear layers through _RowLinearFn."""
    for layer in mlp:
        if isinstance(layer, nn.Linear):
            x2d = ops.row_linear(x2d, layer.weight, layer.bias)
        else:
            x2d = layer(x2d)
    return x2d


def _to_param_dtype(h, mlp):
    """bf16 states from the fused cell meeting fp32 master weights in a library-GEMM head: compute in the parameters' dtype."""
    for p in mlp.parameters():
        return h if h.dtype == p.dtype else h.to(p.dtype)
    return h


def _build_mlp(dimInputMLP, dimLayersMLP, sigma2, sigma3, bias):
    fc = []
    if len(dimLayersMLP) > 0:
        fc.append(nn.Linear(dimInputMLP, dimLayersMLP[0], bias=bias))
        for l in range(len(dimLayersMLP) - 1):
            fc.append(sigma2())
            fc.append(nn.Linear(dimLayersMLP[l], dimLayersMLP[l + 1], bias=bias))
    if sigma3 is not None:
        fc.append(sigma3())
    return nn.Sequential(*fc)


class _GatedGCRNNBase(nn.Module):
    def _init_state(self, inFeatures, stateFeatures, inputFilterTaps, stateFilterTaps, stateNonlinearity,
                    outputNonlinearity, dimLayersMLP, GSO, bias, time_gating, spatial_gating, finalNonlinearity,
                    dimNodeSignals, nFilterTaps):
        S = _as_gso_tensor(GSO)
        self.F_i, self.K_i = inFeatures, inputFilterTaps
        self.F_h, self.K_h = stateFeatures, stateFilterTaps
        self.E, self.N = int(S.shape[0]), int(S.shape[1])
        self.bias = bias
        self.time_gating = time_gating
        self.spatial_gating = spatial_gating
        self.register_buffer('S', S, persistent=False)            # moved by .to(); not in state_dict (as the reference)
        self.sigma1 = stateNonlinearity
        self.stateGCRNN = gml.GGCRNNCell(self.F_i, self.F_h, self.K_i, self.K_h, self.sigma1, self.time_gating,
                                         self.spatial_gating, self.E, self.bias)
        self.stateGCRNN.addGSO(self.S)
        self.dimLayersMLP = dimLayersMLP
        self.sigma2 = outputNonlinearity
        self.sigma3 = finalNonlinearity
        self.F_o = dimNodeSignals
        self.K_o = nFilterTaps
        if dimNodeSignals is not None or nFilterTaps is not None:
            raise NotImplementedError('GNN output heads (SelectionGNN / AggregationGNN) are outside the GCRNN hot path; '
                                      'use the MLP heads (dimNodeSignals=None, nFilterTaps=None)')


class GatedGCRNNforRegression(_GatedGCRNNBase):
    """State cell + MLP head on every h_t (reference architectures.py:1405-1645).

    forward(x: B x T x F_i x N, h0: B x F_h x N) -> B x T x 1 x (N*out).
    """

    def __init__(self, inFeatures, stateFeatures, inputFilterTaps, stateFilterTaps, stateNonlinearity,
                 outputNonlinearity, dimLayersMLP, GSO, bias, time_gating=True, spatial_gating=None,
                 mlpType='oneMlp', finalNonlinearity=None, dimNodeSignals=None, nFilterTaps=None,
                 nSelectedNodes=None, poolingFunction=None, poolingSize=None, maxN=None):
        super().__init__()
        self._init_state(inFeatures, stateFeatures, inputFilterTaps, stateFilterTaps, stateNonlinearity,
                         outputNonlinearity, dimLayersMLP, GSO, bias, time_gating, spatial_gating,
                         finalNonlinearity, dimNodeSignals, nFilterTaps)
        self.mlpType = mlpType
        dimInputMLP = self.N * self.F_h if mlpType == 'oneMlp' else self.F_h     # reference :1545-1554
        assert mlpType in ('oneMlp', 'multipMlp')
        self.outputNN = _build_mlp(dimInputMLP, self.dimLayersMLP, self.sigma2, self.sigma3, self.bias)

    def forward(self, x, h0):
        batchSize, seqLength = x.shape[0], x.shape[1]
        if self.mlpType == 'multipMlp' and not torch.is_grad_enabled() and len(self.outputNN) == 1 and \
                isinstance(self.outputNN[0], nn.Linear) and self.outputNN[0].out_features == 1:
            # inference with the drivers' head (dimLayersMLP = [1]): fused onto the cell's h_t store, H is never materialised
            y = self.stateGCRNN.forward_with_head(x, h0, self.outputNN[0].weight, self.outputNN[0].bias)
            if y is not None:
                return y.to(x.dtype)
        H = self.stateGCRNN(x, h0)                                  # B x T x F_h x N
        flatH = H.reshape(-1, self.F_h, self.N)
        if self.mlpType == 'multipMlp':
            # one perceptron shared by all nodes (reference :1616-1627 loops over nodes; here one batched GEMM)
            assert self.F_h > 1, "the reference's per-node squeeze() breaks for F_h = 1 (architectures.py:1622)"
            lin = self.outputNN[0] if len(self.outputNN) == 1 and isinstance(self.outputNN[0], nn.Linear) else None
            if lin is not None and ops.node_linear_supported(self.F_h, lin.out_features, flatH.dtype, self.N, lin.weight.dtype):
                # the drivers' head (dimLayersMLP = [1]): one kernel on the user layout, no transposes
                flatY = ops.node_linear(flatH, lin.weight, lin.bias)            # (BT) x out x N
                return flatY.reshape(batchSize, seqLength, -1).unsqueeze(2)
            rows = _to_param_dtype(flatH, self.outputNN).transpose(1, 2).reshape(-1, self.F_h)                  # (BT*N) x F_h
            flatY = _apply_mlp_rows(self.outputNN, rows).reshape(flatH.shape[0], self.N, -1).transpose(1, 2)   # (BT) x out x N
        else:
            flatY = self.outputNN(_to_param_dtype(flatH, self.outputNN).reshape(-1, self.F_h * self.N))
        return flatY.reshape(batchSize, seqLength, -1).unsqueeze(2)


class GatedGCRNNforClassification(_GatedGCRNNBase):
    """State cell + MLP on the last state only (reference architectures.py:1647-1859). Returns B x C logits."""

    def __init__(self, inFeatures, stateFeatures, inputFilterTaps, stateFilterTaps, stateNonlinearity,
                 outputNonlinearity, dimLayersMLP, GSO, bias, time_gating=True, spatial_gating=None,
                 finalNonlinearity=None, dimNodeSignals=None, nFilterTaps=None,
                 nSelectedNodes=None, poolingFunction=None, poolingSize=None, maxN=None):
        super().__init__()
        self._init_state(inFeatures, stateFeatures, inputFilterTaps, stateFilterTaps, stateNonlinearity,
                         outputNonlinearity, dimLayersMLP, GSO, bias, time_gating, spatial_gating,
                         finalNonlinearity, dimNodeSignals, nFilterTaps)
        self.outputNN = _build_mlp(self.N * self.F_h, self.dimLayersMLP, self.sigma2, self.sigma3, self.bias)

    def forward(self, x, h0):
        H = self.stateGCRNN(x, h0, last_only=not torch.is_grad_enabled())     # inference: only the last state is materialised
        h = H.select(1, -1)                                          # reference :1844
        return self.outputNN(_to_param_dtype(h, self.outputNN).reshape(-1, self.F_h * self.N))
