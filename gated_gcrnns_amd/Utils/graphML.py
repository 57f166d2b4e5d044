"""Graph-ML primitives of the GCRNN hot path on MI355X (HIP kernels behind include/gcrnn.h).

Mirrors the public surface of the reference's Utils/graphML.py for the hot path
only: same names, positional signatures, parameter
names and shapes, `addGSO`, state_dict keys and AssertionError conventions.

    LSIGF(h, S, x, b=None)                                    reference graphML.py:47-140
    GraphFilter(G, F, K, E=1, bias=True)                      reference graphML.py:1086-1205
    GraphAttentional(G, F, K, E=1, nonlinearity, concatenate) reference graphML.py:1999-2128
    GGCRNNCell(G, F, Kin, Kst, sigma, time_gating, spatial_gating, E, bias)
                                                              reference graphML.py:2130-2427

Differences (all documented in DESIGN.md): the GSO is converted to CSR once in
`addGSO` (the dense tensor is kept only as a buffer for checkpoint/inspection);
data runs in a node-major layout internally; `.to(device)` moves the graph too
(the reference does not, architectures.py:1638-1645); inputs must live on a ROCm
device -- there is no CPU path.
"""
import math
import os

import torch
import torch.nn as nn

from .. import ops
from ..graph import GraphOperator, as_operator

zeroTolerance = 1e-9   # reference graphML.py:42
infiniteNumber = 1e12  # reference graphML.py:43


def LSIGF(h, S, x, b=None):
    """y = sum_e sum_k h[:, e, k, :] (x S_e^k) + b   --  drop-in for reference LSIGF (graphML.py:47-140).

    h: F x E x K x G, S: E x N x N dense tensor (or a GraphOperator), x: B x G x N, b: F x 1.
    Returns B x F x N. Shape errors raise AssertionError as in the reference (graphML.py:99-104).
    """
    F, E, K, G = h.shape
    graph = as_operator(S)
    assert graph.E == E
    N = graph.N
    B = x.shape[0]
    assert x.shape[1] == G
    assert x.shape[2] == N
    Xn = ops.pack_node_major(x.reshape(B, 1, G, N))
    if graph.device != x.device:
        graph = graph.to(x.device)
    Yn = ops.lsigf_node_major(Xn, h, b, graph, 1.0)
    return ops.unpack_node_major(Yn).reshape(B, F, N)


class GraphFilter(nn.Module):
    """Linear graph filter layer; parameters `weight` F x E x K x G, `bias` F x 1 (reference graphML.py:1086-1205)."""

    def __init__(self, G, F, K, E=1, bias=True):
        super().__init__()
        self.G, self.F, self.K, self.E = G, F, K, E
        self.S = None
        self.graph = None
        self.weight = nn.parameter.Parameter(torch.empty(F, E, K, G))
        if bias:
            self.bias = nn.parameter.Parameter(torch.empty(F, 1))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.G * self.K)          # reference graphML.py:1159-1164
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)
        ops.parameters_changed()                        # (.data writes do not move the version counters the packed-parameter cache keys on)

    def addGSO(self, S):
        graph = as_operator(S)
        assert graph.E == self.E
        self.N = graph.N
        self.S = S
        self.graph = graph

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        if self.graph is not None:
            self.graph = self.graph.to(self.weight.device)
        return self

    def forward_node_major(self, Xn):
        return ops.lsigf_node_major(Xn, self.weight, self.bias, self.graph, 1.0)

    def forward(self, x):
        B, F, Nin = x.shape
        if Nin < self.N:                                 # zero padding, reference graphML.py:1181-1185
            x = torch.cat((x, torch.zeros(B, F, self.N - Nin, dtype=x.dtype, device=x.device)), dim=2)
        u = LSIGF(self.weight, self.graph, x, self.bias)
        if Nin < self.N:                                 # reference graphML.py:1192-1193
            u = u[:, :, :Nin]
        return u

    def extra_repr(self):
        return 'in_features=%d, out_features=%d, filter_taps=%d, edge_features=%d, bias=%s, %s' % (
            self.G, self.F, self.K, self.E, self.bias is not None,
            'GSO stored' if self.graph is not None else 'no GSO stored')


def _graph_attention_node_major(u, mixer, weight, graph, negative_slope=0.2):
    """GAT on the CSR support of S + I (reference graphAttention, graphML.py:521-627), node-major.

    u: [T][N][B][G]; mixer: K x E x 2F; weight: K x E x F x G  ->  [T][N][B][K*F] (heads concatenated, pre-ReLU).
    Edge (m, n) of the support carries e = LeakyReLU(a1.Wx_n + a2.Wx_m); alpha = softmax over the
    neighbours n of row m; y_n = sum_m Wx_m (S+I)[m, n] alpha[m, n].
    """
    K, E, F, _ = weight.shape
    if u.dtype == torch.bfloat16:      # (the stand-alone attention kernels are fp32 / fp64: a bf16 layer is evaluated in fp32 and rounded once)
        return _graph_attention_node_major(u.float(), mixer.float(), weight.float(), graph, negative_slope).to(torch.bfloat16)
    outs = []
    for k in range(K):
        yk = None
        for e in range(E):
            Wx = ops.row_linear(u.reshape(-1, u.shape[3]), weight[k, e])          # (T N B) x F
            s12 = ops.row_linear(Wx, mixer[k, e].view(2, F))                      # a1 . Wx_n | a2 . Wx_m
            y = ops.edge_attention(Wx.view(*u.shape[:3], F), s12[:, 0].reshape(u.shape[:3]),
                                   s12[:, 1].reshape(u.shape[:3]), graph, e, negative_slope)
            yk = y if yk is None else yk + y
        outs.append(yk)
    return outs[0] if K == 1 else torch.cat(outs, dim=3)


class GraphAttentional(nn.Module):
    """Graph attention layer used as the edge gate (reference graphML.py:1999-2128): `mixer` K x E x 2F, `weight` K x E x F x G."""

    def __init__(self, G, F, K, E=1, nonlinearity=nn.functional.relu, concatenate=True):
        super().__init__()
        self.G, self.F, self.K, self.E = G, F, K, E
        self.S = None
        self.graph = None
        self.nonlinearity = nonlinearity
        self.concatenate = concatenate
        self.mixer = nn.parameter.Parameter(torch.empty(K, E, 2 * F))
        self.weight = nn.parameter.Parameter(torch.empty(K, E, F, G))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.G * self.K)          # reference graphML.py:2069-2073
        self.weight.data.uniform_(-stdv, stdv)
        self.mixer.data.uniform_(-stdv, stdv)
        ops.parameters_changed()

    def addGSO(self, S):
        graph = as_operator(S)
        assert graph.E == self.E
        self.N = graph.N
        self.S = S
        self.graph = graph

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        if self.graph is not None:
            self.graph = self.graph.to(self.weight.device)
        return self

    def forward_node_major(self, un):
        """un: [T][N][B][G] -> [T][N][B][K*F] (concatenate) or [T][N][B][F] (mean)."""
        y = _graph_attention_node_major(un, self.mixer, self.weight, self.graph)
        if self.concatenate:
            return self.nonlinearity(y)                                      # reference graphML.py:2101
        T, N, B, _ = y.shape
        return self.nonlinearity(y.view(T, N, B, self.K, self.F).mean(dim=3))     # reference graphML.py:2110-2112

    def forward(self, x):
        ops.require_device(x)
        B, F, Nin = x.shape
        if Nin < self.N:
            x = torch.cat((x, torch.zeros(B, F, self.N - Nin, dtype=x.dtype, device=x.device)), dim=2)
        un = ops.pack_node_major(x.reshape(B, 1, F, self.N))
        y = ops.unpack_node_major(self.forward_node_major(un)).reshape(B, -1, self.N)
        if Nin < self.N:
            y = y[:, :, :Nin]
        return y

    def extra_repr(self):
        return 'in_features=%d, out_features=%d, attention_heads=%d, edge_features=%d, %s' % (
            self.G, self.F, self.K, self.E,
            ('GSO stored: number_nodes=%d' % self.N) if self.graph is not None else 'no GSO stored')


class GGCRNNCell(nn.Module):
    """Gated graph convolutional recurrent cell (reference graphML.py:2130-2427).

        h_t = sigma( gi_t [ni_t .] (A(S) x_t + b) + gf_t [nf_t .] (B(S) h_{t-1} + b) )

    Parameters: weight_A F x E x Kin x G, weight_B F x E x Kst x F, bias F x 1 (one bias, added by
    both filters). Gate sub-networks are created in addGSO under the reference's names
    (GFL_in/MLP_in/GFL_forget/MLP_forget/GFL_out/MLP_out, GRNN_node_*/GFL_node_*,
    input_attention/forget_attention) so state_dicts are interchangeable. As in the reference every
    gate is computed from (x_t, h0) -- the initial state -- so all gates and A(S)x_t are evaluated for
    the whole sequence in one batched pass before the sequential state recurrence.
    forward(X: B x T x G x N, h0: B x F x N) -> H: B x T x F x N.
    """

    def __init__(self, G, F, Kin, Kst, sigma=nn.Tanh, time_gating=True, spatial_gating=None, E=1, bias=True):
        super().__init__()
        self.G, self.F, self.Kin, self.Kst, self.E = G, F, Kin, Kst, E
        self.S = None
        self.graph = None
        self.weight_A = nn.parameter.Parameter(torch.empty(F, E, Kin, G))
        self.weight_B = nn.parameter.Parameter(torch.empty(F, E, Kst, F))
        self.sigma = sigma
        self.time_gating = time_gating
        self.spatial_gating = spatial_gating
        self.bias_flag = bias
        if bias:
            self.bias = nn.parameter.Parameter(torch.empty(F, 1))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.G * self.Kin)        # all three with the INPUT filter's stdv (graphML.py:2229-2235)
        self.weight_A.data.uniform_(-stdv, stdv)
        self.weight_B.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)
        ops.parameters_changed()                        # (.data writes do not move the version counters the packed-parameter cache keys on)

    def _sub_cell(self):
        c = GGCRNNCell(self.G, self.F, self.Kin, self.Kst, self.sigma, time_gating=False, E=self.E, bias=self.bias_flag)
        c.addGSO(self.graph)
        return c

    def addGSO(self, S):
        """Store the GSO (dense E x N x N tensor or GraphOperator) and (re-)create the gate sub-networks,
        in the reference's construction order so that seeded initialisation matches (graphML.py:2237-2334)."""
        if isinstance(S, torch.Tensor):
            assert len(S.shape) == 3
            assert S.shape[0] == self.E
            assert S.shape[2] == S.shape[1]
        graph = as_operator(S)
        assert graph.E == self.E
        self.N = graph.N
        self.S = S
        self.graph = graph
        self.__dict__.pop('_pc', None)                     # (the padded-state shadow of _state_padded belongs to the old graph)
        if self.time_gating == True:  # noqa: E712  (the reference compares with ==)
            dimInputMLP = self.N * self.F
            self.GFL_in = self._sub_cell()
            self.MLP_in = nn.Sequential(nn.Linear(dimInputMLP, 1, bias=self.bias_flag), nn.Sigmoid())
            self.GFL_forget = self._sub_cell()
            self.MLP_forget = nn.Sequential(nn.Linear(dimInputMLP, 1, bias=self.bias_flag), nn.Sigmoid())
            self.GFL_out = self._sub_cell()      # built and saved but never used (graphML.py:2280-2290)
            self.MLP_out = nn.Sequential(nn.Linear(dimInputMLP, 1, bias=self.bias_flag), nn.Sigmoid())
        if self.spatial_gating is not None:
            if self.spatial_gating == 'node':
                self.GRNN_node_in = self._sub_cell()
                gni = GraphFilter(self.F, 1, self.Kst, self.E, self.bias_flag)
                gni.addGSO(self.graph)
                self.GFL_node_in = nn.Sequential(gni, nn.Sigmoid())
                self.GRNN_node_forget = self._sub_cell()
                gnf = GraphFilter(self.F, 1, self.Kst, self.E, self.bias_flag)
                gnf.addGSO(self.graph)
                self.GFL_node_forget = nn.Sequential(gnf, nn.Sigmoid())
            elif self.spatial_gating == 'edge':
                self.input_attention = GraphAttentional(self.F, self.F, 1)
                self.input_attention.addGSO(self.graph)
                self.forget_attention = GraphAttentional(self.F, self.F, 1)
                self.forget_attention.addGSO(self.graph)

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        if self.graph is not None:
            self.graph = self.graph.to(self.weight_A.device)
        return self

    def _wants_grad(self, X, h0):
        """Gradient wanted for anything the forward touches: the inputs or ANY parameter of the cell incl. its gate
        sub-networks (a frozen weight_A with trainable weight_B / gates must still record an autograd graph)."""
        return torch.is_grad_enabled() and (X.requires_grad or h0.requires_grad or
                                            any(p.requires_grad for p in self.parameters()))

    # -- node-major building blocks ---------------------------------------------------------------
    def _gate_state(self, Xn, h0n):
        """sigma(A(S)x_t + b + B(S)h0 + b) for all t of an un-gated sub-cell: [T][N][B][F] (graphML.py:2362)."""
        ya = ops.lsigf_node_major(Xn, self.weight_A, self.bias, self.graph, 1.0)
        yb = ops.lsigf_node_major(h0n, self.weight_B, self.bias, self.graph, 1.0)
        return self.sigma(ya + yb)

    @staticmethod
    def _time_gate(sub, mlp, Xn, h0n):
        """sigmoid(Linear(vec_{F,N}(c_t)))  ->  [T][1][B][1]  (graphML.py:2364-2366)."""
        c = sub._gate_state(Xn, h0n)                                   # T x N x B x F
        lin = mlp[0]
        T, N, B, F = c.shape
        wnf = lin.weight.view(F, N).t().contiguous()                    # row-major vec over (f, n) -> [N][F]
        g = torch.einsum('tnbf,nf->tb', c, wnf)
        if lin.bias is not None:
            g = g + lin.bias
        return torch.sigmoid(g).view(T, 1, B, 1)

    @staticmethod
    def _node_gate(sub, gfl, Xn, h0n):
        """sigmoid(GraphFilter_{F->1}(d_t))  ->  [T][N][B][1]  (graphML.py:2383-2389)."""
        d = sub._gate_state(Xn, h0n)
        return torch.sigmoid(gfl[0].forward_node_major(d))

    # -- state widths between the fused kernels' (F = 20 of the reference drivers, kStepPredGRNNs.py:220-222) -------------------
    def _state_padded(self, X, h0):
        """The fused kernels are built for F = 32 / 64 state features. A cell with fewer (un-gated or time-gated, tanh) runs on them
        as the SAME cell with zero-padded state channels: padded rows of the taps / bias / read-out weights are zero, so a padded
        channel computes tanh(0 + 2 * 0) = 0 at every step and feeds nothing back -- results are those of the F-feature cell. Returns
        (shadow cell with F padded, padded h0) or None when the cell does not need / cannot take that route."""
        Fn = nn.functional
        if self.graph is None or self.F in (32, 64) or self.F > 64 or self.spatial_gating not in (None, 'node') or self.E != 1:
            return None
        if self.sigma not in (torch.tanh, Fn.tanh) or X.dtype not in (torch.bfloat16, torch.float32) or h0.dtype != X.dtype:
            return None
        Fp = 32 if self.F <= 32 else 64
        if self.spatial_gating == 'node':
            # node gates (bf16 only: fp32 node-gated cells are the composed path's): the gate cells and their F -> 1 filters pad the same way --
            # a padded gate-cell channel is tanh(0) = 0 under zero filter taps
            if X.dtype != torch.bfloat16 or not ops.fused_node_supported(self.graph, self.N, Fp, self.G, self.Kin, self.Kst, X.dtype, self.E):
                return None
        elif X.dtype == torch.float32:
            if ops.small_supported(self.N, self.graph.fwd[0].nnz, self.G, self.F, self.Kin, self.Kst, X.dtype, self.E) or \
                    self._wants_grad(X, h0) or self.time_gating == True or \
                    not ops.fused_x3_supported(self.graph, self.N, Fp, self.G, self.Kin, self.Kst, X.dtype, self.E):  # noqa: E712
                return None
        elif not ops.fused_supported(self.N, Fp, self.G, self.Kin, self.Kst, X.dtype, self.E):
            return None
        pc = self.__dict__.get('_pc')
        if pc is None or pc.F != Fp or pc.graph is not self.graph or pc.weight_A.dtype != self.weight_A.dtype or \
                pc.weight_A.device != self.weight_A.device:
            with torch.random.fork_rng(devices=[]):          # the shadow's own initialisation must not move the caller's generator
                pc = GGCRNNCell(self.G, Fp, self.Kin, self.Kst, self.sigma, self.time_gating, self.spatial_gating, self.E, self.bias_flag)
                pc.addGSO(self.graph)
            pc = pc.to(device=self.weight_A.device, dtype=self.weight_A.dtype)
            for q in pc.parameters():
                q.requires_grad_(False)
            object.__setattr__(self, '_pc', pc)               # not a sub-module: state_dict keys stay the reference's
        pc.native_layout = bool(getattr(self, 'native_layout', False))
        return pc, Fn.pad(h0, (0, 0, 0, Fp - self.F))

    def _padded_params(self, Fp):
        """This cell's parameters zero-padded to Fp state features (differentiable: autograd drops the padding's gradient)."""
        Fn, F, d, N = nn.functional, self.F, Fp - self.F, self.N
        out = {}
        for name, p in self.named_parameters():
            if name.endswith('weight_A'):
                p = Fn.pad(p, (0, 0, 0, 0, 0, 0, 0, d))
            elif name.endswith('weight_B'):
                p = Fn.pad(p, (0, d, 0, 0, 0, 0, 0, d))
            elif name.endswith('bias') and tuple(p.shape) == (F, 1):
                p = Fn.pad(p, (0, 0, 0, d))
            elif name.endswith('.0.weight') and tuple(p.shape) == (1, F * N):       # gate read-out Linear(N F -> 1), vec over (f, n)
                p = Fn.pad(p.view(1, F, N), (0, 0, 0, d)).reshape(1, Fp * N)
            elif name.endswith('.0.weight') and p.dim() == 4 and p.shape[0] == 1 and p.shape[3] == F:      # node gate's GraphFilter F -> 1: 1 x E x K x F
                p = Fn.pad(p, (0, d))
            out[name] = p
        return out

    def forward(self, X, h0, last_only=False):
        """last_only (an extension the classification model uses in inference): return B x 1 x F x N, the last state only --
        the fused kernels then skip the user-layout store of every other step; the other paths slice."""
        padded = self._state_padded(X, h0)
        if padded is not None:
            from torch.func import functional_call
            pc, h0p = padded
            Hp = functional_call(pc, self._padded_params(pc.F), (X, h0p), {'last_only': last_only})
            H = Hp[:, :, :self.F]
            return H if pc.native_layout else H.contiguous()
        if last_only:
            if (not torch.is_grad_enabled()) and self._use_fused(X, h0):
                return self._forward_fused(X, h0, last_only=True)
            if (not torch.is_grad_enabled()) and self._use_fused_x3(X, h0):
                return ops.fused_cell_forward_x3(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, last_only=True)
            if (not torch.is_grad_enabled()) and self._use_fused_x3(X, h0, time_gated=True):
                return ops.fused_cell_forward_x3_gated(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, self._fused_gates(), last_only=True)
            if (not torch.is_grad_enabled()) and self._use_fused_node(X, h0):
                return self._forward_fused_node(X, h0, last_only=True)
            if (not torch.is_grad_enabled()) and self._use_fused_x3_node(X, h0):
                return ops.fused_node_cell_forward_x3(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, self._node_gate_params(),
                                                      self._fused_gates() if self.time_gating == True else None, last_only=True)  # noqa: E712
            if (not torch.is_grad_enabled()) and self._use_fused_edge(X, h0):
                return self._forward_fused_edge(X, h0, last_only=True)
            return self.forward(X, h0)[:, -1:]
        assert h0.shape[0] == X.shape[0]
        ops.require_device(X, h0, self.weight_A)
        B, T, F_in, N = X.shape
        assert F_in == self.G and N == self.N
        if self._use_fused_training(X, h0):
            Xp, wA = ops.fused_pad_operands(X, self.weight_A)          # G < 32 (the drivers' G = 1): zero-padded channels
            tgates = self._fused_gates() if self.time_gating == True else None  # noqa: E712
            if self.spatial_gating == 'node':
                return ops.fused_node_cell_train(Xp, h0, wA, self.weight_B, self.bias, self.graph, self._node_gate_params(pad=True), tgates)
            if self.spatial_gating == 'edge':
                return ops.fused_edge_cell_train(Xp, h0, wA, self.weight_B, self.bias, self.graph,
                                                 (self.input_attention.mixer, self.input_attention.weight),
                                                 (self.forget_attention.mixer, self.forget_attention.weight), tgates)
            return ops.fused_cell_train(Xp, h0, wA, self.weight_B, self.bias, self.graph, tgates)
        if self._use_fused(X, h0):
            return self._forward_fused(X, h0)
        if self._use_fused_x3_training(X, h0):
            return ops.fused_cell_train_x3(X, h0, self.weight_A, self.weight_B, self.bias, self.graph)
        if self._use_fused_x3_training(X, h0, time_gated=True):
            return ops.fused_cell_train_x3_gated(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, self._fused_gates())
        if self._use_fused_node(X, h0):
            return self._forward_fused_node(X, h0)
        if self._use_fused_edge(X, h0):
            return self._forward_fused_edge(X, h0)
        if self._use_fused_x3(X, h0):
            return ops.fused_cell_forward_x3(X, h0, self.weight_A, self.weight_B, self.bias, self.graph)
        if self._use_fused_x3(X, h0, time_gated=True):
            return ops.fused_cell_forward_x3_gated(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, self._fused_gates())
        if self._use_fused_x3_node(X, h0):
            return ops.fused_node_cell_forward_x3(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, self._node_gate_params(),
                                                  self._fused_gates() if self.time_gating == True else None)  # noqa: E712
        if self._use_fused_x3_edge(X, h0):
            return ops.fused_edge_cell_forward_x3(X, h0, self.weight_A, self.weight_B, self.bias, self.graph,
                                                  self.input_attention.forward_node_major, self.forget_attention.forward_node_major,
                                                  self._fused_gates() if self.time_gating == True else None)  # noqa: E712
        if self._use_small(X, h0):
            return self._forward_small(X, h0)
        if self._use_small_training(X, h0):
            return self._forward_small(X, h0, train=True)
        if self._use_horner(X, h0):
            return self._forward_horner(X, h0)
        if self.weight_A.dtype == torch.bfloat16:
            # bf16 parameters in a variant / shape without bf16 kernels: the composed path has fp32 and fp64 kernels only, so it runs on
            # fp32 views of the parameters (differentiable casts: gradients reach the bf16 parameters) and returns bf16 states
            from torch.func import functional_call
            p32 = {k: v.float() for k, v in self.named_parameters()}
            return functional_call(self, p32, (X.float(), h0.float())).to(torch.bfloat16)
        if X.dtype != self.weight_A.dtype:      # e.g. bf16 batches meeting fp32 master weights in a variant without fused
            X, h0 = X.to(self.weight_A.dtype), h0.to(self.weight_A.dtype)      # kernels: the composed path runs in the parameters' dtype
        Xn = ops.pack_node_major(X)                                     # T x N x B x G
        h0n = ops.pack_node_major(h0.reshape(B, 1, self.F, N))          # 1 x N x B x F
        ya = ops.lsigf_node_major(Xn, self.weight_A, self.bias, self.graph, 1.0)      # all t at once
        gi = gf = None
        if self.time_gating == True:  # noqa: E712
            gi = self._time_gate(self.GFL_in, self.MLP_in, Xn, h0n)
            gf = self._time_gate(self.GFL_forget, self.MLP_forget, Xn, h0n)
        if self.spatial_gating == 'node':
            ni = self._node_gate(self.GRNN_node_in, self.GFL_node_in, Xn, h0n)
            nf = self._node_gate(self.GRNN_node_forget, self.GFL_node_forget, Xn, h0n)
            ya = ni * ya
        elif self.spatial_gating == 'edge':
            ya = self.input_attention.forward_node_major(ya)
        if gi is not None:
            ya = gi * ya
        h = h0n
        Hs = []
        for t in range(T):
            yb = ops.lsigf_node_major(h, self.weight_B, self.bias, self.graph, 1.0)   # 1 x N x B x F
            if self.spatial_gating == 'node':
                yb = nf[t:t + 1] * yb
            elif self.spatial_gating == 'edge':
                yb = self.forget_attention.forward_node_major(yb)
            if gf is not None:
                yb = gf[t:t + 1] * yb
            h = self.sigma(ya[t:t + 1] + yb)
            Hs.append(h)
        Hn = torch.cat(Hs, dim=0)                                       # T x N x B x F
        return ops.unpack_node_major(Hn)

    # -- streaming inference in Horner form (any size, fp32 / fp64, un-gated / time-gated) ------------------
    def _use_horner(self, X, h0):
        if self._wants_grad(X, h0):
            return False            # BPTT runs on the LSIGF autograd nodes
        if X.dtype == torch.bfloat16 and (self.time_gating == True or self.F % 8):  # noqa: E712  (gates are fp32 / fp64 code; 16-byte bf16 rows)
            return False
        return self.spatial_gating is None and self.E == 1 and \
            X.dtype in (torch.float32, torch.float64, torch.bfloat16) and self.weight_A.dtype == X.dtype and h0.dtype == X.dtype

    def _forward_horner(self, X, h0):
        """Taps and shifts act on different axes, so  pre_t = sum_k P^k (gi x_t A_k^T + gf h_{t-1} B_k^T) + (gi + gf) b  is
        evaluated as  acc <- P acc + u_k  (k = K-1 .. 0): K-1 hops over F channels per step, x and h sharing them,
        instead of the reference order's 2 (K-1) hops over G + F channels (graphML.py:118-135 applied twice per step).
        Every hop is one accumulate-SpMM pass over the [N][B F] state (ops.spmm_raw -> gcrnn_spmm_ex); the last hop's
        epilogue adds the bias and applies tanh, writing h_t in place. Taps: one matrix-core pass for all K taps of a step
        in bf16 (gcrnn_taps_bf16_forward), the LDS-tiled tap kernel (gcrnn_taps_forward) in fp32 / fp64 -- no library GEMM."""
        B, T, G, N = X.shape
        F, Kin, Kst = self.F, self.Kin, self.Kst
        K = max(Kin, Kst)
        Xn = ops.pack_node_major(X)                                     # T x N x B x G
        h = ops.pack_node_major(h0.reshape(B, 1, F, N))                 # 1 x N x B x F
        gi = gf = None
        if self.time_gating == True:  # noqa: E712
            gi = self._time_gate(self.GFL_in, self.MLP_in, Xn, h)       # T x 1 x B x 1
            gf = self._time_gate(self.GFL_forget, self.MLP_forget, Xn, h)
        csr = self.graph.fwd[0]
        Hn = torch.empty((T, N, B, F), dtype=X.dtype, device=X.device)
        tanh_fused = self.sigma in (torch.tanh, nn.functional.tanh) and gi is None
        bvec = self.bias.detach().reshape(-1) if self.bias is not None else None
        if X.dtype == torch.bfloat16:
            bvec = bvec.float() if bvec is not None else None           # the kernels take fp32 bias / CSR weights with bf16 rows
            Gp = 32 if G <= 32 else 64
            wA = self.weight_A
            mfma = G <= 64 and ops.taps_bf16_supported(F, Gp, K)
            if mfma and Gp != G:                                        # the matrix-core taps consume whole 32-feature steps
                Xp = Xn.new_zeros((T, N, B, Gp))
                Xp[..., :G] = Xn
                Xn, wA = Xp, nn.functional.pad(wA, (0, Gp - G))
            for t in range(T):
                if mfma:
                    _, rest = ops.taps_bf16(h, Xn[t:t + 1], wA, self.weight_B, out0=Hn[t:t + 1])     # u_0 -> Hn[t], u_1.. -> rest
                else:
                    # shapes outside the matrix-core tap kernel (F not 32 / 64): fp32 taps on the LDS-tiled kernel, rounded to bf16 rows
                    hf, xf = h.float(), Xn[t:t + 1].float()
                    rest = torch.empty((max(K - 1, 1), 1, N, B, F), dtype=X.dtype, device=X.device)
                    for k in range(K):
                        u = ops.taps_rows(hf, self.weight_B[:, 0, k].float()) if k < Kst else None
                        if k < Kin:
                            u = ops.taps_rows(xf, self.weight_A[:, 0, k].float(), out=u, accumulate=u is not None)
                        (Hn[t:t + 1] if k == 0 else rest[k - 1]).copy_(u)
                acc = rest[K - 2] if K > 1 else None
                for k in range(K - 2, -1, -1):
                    dst = rest[k - 1] if k > 0 else Hn[t:t + 1]
                    ops.spmm_raw(csr, acc, out=dst, accumulate=True, bias=bvec, bias_scale=2.0, tanh=(k == 0 and tanh_fused))
                    acc = dst
                if K == 1 or not tanh_fused:
                    pre = Hn[t:t + 1].float() + (2.0 * bvec.view(1, 1, 1, F) if bvec is not None else 0.0)
                    Hn[t:t + 1] = self.sigma(pre).to(X.dtype)
                h = Hn[t:t + 1]
            return ops.unpack_node_major(Hn)
        if gi is None and ops.taps_mfma_supported(X.dtype, F, F, G):
            # every tap of a step in one launch on the fp32 / fp64 matrix cores, then the same hop chain as the bf16 branch
            for t in range(T):
                _, rest = ops.taps_mfma(h, Xn[t:t + 1], self.weight_A, self.weight_B, out0=Hn[t:t + 1])
                acc = rest[K - 2] if K > 1 else None
                for k in range(K - 2, -1, -1):
                    dst = rest[k - 1] if k > 0 else Hn[t:t + 1]
                    ops.spmm_raw(csr, acc, out=dst, accumulate=True, bias=bvec, bias_scale=2.0, tanh=(k == 0 and tanh_fused))
                    acc = dst
                if K == 1 or not tanh_fused:
                    pre = Hn[t:t + 1] + (2.0 * bvec.view(1, 1, 1, F) if bvec is not None else 0.0)
                    Hn[t:t + 1] = self.sigma(pre)
                h = Hn[t:t + 1]
            return ops.unpack_node_major(Hn)
        wA = [self.weight_A[:, 0, k].contiguous() for k in range(Kin)]  # F x G each
        wB = [self.weight_B[:, 0, k].contiguous() for k in range(Kst)]  # F x F each
        for t in range(T):
            acc = None
            for k in range(K - 1, -1, -1):
                dst = Hn[t:t + 1] if k == 0 else torch.empty((1, N, B, F), dtype=X.dtype, device=X.device)
                first = True
                if k < Kst:
                    ops.taps_rows(h, wB[k], out=dst)
                    if gf is not None:
                        dst.mul_(gf[t:t + 1])
                    first = False
                if k < Kin:
                    if gi is None:
                        ops.taps_rows(Xn[t:t + 1], wA[k], out=dst, accumulate=not first)
                    else:
                        u = ops.taps_rows(Xn[t:t + 1], wA[k]).mul_(gi[t:t + 1])
                        dst.copy_(u) if first else dst.add_(u)
                if acc is not None:
                    ops.spmm_raw(csr, acc, out=dst, accumulate=True, bias=bvec, bias_scale=2.0, tanh=(k == 0 and tanh_fused))
                acc = dst
            if K == 1 or not tanh_fused:
                pre = acc
                if bvec is not None:
                    bb = bvec.view(1, 1, 1, F)
                    pre = pre + (2.0 * bb if gi is None else (gi[t:t + 1] + gf[t:t + 1]) * bb)
                Hn[t:t + 1] = self.sigma(pre)
            h = Hn[t:t + 1]
        return ops.unpack_node_major(Hn)

    # -- small-graph persistent path (fp32 / fp64, un-gated / time-gated, sigma = tanh, inference) -------
    def _small_node_gating_ok(self, X, backward):
        """Per-node gates run on the matrix-core family only (dense GSO in LDS)."""
        return self.spatial_gating == 'node' and \
            ops.small_dense_supported(self.N, self.G, self.F, self.Kin, self.Kst, X.dtype, backward=backward, gated=True)

    def _use_small(self, X, h0):
        if self._wants_grad(X, h0):
            return False
        if self.sigma not in (torch.tanh, nn.functional.tanh):
            return False
        if self.spatial_gating is not None and not self._small_node_gating_ok(X, False):
            return False
        return self.weight_A.dtype == X.dtype and h0.dtype == X.dtype and \
            ops.small_supported(self.N, self.graph.fwd[0].nnz, self.G, self.F, self.Kin, self.Kst, X.dtype, self.E)

    def _use_small_training(self, X, h0):
        """Small graphs, gradients wanted for parameters / h0 but not for X: forward and BPTT are one launch each."""
        if not torch.is_grad_enabled() or X.requires_grad:
            return False
        if self.sigma not in (torch.tanh, nn.functional.tanh):
            return False
        if self.spatial_gating is not None and not self._small_node_gating_ok(X, True):
            return False
        return self.weight_A.dtype == X.dtype and h0.dtype == X.dtype and \
            ops.small_training_supported(self.N, self.graph.fwd[0].nnz, self.G, self.F, self.Kin, self.Kst, X.dtype, self.E)

    def _forward_small(self, X, h0, train=False):
        gi = gf = None
        if self.time_gating == True and ops.small_gates_supported(  # noqa: E712
                self.N, self.G, self.F, self.Kin, self.Kst, X.dtype, backward=train):
            # both gates of every step in one launch on the matrix cores (parameters stacked input | forget)
            gin, gfo = self.GFL_in, self.GFL_forget
            lin_i, lin_f = self.MLP_in[0], self.MLP_forget[0]
            gates = ops.small_time_gates(
                X, h0, torch.stack((gin.weight_A[:, 0], gfo.weight_A[:, 0])), torch.stack((gin.weight_B[:, 0], gfo.weight_B[:, 0])),
                torch.stack((gin.bias.view(-1), gfo.bias.view(-1))) if gin.bias is not None else None,
                torch.stack((lin_i.weight.view(-1), lin_f.weight.view(-1))),
                torch.stack((lin_i.bias.view(()), lin_f.bias.view(()))) if lin_i.bias is not None else None, self.graph)
            gi, gf = gates[0], gates[1]
        elif self.time_gating == True:  # noqa: E712   gates read (x_t, h0) only: one batched pass over all t
            B = X.shape[0]
            Xn = ops.pack_node_major(X)
            h0n = ops.pack_node_major(h0.reshape(B, 1, self.F, self.N))
            gi = self._time_gate(self.GFL_in, self.MLP_in, Xn, h0n).reshape(X.shape[1], B)
            gf = self._time_gate(self.GFL_forget, self.MLP_forget, Xn, h0n).reshape(X.shape[1], B)
        if self.spatial_gating == 'node':
            # node gates (graphML.py:2379-2399) from one batched pass over all t; the recurrence takes them per node,
            # multiplied by the time gates when both are on: gates [B][T][N]
            B = X.shape[0]
            Xn = ops.pack_node_major(X)
            h0n = ops.pack_node_major(h0.reshape(B, 1, self.F, self.N))
            ni = self._node_gate(self.GRNN_node_in, self.GFL_node_in, Xn, h0n).squeeze(3).permute(2, 0, 1)      # B x T x N
            nf = self._node_gate(self.GRNN_node_forget, self.GFL_node_forget, Xn, h0n).squeeze(3).permute(2, 0, 1)
            gi = ni if gi is None else ni * gi.t().unsqueeze(2)
            gf = nf if gf is None else nf * gf.t().unsqueeze(2)
        if train:
            return ops.small_cell_train(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, gi, gf)
        return ops.small_cell_forward(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, gi, gf)

    # -- fused flagship path (bf16, un-gated / time-gated, sigma = tanh; inference here, training via ops.fused_cell_train) ----
    def _use_fused(self, X, h0):
        if self._wants_grad(X, h0):
            return False            # BPTT runs on the fused training path or the composed path
        if self.spatial_gating is not None or self.sigma not in (torch.tanh, nn.functional.tanh):
            return False
        return ops.fused_supported(self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E) and \
            self.weight_A.dtype in (X.dtype, torch.float32) and h0.dtype == X.dtype

    def _use_fused_node(self, X, h0):
        """Node-gated cell (optionally time-gated too), bf16, inference: fused kernels (gates, A(S)x_t + b and both gate filters for
        all steps at once; the recurrence on the state-only operand with per-node gates in its epilogue)."""
        if self._wants_grad(X, h0) or self.spatial_gating != 'node' or self.sigma not in (torch.tanh, nn.functional.tanh):
            return False
        if self.bias is None and self.time_gating == True:  # noqa: E712
            return False
        return ops.fused_node_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E) and \
            self.weight_A.dtype in (X.dtype, torch.float32) and h0.dtype == X.dtype

    def _node_gate_params(self, pad=False):
        """pad: the gate cells' input taps zero-padded to the kernels' channel count by a differentiable pad (training)."""
        Gp = ops.fused_padded_inputs(self.F, self.G)
        padf = (lambda w: w) if (not pad or Gp == self.G) else (lambda w: nn.functional.pad(w, (0, Gp - self.G)))
        out = {}
        for name, sub, gfl in (('in', self.GRNN_node_in, self.GFL_node_in[0]), ('forget', self.GRNN_node_forget, self.GFL_node_forget[0])):
            out[name] = (padf(sub.weight_A), sub.weight_B, sub.bias, gfl.weight, gfl.bias)
        return out

    def _forward_fused_node(self, X, h0, last_only=False):
        tg = self._fused_gates() if self.time_gating == True else None  # noqa: E712
        return ops.fused_node_cell_forward(X, h0, self.weight_A, self.weight_B, self.bias, self.graph, self._node_gate_params(),
                                           time_gates=tg, last_only=last_only)

    def _use_fused_edge(self, X, h0):
        """Edge-gated cell (optionally time-gated too), bf16, inference: the attention's mixing matrix is folded into the filter taps,
        the x branch runs for all steps at once, every step is a filter pass plus the attention kernel."""
        if self._wants_grad(X, h0) or self.spatial_gating != 'edge' or self.sigma not in (torch.tanh, nn.functional.tanh):
            return False
        if self.bias is None and self.time_gating == True:  # noqa: E712
            return False
        att = self.input_attention
        if att.K != 1 or att.E != 1 or not att.concatenate or att.nonlinearity is not nn.functional.relu:
            return False
        return ops.fused_edge_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E) and \
            self.weight_A.dtype in (X.dtype, torch.float32) and h0.dtype == X.dtype

    def _forward_fused_edge(self, X, h0, last_only=False):
        tg = self._fused_gates() if self.time_gating == True else None  # noqa: E712
        return ops.fused_edge_cell_forward(X, h0, self.weight_A, self.weight_B, self.bias, self.graph,
                                           (self.input_attention.mixer, self.input_attention.weight),
                                           (self.forget_attention.mixer, self.forget_attention.weight), time_gates=tg, last_only=last_only)

    def _use_fused_x3(self, X, h0, time_gated=False):
        """fp32 inference of the un-gated cell on the fp32-accurate fused kernels (three bf16 planes per operand): graphs that fit
        the fused kernels, too large for the one-launch small-graph kernels, with one weight on every edge. time_gated=True: the same
        question for the time-gated cell (ops.fused_cell_forward_x3_gated: gates and scaled steps composed from the same kernel)."""
        if self._wants_grad(X, h0):
            return False
        if (self.time_gating == True) != bool(time_gated) or self.spatial_gating is not None or self.sigma not in (torch.tanh, nn.functional.tanh):  # noqa: E712
            return False
        if time_gated and (X.shape[0] > 2048 or os.environ.get('GCRNN_NO_X3_GATED')):
            return False
        if X.dtype != torch.float32 or h0.dtype != X.dtype or self.weight_A.dtype != X.dtype:
            return False
        if ops.small_supported(self.N, self.graph.fwd[0].nnz, self.G, self.F, self.Kin, self.Kst, X.dtype, self.E):
            return False
        return ops.fused_x3_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E, X.shape[0], X.shape[1])

    def _use_fused_x3_node(self, X, h0):
        """fp32 inference of the NODE-gated cell (with or without time gates) on the fp32-accurate fused kernels (round 5,
        ops.fused_node_cell_forward_x3): the x3 conditions with G == F."""
        if self._wants_grad(X, h0) or self.spatial_gating != 'node' or self.sigma not in (torch.tanh, nn.functional.tanh):
            return False
        if X.dtype != torch.float32 or h0.dtype != X.dtype or self.weight_A.dtype != X.dtype or X.shape[0] > 2048:
            return False
        if ops.small_supported(self.N, self.graph.fwd[0].nnz, self.G, self.F, self.Kin, self.Kst, X.dtype, self.E):
            return False
        return ops.fused_node_x3_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E, X.shape[0], X.shape[1])

    def _use_fused_x3_edge(self, X, h0):
        """fp32 inference of the EDGE-gated cell (with or without time gates) with both filters on the fp32-accurate fused kernels (round 5,
        ops.fused_edge_cell_forward_x3; the attentions run on the fp32 CSR edge-softmax kernels): the x3 conditions with G == F."""
        if self._wants_grad(X, h0) or self.spatial_gating != 'edge' or self.sigma not in (torch.tanh, nn.functional.tanh):
            return False
        if X.dtype != torch.float32 or h0.dtype != X.dtype or self.weight_A.dtype != X.dtype or X.shape[0] > 2048 or os.environ.get('GCRNN_NO_X3_EDGE'):
            return False
        if ops.small_supported(self.N, self.graph.fwd[0].nnz, self.G, self.F, self.Kin, self.Kst, X.dtype, self.E):
            return False
        return self.G == self.F and ops.fused_x3_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E, X.shape[0], X.shape[1])

    def _use_fused_x3_training(self, X, h0, time_gated=False):
        """fp32 training of the un-gated cell at the north_star's tolerance on the fused kernels (x3 forward, x3 data chain, exact-fp32
        weight gradient): fp32 tensors and parameters, uniform-weight graph (forward and adjoint), gradients for the parameters and h0.
        time_gated=True (r4): the same question for the time-gated cell (ops.fused_cell_train_x3_gated: gradients for every parameter,
        the gates' sub-cells and read-outs included; none for h0)."""
        if not torch.is_grad_enabled() or X.requires_grad:
            return False
        if not (h0.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        if (self.time_gating == True) != bool(time_gated) or self.spatial_gating is not None or self.sigma not in (torch.tanh, nn.functional.tanh):  # noqa: E712
            return False
        if X.dtype != torch.float32 or h0.dtype != X.dtype or self.weight_A.dtype != X.dtype or os.environ.get('GCRNN_NO_X3_TRAINING'):
            return False
        if ops.small_supported(self.N, self.graph.fwd[0].nnz, self.G, self.F, self.Kin, self.Kst, X.dtype, self.E):
            return False
        if time_gated:
            if h0.requires_grad or self.GFL_in.weight_A.dtype != X.dtype or os.environ.get('GCRNN_NO_X3_GATED'):
                return False
            return ops.fused_x3_time_training_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E, X.shape[0], X.shape[1])
        return ops.fused_x3_training_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E, X.shape[0], X.shape[1])

    def _use_fused_training(self, X, h0):
        """bf16 activations (parameters bf16 or fp32 master weights), plain or time-gated cell, gradients wanted for the
        parameters (and, for the plain cell, optionally h0; X too when G == F): forward and BPTT on the fused kernels."""
        if not torch.is_grad_enabled():
            return False
        if X.requires_grad and (self.spatial_gating is not None or not ops.fused_input_grad_ok(self.F, self.G)):
            return False                                   # dX: un-gated and (r4) time-gated cells with G == F (the input filters' adjoint passes, the gate cells' included)
        if not (h0.requires_grad or X.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        if self.spatial_gating == 'node':
            if h0.requires_grad or self.bias is None or self.GRNN_node_in.weight_A.dtype != self.weight_A.dtype or \
                    not ops.fused_node_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, X.dtype, self.E):
                return False
        elif self.spatial_gating == 'edge':
            att = self.input_attention
            if h0.requires_grad or att.K != 1 or att.E != 1 or not att.concatenate or att.nonlinearity is not nn.functional.relu or \
                    att.weight.dtype != self.weight_A.dtype or \
                    not ops.fused_edge_training_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, self.E):
                return False
        elif self.spatial_gating is not None:
            return False
        if self.time_gating == True:  # noqa: E712   the gates' sub-cells share the cell's shapes; (r4) they hand h0 its gradient when G == F
            if (h0.requires_grad and not ops.fused_input_grad_ok(self.F, self.G)) or self.bias is None or self.GFL_in.weight_A.dtype != self.weight_A.dtype:
                return False
        if self.sigma not in (torch.tanh, nn.functional.tanh) or X.dtype != torch.bfloat16 or h0.dtype != X.dtype:
            return False
        if self.weight_A.dtype not in (torch.bfloat16, torch.float32):
            return False
        return ops.fused_training_supported(self.graph, self.N, self.F, self.G, self.Kin, self.Kst, self.E)

    def _fused_gates(self):
        Gp = ops.fused_padded_inputs(self.F, self.G)
        pad = (lambda w: w) if Gp == self.G else (lambda w: nn.functional.pad(w, (0, Gp - self.G)))
        return {'in': (pad(self.GFL_in.weight_A), self.GFL_in.weight_B, self.GFL_in.bias,
                       self.MLP_in[0].weight, self.MLP_in[0].bias),
                'forget': (pad(self.GFL_forget.weight_A), self.GFL_forget.weight_B, self.GFL_forget.bias,
                           self.MLP_forget[0].weight, self.MLP_forget[0].bias)}

    def _forward_fused(self, X, h0, last_only=False, head=None):
        gates = self._fused_gates() if self.time_gating == True else None  # noqa: E712
        # (G < 32, the drivers' G = 1: the taps are padded here, X by the pack kernel itself -- no padded copy of X)
        return ops.fused_cell_forward(X, h0, ops.fused_pad_taps(self.weight_A), self.weight_B, self.bias, self.graph, gates, last_only=last_only, head=head,
                                      native_out=bool(getattr(self, 'native_layout', False)) and head is None)

    def forward_native(self, xs, h0s=None):
        """The un-gated recurrence on the fused kernels' own layout end to end (inference): xs [T][B][NPad][G] bf16 sequence-major
        (ops.to_sequence_major, or a generator that emits it), h0s [B][NPad][F] or None (zeros) -> hs [T][B][NPad][F] bf16;
        `hs.permute(1, 0, 3, 2)[..., :N]` is H in the reference's B x T x F x N shape (graphML.py:2425-2427) without a copy.
        The module flag `native_layout = True` gives the same view from the ordinary forward(X, h0) (user-layout X in)."""
        assert self.graph is not None and self.time_gating != True and self.spatial_gating is None  # noqa: E712
        assert not torch.is_grad_enabled() or not any(p.requires_grad for p in self.parameters()), 'forward_native is an inference path'
        assert xs.dtype == torch.bfloat16 and self.weight_A.dtype == torch.bfloat16 and self.sigma in (torch.tanh, nn.functional.tanh)
        wA = self.weight_A
        if wA.shape[3] != xs.shape[3]:                   # G < 32: the padded taps of ops.fused_pad_operands
            wA = torch.nn.functional.pad(wA.detach(), (0, xs.shape[3] - wA.shape[3]))
        assert ops.fused_supported(self.N, self.F, xs.shape[3], self.Kin, self.Kst, torch.bfloat16, self.E), 'shape outside the fused kernels'
        return ops.fused_cell_forward_native(xs, h0s, wA, self.weight_B, self.bias, self.graph, self.N)

    def forward_with_head(self, X, h0, weight, bias):
        """Inference of cell + output head Linear(F -> 1) shared by all nodes (the regression model's `multipMlp` head with one
        output, reference architectures.py:1616-1627) with the head fused onto the h_t store of the fused step kernel: returns
        y: B x T x 1 x N (fp32) without ever writing H in the user layout; None when this cell / input does not run on that path."""
        if torch.is_grad_enabled() or not self._use_fused(X, h0) or self.N % 8 != 0 or os.environ.get('GCRNN_NO_FUSED_HEAD'):      # env: A/B switch
            return None
        return self._forward_fused(X, h0, head=(weight, bias))

    def extra_repr(self):
        return 'in_features=%d, state_features=%d, taps=(%d,%d), time_gating=%s, spatial_gating=%s, %s' % (
            self.G, self.F, self.Kin, self.Kst, self.time_gating, self.spatial_gating,
            ('GSO stored: number_nodes=%d' % self.N) if self.graph is not None else 'no GSO stored')
