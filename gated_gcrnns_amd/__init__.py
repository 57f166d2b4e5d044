"""MI355X-native gated GCRNN recurrence (drop-in for the hot path of luanaruiz9/gated_gcrnns).

    import gated_gcrnns_amd.Utils.graphML as gml            # LSIGF, GraphFilter, GGCRNNCell, GraphAttentional
    import gated_gcrnns_amd.Modules.architectures as archit  # GatedGCRNNforRegression / forClassification

The hot path -- graph shifts, filter taps, the fused / small-graph / streaming recurrences with their BPTT, the
per-node head, the edge softmax, the L1 loss, the MSE metric, Adam -- runs in hand-written HIP kernels for gfx950 behind
the C ABI of include/gcrnn.h (libgcrnn_hip.so, loaded with ctypes). There is no CPU path. What is NOT ours, stated
plainly: dense layers run on the ROCm BLAS libraries through torch (the `oneMlp` / classification heads
`nn.Linear(N*F, .)`, `ops.row_linear` = the attention projections W u and [a1 a2] Wx of the edge gate and per-node MLPs
deeper than one layer), and the composed path glues kernels with torch elementwise / reduction ops (gate products, the
time gate's read-out reduction, sigmoid / tanh).
"""
from . import _lib                      # noqa: F401  (fails loudly if the library is not built)
from .graph import GraphOperator        # noqa: F401

__version__ = '0.1.0'
