"""MI355X-native gated GCRNN recurrence (drop-in for the hot path of luanaruiz9/gated_gcrnns).

    import gated_gcrnns_amd.Utils.graphML as gml            # LSIGF, GraphFilter, GGCRNNCell, GraphAttentional
    import gated_gcrnns_amd.Modules.architectures as archit  # GatedGCRNNforRegression / forClassification

All compute runs in hand-written HIP kernels for gfx950 behind the C ABI of
include/gcrnn.h (libgcrnn_hip.so, loaded with ctypes). There is no CPU path.
"""
from . import _lib                      # noqa: F401  (fails loudly if the library is not built)
from .graph import GraphOperator        # noqa: F401

__version__ = '0.1.0'
