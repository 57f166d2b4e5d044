"""ctypes binding of libgcrnn_hip.so (the C ABI declared in include/gcrnn.h).

There is no CPU fallback: if the library is missing, import fails loudly with
the build command. Entry points return an int status; `check` raises.
"""
import ctypes as C
import os

# torch FIRST: it ships its own libamdhip64.so.7. libgcrnn_hip.so must bind to that same runtime instance
# (device pointers and streams are only meaningful inside one HIP runtime); loading ours first would pull
# the system /opt/rocm runtime into the process beside torch's and launches would see "no device".
import torch  # noqa: F401

from . import build as _build

F32, F64, BF16 = 0, 1, 2

_c_i64 = C.c_int64
_c_p = C.c_void_p


class GcrnnError(RuntimeError):
    pass


def _load():
    path = os.environ.get('GCRNN_LIBPATH')      # A/B builds for profiling (tools/ablate.sh); default: the in-tree library
    if path:
        return _bind(C.CDLL(path))
    path = _build.LIBPATH
    if _build.needs_build() and os.path.exists(_build.HIPCC):
        _build.build(verbose=False)          # in-tree, gfx950; sources newer than the .so (or no .so yet)
    if not os.path.exists(path):
        raise ImportError(
            'gated_gcrnns_amd: %s is missing. Build it with `python -m gated_gcrnns_amd.build` '
            '(needs hipcc; there is no CPU fallback).' % path)
    return _bind(C.CDLL(path))


def _bind(lib):
    sig = {
        'gcrnn_version': (C.c_int, []),
        'gcrnn_status_string': (C.c_char_p, [C.c_int]),
        'gcrnn_last_hip_error': (C.c_char_p, []),
        'gcrnn_csr_count': (C.c_int, [_c_p, _c_i64, C.c_int, C.c_int, C.c_double, C.POINTER(_c_i64)]),
        'gcrnn_csr_fill': (C.c_int, [_c_p, _c_i64, C.c_int, C.c_int, C.c_double, _c_p, _c_p, _c_p]),
        'gcrnn_degree_order': (C.c_int, [_c_p, _c_i64, _c_p]),
        'gcrnn_pack_node_major': (C.c_int, [C.c_int, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
        'gcrnn_pack_node_major_sum_f32': (C.c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_node_gate_filter_supported': (C.c_int, [_c_i64, _c_i64, _c_i64, C.c_double]),
        'gcrnn_node_gate_filter_f32': (C.c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_i64, C.c_double, _c_p, C.c_int, _c_p]),
        'gcrnn_unpack_node_major': (C.c_int, [C.c_int, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
        'gcrnn_spmm': (C.c_int, [C.c_int, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, C.c_int, _c_p]),
        'gcrnn_spmm_ex': (C.c_int, [C.c_int, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, C.c_int, _c_p, C.c_double, _c_i64,
                                    C.c_int, C.c_int, C.c_int, C.c_int, _c_p]),
        'gcrnn_taps_mfma_supported': (C.c_int, [C.c_int, _c_i64, _c_i64, _c_i64]),
        'gcrnn_taps_mfma_forward': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_taps_bf16_supported': (C.c_int, [_c_i64, _c_i64, _c_i64]),
        'gcrnn_taps_bf16_forward': (C.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_batch_time_mse_slabs': (_c_i64, [_c_i64, _c_i64]),
        'gcrnn_batch_time_mse': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p]),
        'gcrnn_adam_flat': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_i64, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, _c_p, _c_p]),
        'gcrnn_taps_forward': (C.c_int, [C.c_int, _c_p, _c_p, _c_i64, _c_p, _c_p, C.c_double, _c_p,
                                         _c_i64, _c_i64, _c_i64, _c_i64, C.c_int, _c_p]),
        'gcrnn_taps_backward_data': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                               _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_taps_backward_weight_parts': (C.c_int, [_c_i64, _c_i64, _c_i64, _c_i64, C.POINTER(_c_i64), C.POINTER(_c_i64)]),
        'gcrnn_taps_backward_weight': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, C.c_double,
                                                 _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_ell_size': (C.c_int, [_c_p, _c_i64, _c_p, C.c_int, C.c_int, _c_i64, C.POINTER(_c_i64)]),
        'gcrnn_ell_fill': (C.c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, C.c_int, C.c_int, _c_i64, _c_p, _c_p, _c_p, _c_p]),
        'gcrnn_ell_assign_rows_z': (C.c_int, [_c_p, _c_p, _c_i64, _c_p, C.c_int, _c_i64, _c_i64, _c_p]),
        'gcrnn_ell_fill_z': (C.c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, C.c_int, C.c_int, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p]),
        'gcrnn_ell_assign_rows': (C.c_int, [_c_p, _c_p, _c_i64, _c_p, C.c_int, _c_i64, _c_p]),
        'gcrnn_fused_supported': (C.c_int, [_c_i64, _c_i64, _c_i64, _c_i64]),
        'gcrnn_fused_padded_nodes': (_c_i64, []),
        'gcrnn_fused_step_waves': (_c_i64, []),
        'gcrnn_fused_wgrad_waves': (_c_i64, []),
        'gcrnn_pack_seq_major': (C.c_int, [C.c_int, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
        'gcrnn_pack_seq_major_padded': (C.c_int, [_c_p, _c_p] + [_c_i64] * 6 + [_c_p]),
        'gcrnn_unpack_seq_major': (C.c_int, [C.c_int, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
        'gcrnn_fused_pack_weights': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_fused_forward_bf16': (C.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                               _c_p, _c_p,
                                               _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, C.c_int, _c_p, C.c_double, _c_p, _c_p, _c_p, _c_p]),
        'gcrnn_fused_pack_weights_wide': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, C.c_double, _c_p]),
        'gcrnn_fused_forward_wide_supported': (C.c_int, [_c_i64] * 7 + [C.c_double, C.c_int, C.c_int]),
        'gcrnn_fused_forward_wide_bf16': (C.c_int, [_c_p] * 10 + [_c_i64] * 7 + [_c_p, C.c_int, _c_p, _c_p, _c_p, _c_p]),
        'gcrnn_fused_backward_data_wide_supported': (C.c_int, [_c_i64] * 6 + [C.c_double, C.c_int, C.c_int]),
        'gcrnn_fused_backward_data_wide_bf16': (C.c_int, [_c_p] * 8 + [_c_i64] * 6 + [_c_p] * 7),
        'gcrnn_gate_readout_finish': (C.c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
        'gcrnn_fused_filter_output_wide_supported': (C.c_int, [_c_i64] * 7 + [C.c_double, C.c_int, C.c_int]),
        'gcrnn_fused_filter_output_wide_bf16': (C.c_int, [_c_p] * 7 + [_c_i64] * 7 + [_c_p, _c_p]),
        'gcrnn_fused_node_forward_wide_supported': (C.c_int, [_c_i64] * 6 + [C.c_double, C.c_int]),
        'gcrnn_fused_node_forward_wide_bf16': (C.c_int, [_c_p] * 11 + [_c_i64] * 6 + [_c_p, C.c_int, _c_p]),
        'gcrnn_fused_gate_pair_wide_supported': (C.c_int, [_c_i64] * 7 + [C.c_double, C.c_int, C.c_int]),
        'gcrnn_fused_gate_pair_prepass_wide_bf16': (C.c_int, [_c_p] * 12 + [_c_i64] * 7 + [_c_p, _c_p, _c_p, _c_p]),
        'gcrnn_fused_gate_pair_prepass_taps_wide_bf16': (C.c_int, [_c_p] * 7 + [_c_i64] + [_c_p] * 5 + [_c_i64] * 7 + [_c_p, _c_p, _c_p, _c_p]),
        'gcrnn_fused_inline_pack_supported': (C.c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, C.c_double]),
        'gcrnn_fused_seq_steps_per_launch': (_c_i64, [_c_i64] * 7 + [C.c_double, C.c_int, C.c_int, C.c_int]),
        'gcrnn_fused_filter_output_bf16': (C.c_int, [_c_p] * 11 + [_c_i64] * 7 + [C.c_double, C.c_int, _c_p]),
        'gcrnn_fused_node_forward_bf16': (C.c_int, [_c_p] * 15 + [_c_i64] * 6 + [_c_p, C.c_int, C.c_double, _c_p]),
        'gcrnn_fused_node_backward_data_bf16': (C.c_int, [_c_p] * 12 + [_c_i64] * 6 + [C.c_double, _c_p, C.c_int, _c_p]),
        'gcrnn_node_cell_backward': (C.c_int, [_c_p] * 11 + [_c_i64] * 5 + [_c_p]),
        'gcrnn_node_gate_dot': (C.c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_node_gate_dot_backward': (C.c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_fused_edge_attention_supported': (C.c_int, [_c_i64, _c_i64]),
        'gcrnn_fused_edge_attention_bf16': (C.c_int, [_c_p] * 13 + [_c_i64] * 5 + [C.c_double, _c_p]),
        'gcrnn_fused_edge_attention_backward_supported': (C.c_int, [_c_i64, _c_i64, _c_i64]),
        'gcrnn_fused_edge_attention_backward_bf16': (C.c_int, [_c_p] * 14 + [_c_i64] * 6 + [C.c_double, _c_p]),
        'gcrnn_fused_backward_step_bf16': (C.c_int, [_c_p] * 11 + [_c_i64] * 5 + [C.c_double, _c_p, _c_p, _c_i64, C.c_int, _c_p]),
        'gcrnn_fused_backward_seed_bf16': (C.c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p]),
        'gcrnn_fused_x3_supported': (C.c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
        'gcrnn_pack_seq_major_x3': (C.c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_fused_pack_weights_x3': (C.c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_fused_forward_x3': (C.c_int, [_c_p] * 8 + [_c_i64] * 7 + [C.c_double, _c_p, C.c_int, _c_p, _c_p]),
        'gcrnn_fused_forward_x3_scaled': (C.c_int, [_c_p] * 9 + [_c_i64] * 7 + [C.c_double, _c_p, _c_i64, _c_p, _c_p]),
        'gcrnn_fused_x3_training_supported': (C.c_int, [_c_i64] * 5),
        'gcrnn_fused_backward_data_x3': (C.c_int, [_c_p] * 8 + [_c_i64] * 6 + [C.c_double, _c_p, _c_p]),
        'gcrnn_fused_backward_weight_f32': (C.c_int, [_c_p] * 9 + [_c_i64] * 7 + [C.c_double, _c_p, _c_p]),
        'gcrnn_fused_backward_data_x3_gated': (C.c_int, [_c_p] * 8 + [_c_i64] * 6 + [C.c_double, _c_p, _c_p, _c_p, _c_p, _c_p]),
        'gcrnn_fused_filter_x3': (C.c_int, [_c_p] * 6 + [_c_i64] * 5 + [C.c_double, _c_p, _c_p]),
        'gcrnn_fused_backward_weight_f32_gated': (C.c_int, [_c_p] * 9 + [_c_i64] * 7 + [C.c_double, _c_p, _c_p, C.c_int, _c_p, _c_p]),
        'gcrnn_fused_gate_cells_x3': (C.c_int, [_c_p] * 8 + [_c_i64] * 7 + [C.c_double, _c_p, C.c_int, _c_p, _c_p]),
        'gcrnn_pack_seq_major_x3_ex': (C.c_int, [_c_p, _c_p] + [_c_i64] * 5 + [_c_p, _c_p, C.c_int, _c_i64, _c_p]),
        'gcrnn_x3_item_dots': (C.c_int, [_c_p] * 5 + [_c_i64] * 4 + [_c_p]),
        'gcrnn_x3_node_gate_step': (C.c_int, [_c_p] * 7 + [_c_i64] * 5 + [_c_p]),
        'gcrnn_pack_seq_major_steps': (C.c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_fused_gate_prepass_bf16': (C.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                                    _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, C.c_double, C.c_int, _c_p]),
        'gcrnn_fused_gate_prepass_pack_bf16': (C.c_int, [_c_p] * 14 + [_c_i64] * 7 + [_c_p, C.c_double, C.c_int, _c_p]),
        'gcrnn_fused_gate_prepass_lays_out': (_c_i64, [_c_i64] * 7 + [C.c_double, C.c_int]),
        'gcrnn_fused_gate_prepass_taps_bf16': (C.c_int, [_c_p] * 7 + [_c_i64] + [_c_p] * 7 + [_c_i64] * 7 + [_c_p, C.c_double, C.c_int, _c_p]),
        'gcrnn_fused_gate_prepass_taps_supported': (C.c_int, [_c_i64] * 7 + [C.c_double, C.c_int, C.c_int, _c_i64]),
        'gcrnn_fused_gate_readout_slabs': (_c_i64, [_c_i64]),
        'gcrnn_all_zero_flag_bf16': (C.c_int, [_c_p, _c_i64, _c_p, _c_p]),
        'gcrnn_fused_gate_readout_backward_bf16': (C.c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_fused_gate_grad_bf16': (C.c_int, [_c_p] * 12 + [_c_i64] * 7 + [C.c_double, C.c_int, _c_p]),
        'gcrnn_fused_backward_data_bf16': (C.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                                     _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, C.c_double, _c_p, C.c_int, _c_p]),
        'gcrnn_fused_wgrad_slots': (_c_i64, [_c_i64, _c_i64]),
        'gcrnn_fused_wgrad_bf16_slots': (_c_i64, [_c_i64, _c_i64, _c_i64, _c_i64, C.c_int]),
        'gcrnn_fused_backward_weight_bf16': (C.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                                       _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p,
                                                       C.c_int, _c_p, C.c_double, _c_p, _c_p, _c_p]),
        'gcrnn_small_supported': (C.c_int, [C.c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
        'gcrnn_small_forward': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                          _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_small_backward_supported': (C.c_int, [C.c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
        'gcrnn_small_backward': (C.c_int, [C.c_int] + [_c_p] * 21 + [_c_i64] * 8 + [_c_p]),
        'gcrnn_small_dense_supported': (C.c_int, [C.c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, C.c_int, C.c_int]),
        'gcrnn_small_dense_forward': (C.c_int, [C.c_int] + [_c_p] * 9 + [_c_i64] * 10 + [_c_p]),
        'gcrnn_small_dense_backward': (C.c_int, [C.c_int] + [_c_p] * 16 + [_c_i64] * 10 + [_c_p]),
        'gcrnn_small_gates_supported': (C.c_int, [C.c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, C.c_int]),
        'gcrnn_small_gates_forward': (C.c_int, [C.c_int] + [_c_p] * 9 + [_c_i64] * 7 + [_c_p]),
        'gcrnn_small_gates_backward': (C.c_int, [C.c_int] + [_c_p] * 14 + [_c_i64] * 7 + [_c_p]),
        'gcrnn_node_linear_blocks': (_c_i64, [_c_i64, _c_i64]),
        'gcrnn_node_linear_forward': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_node_linear_backward': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_node_linear_bf16_supported': (C.c_int, [_c_i64, _c_i64, _c_i64]),
        'gcrnn_node_linear_bf16_blocks': (_c_i64, [_c_i64, _c_i64]),
        'gcrnn_node_linear_bf16_forward': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_node_linear_bf16_backward': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
        'gcrnn_l1_loss_blocks': (_c_i64, [_c_i64]),
        'gcrnn_l1_loss': (C.c_int, [C.c_int, _c_p, _c_p, _c_p, _c_p, _c_i64, C.c_double, _c_p]),
        'gcrnn_scale_unless_one': (C.c_int, [C.c_int, _c_p, _c_p, _c_i64, _c_p]),
        'gcrnn_attention_forward': (C.c_int, [C.c_int] + [_c_p] * 11 + [_c_i64] * 5 + [C.c_double, _c_p]),
        'gcrnn_attention_backward': (C.c_int, [C.c_int] + [_c_p] * 15 + [_c_i64] * 5 + [C.c_double, _c_p]),
        'gcrnn_ell_conflict_cycles': (C.c_int, [_c_p, _c_i64, _c_p, C.POINTER(_c_i64)]),
        'gcrnn_ell_pack_lds': (C.c_int, [_c_p, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTS = _load()


def check(status, what=''):
    if status != 0:
        detail = lib.gcrnn_last_hip_error().decode() if status == 5 else ''
        raise GcrnnError('%s failed: %s (status %d) %s' % (what or 'gcrnn call', lib.gcrnn_status_string(status).decode(),
                                                        status, detail))


def dtype_code(torch_dtype):
    import torch
    try:
        return {torch.float32: F32, torch.float64: F64, torch.bfloat16: BF16}[torch_dtype]
    except KeyError:
        raise GcrnnError('unsupported dtype %s (float32, float64, bfloat16)' % torch_dtype)
