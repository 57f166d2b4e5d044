/*
 * gcrnn.h -- C ABI of the MI355X-native gated-GCRNN hot path (libgcrnn_hip.so).
 *
 * The reference (luanaruiz9/gated_gcrnns) is pure Python/PyTorch and has no
 * FFI; its hot path sits behind torch.nn.Module classes.  This header is the
 * boundary a binding of that path would target: plain pointers and sizes, no
 * torch types.  Every entry point names the reference code it replaces
 * (paths relative to the reference repo).  The Python nn.Module mirror in
 * gated_gcrnns_amd/ reaches these through ctypes.
 *
 * Conventions
 *   - dtype: GCRNN_F32 / GCRNN_F64 (/ GCRNN_BF16 where stated): element type of
 *     every data/weight pointer of the call.
 *   - "user layout"      x[B][T][C][N]   node index minor   (graphML.py:2186-2192)
 *   - "node-major layout" X[T][N][B][C]  channel minor; one (t) slice is an
 *     [N][L = B*C] row matrix on which the graph shift is a CSR row SpMM.
 *   - The graph shift operator S (E = 1 slice) is passed as CSR arrays that live
 *     in DEVICE memory: int32 rowptr[N+1], int32 col[nnz], val[nnz] (dtype of
 *     the call; fp32 for GCRNN_BF16).  The reference's row-vector shift
 *     (x S)[g,n] = sum_m x[g,m] S[m,n] (graphML.py:116-123) is a row SpMM with
 *     CSR(S^T) on node-major data; its adjoint uses CSR(S).
 *   - stream: a hipStream_t passed as void* (NULL = default stream).  No entry
 *     point allocates, frees or synchronises; all are hipGraph-capturable.
 *   - return: 0 = GCRNN_OK, otherwise an error code (gcrnn_status_string()).
 *     Shape errors are reported, never clamped; the Python mirror raises.
 */
#ifndef GCRNN_H
#define GCRNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GCRNN_F32 = 0, GCRNN_F64 = 1, GCRNN_BF16 = 2 };

enum {
  GCRNN_OK = 0,
  GCRNN_ERR_BAD_DTYPE = 1,
  GCRNN_ERR_BAD_SHAPE = 2,
  GCRNN_ERR_NULL_POINTER = 3,
  GCRNN_ERR_UNSUPPORTED = 4,
  GCRNN_ERR_LAUNCH = 5,
  GCRNN_ERR_WORKSPACE = 6
};

int gcrnn_version(void);
const char* gcrnn_status_string(int status);
/* Text of the HIP runtime error behind the calling thread's last GCRNN_ERR_LAUNCH ("" if none). */
const char* gcrnn_last_hip_error(void);

/* ---- host-side graph preparation (no GPU needed) ------------------------------------------
 * Dense S (row-major N x N, double) -> CSR with ascending columns, keeping |S[i][j]| > tol.
 * transpose != 0 builds CSR of S^T (the forward shift operator on node-major data);
 * add_identity != 0 builds the pattern of S + I used by the attention mask
 * (graphML.py:577, 611-613).  Replaces the dense E x N x N GSO of graphML.py:117,123. */
int gcrnn_csr_count(const double* S, int64_t N, int transpose, int add_identity, double tol, int64_t* nnz);
int gcrnn_csr_fill(const double* S, int64_t N, int transpose, int add_identity, double tol,
                   int32_t* rowptr, int32_t* col, double* val);
/* Row processing order for the fused kernels: nodes sorted by descending degree (stable). */
int gcrnn_degree_order(const int32_t* rowptr, int64_t N, int32_t* order);

/* ---- layout ------------------------------------------------------------------------------
 * pack:   dst[t][n'][b][c] = src[b][t][c][perm ? perm[n'] : n']     (user -> node-major)
 * unpack: dst[b][t][c][perm ? perm[n'] : n'] = src[t][n'][b][c]     (node-major -> user)
 * perm (device int32[N]) may be NULL.  These replace the reference's narrow/view/cat state
 * plumbing (graphML.py:2353-2354, 2425-2427). */
int gcrnn_pack_node_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                          const int32_t* perm, void* stream);
/* gcrnn_pack_node_major of the sum over S slices (fp32): src [B][S][T][C][N] -> dst [T][N][B][C] = sum_s src[b][s][t][c][n], slices added in
 * order (deterministic). The node gates' tap dots (Utils/graphML.py:2387) leave the gate pre-pass as one partial per 32-feature chunk. */
int gcrnn_pack_node_major_sum_f32(const void* src, void* dst, int64_t B, int64_t S, int64_t T, int64_t C, int64_t N, void* stream);
/* Second stage of the node gates' F -> 1 GraphFilter (Utils/graphML.py:2387-2399) in one pass over the tap dots: parts fp32 [items][S][K][N]
 * (item = (t B + b) groups + g; S per-chunk partials, added in order) -> out fp32 [T][groups][B][N] = act(sum_k P^k u_k + bias[g]), P = the CSR
 * rows (rowptr int32[N + 1], col int32[nnz], val fp32[nnz]; uniform_w != 0: every stored entry has that weight and val is not read),
 * bias fp32[groups] or NULL, act = sigmoid (1) or identity (0).
 * gcrnn_node_gate_filter_supported: K <= 5, N <= 1024 and the CSR rows fit in LDS beside the running signal. */
int gcrnn_node_gate_filter_supported(int64_t K, int64_t N, int64_t nnz, double uniform_w);
int gcrnn_node_gate_filter_f32(const void* parts, void* out, int64_t items, int64_t S, int64_t K, int64_t N, int64_t groups, int64_t B,
                               const int32_t* rowptr, const int32_t* col, const void* val, int64_t nnz, double uniform_w, const void* bias,
                               int sigmoid, void* stream);
int gcrnn_unpack_node_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                            const int32_t* perm, void* stream);

/* ---- graph shift: batched CSR row SpMM ------------------------------------------------------
 * Y[i][n][l] = (accumulate ? Y[i][n][l] : 0) + sum_{j in row n} val[j] * X[i][col[j]][l],  i < nbatch, l < L.
 * Replaces x = torch.matmul(x, S) (graphML.py:123) and, with CSR(S), its adjoint. */
int gcrnn_spmm(int dtype, int64_t N, const int32_t* rowptr, const int32_t* col, const void* val,
               const void* X, void* Y, int64_t L, int64_t nbatch, int accumulate, void* stream);
/* The streaming kernel behind gcrnn_spmm with its tuning and epilogue exposed (graphs beyond LDS: BASELINE configs[4]).
 * Rows must be whole 16-byte vectors (L % (16 / elt) == 0), X / Y 16-byte aligned; GCRNN_BF16: bf16 rows, fp32 val / bias.
 * act = 1: Y = tanh(. + bias_scale * bias[l % F]) (bias may be NULL) -- the last hop of a Horner-form step writes h_t
 * (graphML.py:2420-2423: the one bias enters through both filters, bias_scale = 2, or gi + gf).
 * piece_lanes (0 = auto; 4..64, power of two): lanes per gathered row piece = column chunk of 16 * piece_lanes bytes; all
 * workgroups of one XCD work on the same chunk. unroll (0 = auto; 2, 4, 8): gather instructions in flight per wave.
 * rows_per_wave (0 = auto). Replaces x = torch.matmul(x, S) (graphML.py:123). */
int gcrnn_spmm_ex(int dtype, int64_t N, const int32_t* rowptr, const int32_t* col, const void* val, const void* X, void* Y,
                  int64_t L, int64_t nbatch, int accumulate, const void* bias, double bias_scale, int64_t F, int act,
                  int piece_lanes, int unroll, int rows_per_wave, void* stream);
/* All K taps of a Horner-form step in one pass on the matrix cores (bf16 rows, fp32 accumulation):
 *   u_k[r][:] = zh[r][:] B_k^T + zx[r][:] A_k^T,  k < K,  r < R rows of the node-major layout ([N][B] flattened).
 * wpack = gcrnn_fused_pack_weights(A, B); out0 [R][F] receives tap 0, outrest [K-1][R][F] taps 1..K-1. F in {32, 64},
 * G in {0, 32, 64} (zx NULL when G == 0). Replaces the per-tap contraction of LSIGF (graphML.py:134-135). */
/* The same on the fp32 / fp64 matrix cores (exact fp32 / fp64 products and sums: the 1e-5 / 1e-11 parity modes): zh [R][Ch],
 * zx [R][Cx] (NULL when Cx == 0), wB [F][Kst][Ch], wA [F][Kin][Cx] (the reference's F x 1 x K x C tap tensors as they are).
 * F % 16 == 0; Ch, Cx multiples of 16 (F32) / 8 (F64) with (Ch + Cx) / that in {1, 2, 4, 8, 16}. */
int gcrnn_taps_mfma_supported(int dtype, int64_t F, int64_t Ch, int64_t Cx);
int gcrnn_taps_mfma_forward(int dtype, const void* zh, const void* zx, const void* wB, const void* wA, void* out0, void* outrest,
                            int64_t R, int64_t F, int64_t Ch, int64_t Cx, int64_t Kst, int64_t Kin, void* stream);
int gcrnn_taps_bf16_supported(int64_t F, int64_t G, int64_t K);
int gcrnn_taps_bf16_forward(const void* zh, const void* zx, const void* wpack, void* out0, void* outrest, int64_t R,
                            int64_t F, int64_t G, int64_t K, void* stream);

/* ---- filter taps ----------------------------------------------------------------------------
 * rows = number of (t, n, b) rows; KK = E*K taps; z_0 = z0, z_k = zrest + (k-1)*zstride (k >= 1),
 * each [rows][G]; w = [F][KK][G] (the reference's F x E x K x G weight, graphML.py:2215-2216);
 * bias = [F] or NULL.
 * forward : y[r][f] = (accumulate ? y[r][f] : 0) + sum_{k,g} z_k[r][g] w[f][k][g] + bias_scale*bias[f]
 *           (graphML.py:134-139; bias_scale = 2 folds the double bias add of graphML.py:2420-2421)
 * bwd_data: dz_k[r][g] = sum_f dy[r][f] w[f][k][g]           (dz0 / dzrest mirror z0 / zrest)
 * bwd_wgt : dw_part[s][f][k][g] = sum over row split s of dy[r][f] z_k[r][g] ; dbias_part[b][f] = bias_scale * sum over row
 *           block b of dy[r][f]; s < splits, b < bias_blocks from gcrnn_taps_backward_weight_parts. Plain stores: the caller
 *           adds the partials in a fixed order (deterministic; no atomics). */
int gcrnn_taps_forward(int dtype, const void* z0, const void* zrest, int64_t zstride, const void* w,
                       const void* bias, double bias_scale, void* y, int64_t rows, int64_t KK, int64_t G,
                       int64_t F, int accumulate, void* stream);
int gcrnn_taps_backward_data(int dtype, const void* dy, const void* w, void* dz0, void* dzrest, int64_t zstride,
                             int64_t rows, int64_t KK, int64_t G, int64_t F, void* stream);
int gcrnn_taps_backward_weight_parts(int64_t rows, int64_t KK, int64_t G, int64_t F, int64_t* splits, int64_t* bias_blocks);
int gcrnn_taps_backward_weight(int dtype, const void* dy, const void* z0, const void* zrest, int64_t zstride,
                               void* dw_part, void* dbias_part, double bias_scale, int64_t rows, int64_t KK, int64_t G,
                               int64_t F, void* stream);

/* ==== fused flagship path (N <= gcrnn_fused_padded_nodes(), bf16 storage, fp32 accumulate) =====
 * Replaces the whole body of GGCRNNCell.forward's time loop (graphML.py:2351-2427) for the un-gated and
 * the time-gated cell: one launch per step computes h_t = tanh(gi (A(S)x_t + b) + gf (B(S)h_{t-1} + b)).
 *
 * Graph: degree-sorted sliced ELL built on the host from CSR(S^T):
 *   order[p]   = node in slot p (gcrnn_degree_order): tile t works on nodes order[16t .. 16t+15]; data rows
 *                stay in natural node order, the ordering only groups nodes of similar degree into a tile;
 *   tile_off[] = first entry of each tile (ntiles+1 values), entries padded to multiples of `pad` (= 4);
 *   ell_col / ell_val = [entry][16]: neighbour node id and weight of each slot (0, 0.0 for padding). */
int gcrnn_ell_size(const int32_t* rowptr, int64_t N, const int32_t* order, int tile, int pad, int64_t ntiles,
                   int64_t* nentries);
int gcrnn_ell_fill(const int32_t* rowptr, const int32_t* col, const double* val, int64_t N, const int32_t* order,
                   int tile, int pad, int64_t ntiles, const int32_t* node_addr, int32_t* tile_off, int32_t* ell_col,
                   float* ell_val);
/* ==== node-gated cell on the fused path (Utils/graphML.py:2379-2407) ================================================================
 *   h_t = tanh( gi ni_t[n] (A(S)x_t + b)[n][f] + gf nf_t[n] (B(S)h_{t-1} + b)[n][f] ),  ni / nf = sigmoid(GraphFilter_{F->1}(gate cell state))
 * gcrnn_fused_filter_output_bf16: out[t][b] = W(S) z + bias for every (t, b) in one launch (the x part does not depend on the
 *   recurrence); operand conventions of gcrnn_fused_gate_grad_bf16 (xs == NULL: zs is the operand and G = 0; else [0 | x_t]).
 * gcrnn_fused_node_forward_bf16: the T recurrent launches on the state-only operand; yx from the call above, ngates fp32
 *   [T][2][B][N] (input gates, forget gates), gi / gf fp32 [T][B] scalar time gates or both NULL, wpackB = state taps packed with
 *   G = 0, yh_out (or NULL) [T][B][NPad][F] receives B(S)h_{t-1} + b for the BPTT, Huser as in gcrnn_fused_forward_bf16.
 * gcrnn_node_gate_dot: s[item][k][n] = sum_f d[item][n][f] w[k][f] -- the F -> 1 node-gate filter taps-first (d = gate cell states
 *   [items][NPad][F] bf16 from gcrnn_fused_gate_prepass_bf16, w [K][F] fp32 = the GraphFilter weight 1 x 1 x K x F, graphML.py:2303);
 *   the K-1 hops then run on one-channel signals (gcrnn_spmm_ex on [N][items]). gcrnn_node_gate_dot_backward: ds [items][K][N] ->
 *   d overwritten by the gate cell's pre-activation gradient (sum_k ds_k w_k)(1 - d^2), dw_part [items][K][F] partial sums. */
int gcrnn_fused_filter_output_bf16(const void* zs, const void* xs, const void* wpack, const float* bias, void* out,
                                   const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col, const float* ell_val,
                                   const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N,
                                   int64_t F, int64_t G, int64_t K, double uniform_w, int img16 /* as in gcrnn_fused_backward_data_bf16 (forward plan) */,
                                   void* stream);
/* (huser_last_only: bit 0 and bit 1 as in gcrnn_fused_forward_bf16) */
int gcrnn_fused_node_forward_bf16(const void* h0s, void* hs, const void* yx, const float* ngates, const float* gi, const float* gf,
                                  const void* wpackB, const float* bias, void* yh_out, const int32_t* tile_nodes,
                                  const int32_t* tile_off, const int32_t* ell_col, const float* ell_val, const void* ell_val4,
                                  const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t K,
                                  void* Huser, int huser_last_only, double uniform_w, void* stream);
/* BPTT of the node-gated cell. gcrnn_fused_node_backward_data_bf16: the data-gradient chain dpre_t = (dH_t + rec_t)(1 - h_t^2),
 * rec_{t-1} = sum_k S^k ((gf nf)_t . dpre_t B_k): dHs, hs, dpre (out), dyh (out = (gf nf) . dpre) [T][B][NPad][F] bf16; ngf fp32
 * [T][B][N] = gf_t[b] nf_t[b][n]; wpackT and the graph arrays as in gcrnn_fused_backward_data_bf16 (ELL of CSR(S)).
 * gcrnn_node_cell_backward: all items at once -- d ni = gi <dpre, Yx>_f, d nf = gf <dpre, Yh>_f per node, d gi / d gf per item,
 * dyx = gi ni . dpre (the x filter's pre-activation gradient). The weight gradients then come from
 * gcrnn_fused_backward_weight_bf16 on dyx (input filter) and dyh (state filter). */
int gcrnn_fused_node_backward_data_bf16(const void* dHs, const void* hs, void* dpre, void* dyh, const float* ngf, const void* wpackT,
                                        const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col,
                                        const float* ell_val, const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B,
                                        int64_t T, int64_t N, int64_t F, int64_t K, double uniform_w,
                                        const void* dHuser_inline /* as in gcrnn_fused_backward_data_bf16 */, int img16 /* likewise */, void* stream);
int gcrnn_node_cell_backward(const void* dpre, const void* yx, const void* yh, const float* ngates, const float* gi, const float* gf,
                             void* dyx, float* dni, float* dnf, float* dgi, float* dgf, int64_t B, int64_t T, int64_t N,
                             int64_t NPad, int64_t F, void* stream);
int gcrnn_node_gate_dot(const void* d, const float* w, float* s, int64_t items, int64_t N, int64_t NPad, int64_t F, int64_t K,
                        void* stream);
int gcrnn_node_gate_dot_backward(void* d, const float* ds, const float* w, float* dw_part, int64_t items, int64_t N, int64_t NPad,
                                 int64_t F, int64_t K, void* stream);

/* ==== Edge-gated cell on the fused path (Utils/graphML.py:2409-2416; GraphAttentional graphML.py:1999-2128, graphAttention 521-627)
 *   h_t = tanh( gi att_in(A(S)x_t + b) + gf att_f(B(S)h_{t-1} + b) ),  att(y)[n] = relu(sum_m z_m (S+I)[m][n] alpha[m][n]),  z = W y,
 *   alpha[m][.] = softmax over the support row m of LeakyReLU(a1.z_n + a2.z_m).
 * W commutes with the shift, so z is a filter output with composite taps C_k W^T and bias W b: gcrnn_fused_filter_output_bf16
 * produces it (all items for the x branch, one call per step for the state branch) and gcrnn_fused_edge_attention_bf16 is the
 * attention itself, one workgroup per item: z [items][NPad][F] bf16, a12 fp32 [2][F] (mixer halves a1, a2), support rows
 * rowptr / r_edge = {n, bits of (S+I)[m][n]} and columns t_rowptr / t_edge = {m, bits} (int32 pairs), t_order = the nodes by
 * descending in-degree (the order in which the workgroup visits them).
 *   gx == NULL: out_seq = relu(att(z)) (the x branch, all items);
 *   gx != NULL: out_seq = h = tanh(gi gx + gf relu(att(z))) with per-item scalars gi / gf (or both NULL); r_out (or NULL) keeps
 *   relu(att(z)) for the BPTT; Huser (or NULL): item i's block in the user layout starts at Huser + i * huser_item_stride elements,
 *   element (f, n) at f * N + n. Rows >= N of the sequence-major outputs are written as zeros. N % 8 == 0, F in {32, 64}. */
int gcrnn_fused_edge_attention_supported(int64_t N, int64_t F);
int gcrnn_fused_edge_attention_bf16(const void* z, const float* a12, const void* gx, const float* gi, const float* gf,
                                    const int32_t* rowptr, const void* r_edge, const int32_t* t_rowptr, const void* t_edge,
                                    const int32_t* t_order, void* out_seq, void* r_out, void* Huser, int64_t huser_item_stride, int64_t items, int64_t N,
                                    int64_t NPad, int64_t F, double negative_slope, void* stream);

/* Backward of gcrnn_fused_edge_attention_bf16 for one branch (autograd of graphAttention composed with the layer's ReLU and the
 * branch's scalar gate): dpre = d loss / d (cell pre-activation) [items][NPad][F] bf16, r = relu(att(z)) kept by the forward,
 * g [items] (or NULL = 1) -> dz [items][NPad][F] bf16 (w.r.t. the composite filter output), da_part fp32 [items][2][F] (per-item
 * partials of the mixer gradient), dgate fp32 [items] = sum dpre . r (or NULL). r_order = support rows by descending out-degree,
 * t_pos = position of every column-ordered support edge in the row order, scratch fp32 [items][nnz]. Rows with more than 32 support
 * entries take a slower chunked loop. Deterministic (gathers and fixed-order sums only).
 * gcrnn_fused_backward_step_bf16: ONE launch of the BPTT data chain with explicit arrays, dpre_prev = (sum_k S^k (operand W_k) +
 * dH_prev)(1 - h_prev^2) -- the edge-gated cell alternates it with the attention backward (the chain's operand of step t is dz_t);
 * gcrnn_fused_backward_seed_bf16: dpre = dH (1 - h^2) on bf16 arrays (the last step). */
int gcrnn_fused_edge_attention_backward_supported(int64_t N, int64_t F, int64_t max_out_degree);
int gcrnn_fused_edge_attention_backward_bf16(const void* dpre, const void* r, const float* g, const void* z, const float* a12,
                                             const int32_t* rowptr, const void* r_edge, const int32_t* r_order,
                                             const int32_t* t_rowptr, const int32_t* t_pos, float* scratch, void* dz, float* da_part,
                                             float* dgate, int64_t items, int64_t N, int64_t NPad, int64_t F, int64_t nnz,
                                             int64_t max_out_degree /* of the support: > 32 selects the hub variant (8 waves, chunked extra records) */,
                                             double negative_slope, void* stream);
int gcrnn_fused_backward_step_bf16(const void* operand, const void* dH_prev, const void* h_prev, void* dpre_prev, const void* wpackT,
                                   const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col, const float* ell_val,
                                   const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B, int64_t N, int64_t F,
                                   int64_t K, double uniform_w,
                                   const void* dHuser_next /* NULL, or the user-layout block dH[0][t-2] ... */, void* dHs_next /* ... laid out into dHs[t-2] */,
                                   int64_t T /* sequence length (item stride T F N of the user-layout tensor) */,
                                   int img16 /* as in gcrnn_fused_backward_data_bf16 */, void* stream);
int gcrnn_fused_backward_seed_bf16(const void* dH, const void* h, void* dpre, int64_t elements, void* stream);

/* ==== fp32-accurate fused path ("x3": three bf16 planes per fp32 operand, six partial products on the bf16 matrix cores) ========
 * The un-gated cell h_t = tanh(A(S)x_t + b + B(S)h_{t-1} + b) (Utils/graphML.py:2420-2423) to fp32 accuracy (the north_star's
 * 1e-5 mode) at fused-kernel speed: v = v1 + v2 + v3 with v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2) for state,
 * input and taps; z.w = z1w1 + z1w2 + z1w3 + z2w1 + z2w2 + z3w1 with fp32 accumulation; hops / bias / tanh in fp32.
 * gcrnn_pack_seq_major_x3: user fp32 [B][T][C][N] -> planes [T][3][B][NPad][C] bf16 (C even), rows N..NPad-1 zero.
 * gcrnn_fused_pack_weights_x3: fp32 taps wA [F][Kin][G], wB [F][Kst][F] -> wpack3 [3][F/16][K][(F+G)/32][64][8] bf16.
 * gcrnn_fused_forward_x3: T launches; xs3 [T][3][B][NPad][G], h03 [3][B][NPad][F], hs3 [T][3][B][NPad][F]; bias fp32 [F] or NULL;
 * graph arrays of a UNIFORM-weight plan (gcrnn_ell_assign_rows_z / gcrnn_ell_fill_z; ell_col4 = the column half of
 * gcrnn_ell_pack_lds; uniform_w = the one weight); Huser fp32 [B][T][F][N] (or [B][1][F][N] with last_only != 0) or NULL.
 * gcrnn_fused_x3_supported: F = G in {32, 64} or F = 64 with G = 32, K in 2..5, N <= 1024, N % 4 == 0, and the image fits LDS
 * (64 KiB state + 3 tap planes + 32 B x entries <= 160 KiB). */
/* (r4) rank1 (last pointer of every x3 compute entry; NULL = a uniform-weight graph): a RANK-1-weighted graph S[m][n] = a[m] b[n] on its support
 * (normalised adjacencies, Utils/graphTools.py:64) runs on the plan of its 0/1 PATTERN (uniform_w = 1) with a [4][NPad] fp32 table
 * a | a b | 1 / b | b (1 where b = 0; zeros in the padding rows of the first two): the Horner recursion is carried in t' = t / b, t'_j = u_j / b +
 * sum_{m in N(n)} (a b t'_{j+1})[m], t_0 = b t'_0, so the uniform stream still adds in place. Adjoint plans take the table of S^T (a and b swap). */
int gcrnn_fused_x3_supported(int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries);
int gcrnn_pack_seq_major_x3(const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N, int64_t NPad, void* stream);
int gcrnn_fused_pack_weights_x3(const void* wA, const void* wB, void* wpack3, int64_t F, int64_t G, int64_t Kin, int64_t Kst,
                                void* stream);
int gcrnn_fused_forward_x3(const void* xs3, const void* h03, void* hs3, const void* wpack3, const float* bias,
                           const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B,
                           int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, double uniform_w, void* Huser, int last_only,
                           const float* rank1, void* stream);
/* gcrnn_fused_forward_x3 with a per-(t, b) weight of the bias (bias_scale [T][B] fp32, NULL = 2) and an explicit distance between the
 * sequences of Huser (huser_seq_stride elements, 0 = T*F*N; Huser = the block of step 0). The time-gated cell at fp32 accuracy
 * (Utils/graphML.py:2357-2374, 2420-2423) is composed from it: gi (A(S)x_t + b) + gf (B(S)h_{t-1} + b) = A(S)(gi x_t) + B(S)(gf h_{t-1}) +
 * (gi + gf) b -- the caller scales the operands in fp32 and passes gi + gf. */
int gcrnn_fused_forward_x3_scaled(const void* xs3, const void* h03, void* hs3, const void* wpack3, const float* bias,
                                  const float* bias_scale, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4,
                                  int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, double uniform_w,
                                  void* Huser, int64_t huser_seq_stride, const float* rank1, void* stream);
/* fp32-accurate BPTT of the un-gated cell on the fused kernels (round 3): the training loop of the reference runs in the drivers'
 * precision (Modules/train_rnn.py:247-281 under kStepPredGRNNs.py:44), i.e. its gradients are autograd's of Utils/graphML.py:2420-2423.
 * gcrnn_fused_backward_data_x3: the data chain dpre_{t-1} = (sum_k S^k (dpre_t B_k^T) + dH_{t-1}) (1 - h_{t-1}^2) on the x3 step kernel
 *   (three bf16 planes per operand, six partial products, fp32 hops); dHs3 / hs3 / dpre3 [T][3][B][NPad][F] planes, dh03 [3][B][NPad][F]
 *   or NULL, wpack3T = gcrnn_fused_pack_weights_x3 of the TRANSPOSED state taps with G = 0, graph arrays of the ADJOINT uniform plan.
 * gcrnn_fused_backward_weight_f32: dW_k = sum du_k^T [h_{t-1} | x_t] with exact fp32 products (v_mfma_f32_16x16x4_f32); X / H / h0 fp32
 *   in the USER layout, per-slot partial sums dW [slots][F][K][F+G], dbsum [slots][F] (= 2 sum dpre: the bias enters both filters)
 *   with slots = gcrnn_fused_wgrad_slots(B T, F), added by the caller in a fixed order (bit-reproducible).
 * gcrnn_fused_x3_training_supported: the forward's shapes and an adjoint image that fits next to the two fp32 images. */
int gcrnn_fused_x3_training_supported(int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries_adj);
int gcrnn_fused_backward_data_x3(const void* dHs3, const void* hs3, void* dpre3, void* dh03, const void* wpack3T,
                                 const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                 int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, double uniform_w, const float* rank1, void* stream);
int gcrnn_fused_backward_weight_f32(const void* dpre3, const void* Xuser, const void* Huser, const void* h0user, float* dW,
                                    float* dbsum, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4,
                                    int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K,
                                    double uniform_w, const float* rank1, void* stream);
/* fp32-accurate BPTT of the TIME-GATED cell (round 4; Utils/graphML.py:2357-2374 + 2420-2423 under autograd -- the reference's default
 * cell, :2196, in the drivers' precision, kStepPredGRNNs.py:44):
 * gcrnn_fused_backward_data_x3_gated: dpre_{t-1} = (gf_t sum_k S^k (dpre_t B_k^T) + dH_{t-1}) (1 - h_{t-1}^2), gf [T][B] fp32; dh03 (required)
 *   receives gf_0 x the raw gradient of the initial state; dgf_parts (or NULL; needs h03 = planes of h0 [3][B][NPad][F]):
 *   [T][B][F/16 * 8] fp32 partials of <h_{t-1}, sum_k S^k (dpre_t B_k^T)> = <B(S) h_{t-1}, dpre_t> (adjoint identity), the filter part of
 *   d loss / d gf_t, added by the caller in a fixed order.
 * gcrnn_fused_filter_x3: ONE graph filter without bias or nonlinearity (Utils/graphML.py:47-140 LSIGF) out = sum_k S^k (z W_k), z3 / out3
 *   [3][B][NPad][F] planes, wpack3 = gcrnn_fused_pack_weights_x3 of the taps with G = 0 (F features in and out), uniform plan: d loss / d gi_t
 *   = <A(S) x_t + b, dpre_t> is read off A(S) x_t over all items (t, b).
 * gcrnn_fused_backward_weight_f32_gated: gcrnn_fused_backward_weight_f32 with item (t, b) entering the input-filter columns with weight
 *   gi[t][b] and the state-filter columns with gf[t][b] (fp32 scaling of the operand, as the forward scaled it); dbsum = partials of
 *   sum (gi + gf) sum_n dpre. */
int gcrnn_fused_backward_data_x3_gated(const void* dHs3, const void* hs3, void* dpre3, void* dh03, const void* wpack3T,
                                       const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                       int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, double uniform_w, const float* gf,
                                       const void* h03, float* dgf_parts, const float* rank1, void* stream);
int gcrnn_fused_filter_x3(const void* z3, void* out3, const void* wpack3, const int32_t* tile_nodes, const int32_t* tile_off,
                          const void* ell_col4, int64_t entries, int64_t B, int64_t N, int64_t F, int64_t K, double uniform_w,
                          const float* rank1, void* stream);
int gcrnn_fused_backward_weight_f32_gated(const void* dpre3, const void* Xuser, const void* Huser, const void* h0user, float* dW,
                                          float* dbsum, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4,
                                          int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K,
                                          double uniform_w, const float* gi, const float* gf, int h_is_h0, const float* rank1, void* stream);
/* ... and the pieces its gate cells (Utils/graphML.py:2362-2374: both gates read (x_t, h0), never h_{t-1}) run on:
 * gcrnn_fused_backward_weight_f32_gated with gi = gf = NULL (weights 1 / 2) and h_is_h0 != 0: every item's state operand is h0 (Huser unused;
 *   h0user = NULL: a zero initial state, train_rnn.py:256 -- the state columns are skipped and stay zero).
 * gcrnn_fused_gate_cells_x3: the T x B independent one-step cells c[t][b] = tanh(A_g(S) x_t + B_g(S) h0 + 2 b_g) as T launches of the x3 step
 *   kernel that all read h03; xs3 = the planes of X [T][3][B][NPad][G], scratch3 [3][B][NPad][F], Cuser fp32 [B][T][F][N] (out); state_zero != 0: h0 is all
 *   zeros and the state half of the tap products is skipped (exact).
 * gcrnn_pack_seq_major_x3_ex: gcrnn_pack_seq_major_x3 of v' = item_scale[t][b] * rowmul[c][n] * (one_minus_square ? 1 - v^2 : v) (factors
 *   optional; src_seq_stride = elements between the sequences of src, 0 = contiguous): the scaled operands gi x_t / gf h_{t-1} and the gate
 *   cells' upstream gradient d logit * w * (1 - c^2) without an fp32 copy.
 * gcrnn_x3_item_dots: per item (t, b) of plane arrays [T][3][B][NPad][C]: <a, b> and sum_{n,c} a[n][c] vec[c] (either optional), fixed order:
 *   d loss / d gi_t = <A(S) x_t, dpre_t> + <b, sum_n dpre_t>. */
int gcrnn_fused_gate_cells_x3(const void* xs3, const void* h03, void* scratch3, const void* wpack3, const float* bias,
                              const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B,
                              int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, double uniform_w, void* Cuser, int state_zero, const float* rank1, void* stream);
int gcrnn_pack_seq_major_x3_ex(const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N, int64_t NPad,
                               const float* item_scale, const float* rowmul, int one_minus_square, int64_t src_seq_stride, void* stream);
int gcrnn_x3_item_dots(const void* a3, const void* b3, const float* vec, float* out_ab, float* out_av, int64_t B, int64_t T, int64_t NPad,
                       int64_t C, void* stream);
/* One step of the NODE-gated cell at fp32 accuracy behind its two x3 filter passes (round 5; Utils/graphML.py:2402-2407, 2420-2423):
 *   h = tanh( ni (ya + bias) + nf (yb + bias) ), ya3 / yb3 / h3 [3][B][NPad][F] bf16 planes (h3: the next step's operand), ni / nf fp32 [B][NPad] (the
 *   step's node gates, time gate folded in by the caller), bias fp32 [F] or NULL, Huser fp32 or NULL: element (b, f, n) at b huser_seq_stride + f N + n.
 *   F in {32, 64}, N % 4 == 0, NPad % 64 == 0. */
int gcrnn_x3_node_gate_step(const void* ya3, const void* yb3, const float* ni, const float* nf, const float* bias, void* h3, void* Huser,
                            int64_t huser_seq_stride, int64_t B, int64_t N, int64_t NPad, int64_t F, void* stream);


/* LDS placement of the hop state (tile = 16). node_addr[n] = (row << 6) | (swz << 4): node n's 64-byte state row sits at
 * LDS row `row`, its four 16-byte quads XOR-swizzled by `swz` (NULL = identity: row n, swz (n >> 2) & 3).
 * gcrnn_ell_assign_rows chooses rows and swizzles (local search over the 16 bank-quad keys a node can take) so that no
 * tile asks for one key more often than it has entries AND (r2) the 8 slots of every half tile sit on 8 different write keys
 * (row & 1) << 2 | swz, which makes the kernels' ds_write_b128 write-backs of the state conflict-free (`order` must then place
 * every node in exactly one slot; GCRNN_PLAN_LEGACY_KEYS=1 in the environment keeps the gather-only search);
 * gcrnn_ell_fill then orders each row's neighbours -- and aims its
 * zero-weight padding entries -- by peeling perfect matchings off the slots x keys multigraph: the 16-lane groups of the
 * kernel's ds_read_b128 gathers hit distinct banks (greedy maximum matchings where a key is over-subscribed).
 * gcrnn_ell_conflict_cycles reports the LDS cycles per gather summed over entries (4 * entries = conflict-free);
 * gcrnn_ell_pack_lds builds the kernel's LDS image: val4 [entries/4][16][4] fp32, col4 [entries/4][16][4] uint16 =
 * node_addr of the neighbour. N = padded node count here (multiple of 16, <= 1024). */
int gcrnn_ell_assign_rows(const int32_t* rowptr, const int32_t* col, int64_t N, const int32_t* order, int pad,
                          int64_t ntiles, int32_t* node_addr);
/* The same two steps for a graph whose non-zeros all carry ONE weight (the reference drivers' S = W / lambda_max of an
 * unweighted adjacency, kStepPredGRNNs.py:768): zero_from >= 0 names the first padding row (rows zero_from .. N-1 hold zeros
 * for ever); the first 16 of them get the 16 bank keys (fixed) and every padding entry points at one of them, so a kernel may drop the weight
 * image and compute init + w * (sum of the gathered rows). zero_from = -1: the calls above. GCRNN_ERR_UNSUPPORTED when fewer
 * than 16 padding rows exist. */
int gcrnn_ell_assign_rows_z(const int32_t* rowptr, const int32_t* col, int64_t N, const int32_t* order, int pad,
                            int64_t ntiles, int64_t zero_from, int32_t* node_addr);
int gcrnn_ell_fill_z(const int32_t* rowptr, const int32_t* col, const double* val, int64_t N, const int32_t* order,
                     int tile, int pad, int64_t ntiles, const int32_t* node_addr, int64_t zero_from, int32_t* tile_off,
                     int32_t* ell_col, float* ell_val);
int gcrnn_ell_conflict_cycles(const int32_t* ell_col, int64_t entries, const int32_t* node_addr, int64_t* cycles);
int gcrnn_ell_pack_lds(const int32_t* ell_col, const float* ell_val, int64_t entries, const int32_t* node_addr, float* val4,
                       uint16_t* col4);
/* 1 if gcrnn_fused_forward_bf16 has a kernel for this shape (K = max(Kin, Kst) taps). */
int gcrnn_fused_supported(int64_t N, int64_t F, int64_t G, int64_t K);
int64_t gcrnn_fused_padded_nodes(void);
/* Waves per workgroup of the step kernels (forward, gate pre-pass, BPTT data gradient) and of the weight-gradient kernel.
 * A plan's storage tiles are dealt out wave-major: tile w * (NPad/16/waves) + i belongs to wave w, so a plan is built for
 * one of the two wave counts (degree-ranked tiles go round-robin over the waves). */
int64_t gcrnn_fused_step_waves(void);
int64_t gcrnn_fused_wgrad_waves(void);
/* user [B][T][C][N] <-> sequence-major [T][B][NPad][C]; row p holds node perm[p] (perm NULL = identity);
 * rows >= N are zero. dtype GCRNN_BF16 or GCRNN_F32 (same type both sides). */
int gcrnn_pack_seq_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                         int64_t NPad, const int32_t* perm, void* stream);
/* bf16 pack with channel padding: user [B][T][Cs][N] -> sequence-major [T][B][NPad][C], channels Cs .. C-1 and rows N .. NPad-1 zero
 * (C even, N even). The fused kernels consume the input in 32-feature steps; the reference drivers feed G = 1 (kStepPredGRNNs.py:220,
 * epicenterEstimation.py:170): this lays their X out for the kernels without a zero-padded copy in the user layout. */
int gcrnn_pack_seq_major_padded(const void* src, void* dst, int64_t B, int64_t T, int64_t Cs, int64_t C, int64_t N, int64_t NPad,
                                void* stream);
int gcrnn_unpack_seq_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                           int64_t NPad, const int32_t* perm, void* stream);
/* The bf16 pack (no permutation; N and C even) of the time steps [t0, t1) only, on a kernel with a 4 KiB LDS footprint:
 * issued on a second stream it runs beside the step kernels (one workgroup per CU, a few KiB of LDS left), so the packs of
 * later steps hide behind the recurrence of earlier ones (gcrnn_fused_forward_bf16's step_events). */
int gcrnn_pack_seq_major_steps(const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N, int64_t NPad,
                               int64_t t0, int64_t t1, int64_t max_blocks /* > 0: bound on the grid (the tiles are walked) */,
                               void* stream);
/* weight_A [F][Kin][G] and weight_B [F][Kst][F] (E = 1; wdtype GCRNN_F32 or GCRNN_BF16) -> bf16 MFMA A-operand
 * fragments wpack[F/16][K][(F+G)/32][64 lanes][8], K = max(Kin, Kst), missing taps zero. */
int gcrnn_fused_pack_weights(int wdtype, const void* wA, const void* wB, void* wpack, int64_t F, int64_t G,
                             int64_t Kin, int64_t Kst, void* stream);
/* xs [T][B][NPad][G], h0 [B][NPad][F], hs [T][B][NPad][F]: bf16 sequence-major, natural node order;
 * bias fp32 [F] or NULL; gi / gf fp32 [T][B] time gates or both NULL (un-gated);
 * tile_nodes int32 [NPad]: slot p = (node << 16) | node_addr[node], node = order padded with the unused row ids
 * N..NPad-1; entries = tile_off[ntiles]; ell_col int32 [entries][16] = node_addr of each neighbour (NOT the node id).
 * ell_val4 / ell_col4 = device copies of gcrnn_ell_pack_lds's output (may be NULL).
 * Huser (may be NULL; needs N % 8 == 0): the states are ALSO written in the user layout H[B][T][F][N] by the step kernels
 * themselves (LDS-transposed 16-byte row stores), which replaces gcrnn_unpack_seq_major over the whole sequence.
 * T launches on `stream`. The graph is kept resident in LDS when 64 KiB + weights + 96*entries B <= 160 KiB.
 * huser_last_only bit 1 (value 2, r2): tile_nodes / ell_col4 are the arrays of a bf16 hop image (un-gated or time-gated cell, uniform_w != 0;
 *   GraphOperator.fused_plan_img16 (head_w allowed): 32-byte state rows, neighbour rows summed on the matrix cores); GCRNN_ERR_UNSUPPORTED otherwise.
 * huser_last_only bit 0: Huser is [B][1][F][N] and receives the LAST state only (the classification models read nothing else,
 * architectures.py:1841-1850); the other steps skip the user-layout store.
 * step_events (or NULL): host array of T hipEvent_t (entries may be NULL); the launch of step t first makes `stream` wait for
 * step_events[t] -- xs[t] is then allowed to be produced on another stream while earlier steps run.
 * uniform_w != 0: every non-zero of the graph carries this one weight AND the graph arrays come from gcrnn_ell_assign_rows_z /
 * gcrnn_ell_fill_z (padding entries aimed at zero rows): the step kernels then keep only the column words in LDS and sum the
 * gathered rows (acc = init + w * sum). 0 = weighted graph image. The same trailing parameter exists on the gate pre-pass, the
 * filter-output pass, the node-gated steps and both BPTT data chains. */
int gcrnn_fused_inline_pack_supported(int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries, double uniform_w);
/* Time steps ONE launch of gcrnn_fused_forward_bf16 (backward = 0) / gcrnn_fused_backward_data_bf16 (backward != 0) covers for this
 * problem: T (T - 1 for the chain) when the sequence-resident persistent kernel takes it -- one workgroup per sequence keeps the
 * operand [h_{t-1} | x_t] in registers for all F/16 chunks and runs every time step of the recurrence (the T-loop of
 * Utils/graphML.py:2351-2427) inside one launch --, 1 when that kernel is launched per step (GCRNN_SEQ_PERSIST=0), 0 when the
 * chunk-parallel step kernel runs (gated cells, fused head, weighted graphs, batches that do not fill rounds of 256 sequences). */
int64_t gcrnn_fused_seq_steps_per_launch(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                         double uniform_w, int img16, int inline_pack, int backward);
int gcrnn_fused_forward_bf16(const void* xs, const void* h0, void* hs, const void* wpack, const float* bias,
                             const float* gi, const float* gf, const int32_t* tile_nodes, const int32_t* tile_off,
                             const int32_t* ell_col, const float* ell_val, const void* ell_val4, const void* ell_col4,
                             int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K,
                             void* Huser, int huser_last_only, void* const* step_events, double uniform_w,
                             const void* Xuser_inline /* NULL, or X [B][T][G][N] bf16 in the user layout: launch t also lays out x_{t+1} into
                                xs[t+1] (only xs[0] has to be packed by the caller); un-gated cell, gcrnn_fused_inline_pack_supported() */,
                             const float* head_w /* NULL, or [F]: the weights of an output head Linear(F -> 1) shared by all nodes (architectures.py:
                                1616-1627 with one output), fused onto the h_t store ... */,
                             float* head_part /* ... its partial sums fp32 [T][B][F/16][N] (the caller adds the F/16 chunks and the bias); with a head,
                                Huser may be NULL: the regression model's inference then never materialises H */,
                             void* stream);

/* ---- The wide sequence-resident kernel (round 4; csrc/gcrnn_fused_seq32.h): the recurrence of Utils/graphML.py:2351-2427 as ONE launch,
 * one workgroup per sequence, 32 output features per chunk as two 16-feature image planes read by one hop stream, the graph weight folded
 * into the taps (W_k <- w^k W_k), h_t handed to step t+1 in registers. Replaces gcrnn_fused_forward_bf16 for un-gated and time-gated cells
 * on uniform-weight bf16-image plans when the batch fills whole rounds of the chip, and both gcrnn_fused_gate_prepass_*_bf16 launches of a
 * time-gated cell by ONE pass over the (t, b) items.
 * gcrnn_fused_pack_weights_wide: taps -> MFMA A fragments wpack [Fout/32][K][2][(F+G)/32][64][8] bf16 (Fout output rows over F state and G
 *   input features, all multiples of 32: Fout = F for a cell, 2 F for the two time gates' sub-cells stacked); MFMA row m of (chunk c, half h)
 *   is output feature 32 c + 8 (m >> 2) + 4 h + (m & 3); tap k is scaled by uniform_w^k.
 * gcrnn_fused_forward_wide_supported: 1 when gcrnn_fused_forward_wide_bf16 takes the problem (inline_pack: with Xuser_inline).
 * gcrnn_fused_forward_wide_bf16: xs [T][B][NPad][G] bf16 sequence-major -- every step laid out, or with Xuser_inline (the user-layout
 *   X [B][T][G][N] bf16, N % 8 == 0; un-gated only) steps 0 and 1 only: the launch lays out the rest, each step one hop before the step that
 *   reads it ends --, h0 [B][NPad][F], hs [T][B][NPad][F] (out), bias [F] fp32 or NULL (added by both filters, graphML.py:2420-2421),
 *   gi / gf: NULL, or the scalar time gates [T][B] fp32 (graphML.py:2357-2374), tile_nodes / tile_off / ell_col4 = the bf16-image plan
 *   (graph.fused_plan_img16), Huser [B][T][F][N] bf16 or NULL (huser_last_only: [B][1][F][N], the last state only).
 * gcrnn_fused_gate_pair_wide_supported: 0, or -- with_pack -- the number of leading time steps of xs the caller lays out itself (1 without).
 * gcrnn_fused_gate_pair_prepass_wide_bf16: both time gates of every (t, b) in one launch; wpack = gcrnn_fused_pack_weights_wide(Fout = 2 F) of
 *   [GFL_in ; GFL_forget] stacked over the output features, bias2 [2 F], gw2 [2][N][F] fp32 = the read-outs' weights node-major, parts
 *   [T*B][2 * F/32 * 8] fp32 partial dot products (chunks 0 .. F/32-1 the input gate's; fixed-order sum by the caller), cs_in / cs_f (both or
 *   neither) the sub-cells' states [T][B][NPad][F] bf16, x_user as in gcrnn_fused_gate_prepass_pack_bf16, h0_zero_flag as in
 *   gcrnn_fused_gate_prepass_bf16; rank1_a / rank1_b as in gcrnn_fused_forward_wide_bf16 (`img16` bit 1 of the query), which now takes them
 *   together with gi / gf too (the time-gated recurrence on a rank-1-weighted graph). */
int gcrnn_fused_pack_weights_wide(int wdtype, const void* wA, const void* wB, void* wpack, int64_t Fout, int64_t F, int64_t G, int64_t Kin,
                                  int64_t Kst, double uniform_w, void* stream);
int gcrnn_fused_forward_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries, double uniform_w,
                                       int img16, int inline_pack);
int gcrnn_fused_forward_wide_bf16(const void* xs, const void* h0, void* hs, const void* wpack, const float* bias, const float* gi,
                                  const float* gf, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                  int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, void* Huser, int huser_last_only,
                                  const void* Xuser_inline,
                                  const float* rank1_a, const float* rank1_b /* both NULL, or -- un-gated forward on a RANK-1-weighted graph
                                     S[m][n] = a[m] b[n] (normalised adjacencies, Utils/graphTools.py:64) -- the two factors [NPad] fp32 (zero for
                                     padding rows); the plan arrays are then those of the graph's 0/1 pattern and wpack carries uniform_w = 1 */,
                                  void* stream);
int gcrnn_fused_gate_pair_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries, double uniform_w,
                                         int img16, int with_pack);
/* The BPTT data chain as ONE launch of the wide kernel: gcrnn_fused_backward_data_bf16's contract (seed included), with wpackT =
 * gcrnn_fused_pack_weights_wide of the transposed state taps (G = 0), the bf16-image plan of the ADJOINT graph, dgf_parts [T][B][F/32*8].
 * rank1_a / rank1_b (both or neither; `img16` bit 1 of the query): a rank-1-weighted graph S[m][n] = a[m] b[n] (normalised adjacencies) on the
 * adjoint plan of its 0/1 pattern -- the factor tables [NPad] fp32 of S^T, i.e. the forward direction's b and a (uniform_w = 1 in the weight pack). */
int gcrnn_fused_backward_data_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, int64_t entries, double uniform_w, int img16,
                                             int inline_pack);
int gcrnn_fused_backward_data_wide_bf16(const void* dHs, const void* hs, void* dpre, void* dh0, const void* wpackT, const int32_t* tile_nodes,
                                        const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N,
                                        int64_t F, int64_t K, const float* gf, const void* h0s, float* dgf_parts, const void* dHuser_inline,
                                        const float* rank1_a, const float* rank1_b, void* stream);
/* Round 5: the node-gated cell's two state-size passes on the wide kernel (csrc/gcrnn_fused_seq32.h modes 3 and 4; reference
 * Utils/graphML.py:2379-2407, 2420-2423). Until round 4 they ran the 16-feature kernels (gcrnn_fused_filter_output_bf16,
 * gcrnn_fused_node_forward_bf16: 1.45 + 1.75 ms per forward at N = 1000, F = G = 64, K = 5, T = 32, B = 256).
 * gcrnn_fused_filter_output_wide_bf16: A(S) x_t + b for every (t, b) item in ONE launch -- xs [T][B][NPad][G] bf16 sequence-major, wpack =
 *   gcrnn_fused_pack_weights_wide(Fout = F) of the input taps with ZERO state taps (operand [0 | x_t]; the state half is neither loaded nor
 *   multiplied), bias [F] fp32 or NULL, out [T][B][NPad][F] bf16 (no activation).
 * gcrnn_fused_node_forward_wide_bf16: gcrnn_fused_node_forward_bf16's contract without yh_out (inference) as ONE launch: h0s [B][NPad][F], hs
 *   [T][B][NPad][F] (out), yx = the filter-output pass, ngates fp32 [T][2][B][N] (input gates, forget gates), gi / gf fp32 [T][B] or both NULL,
 *   wpackB = gcrnn_fused_pack_weights_wide of the state taps alone (G = 0), Huser [B][T or 1][F][N] bf16 or NULL.
 * The _supported queries return 1 when the problem is taken (uniform-weight bf16-image plan: img16 == 1, a batch that fills the chip, LDS room). */
int gcrnn_fused_filter_output_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries, double uniform_w,
                                             int img16, int with_pack /* 1: returns the number of leading time steps of xs the caller lays out itself */);
int gcrnn_fused_filter_output_wide_bf16(const void* xs, const void* wpack, const float* bias, void* out, const int32_t* tile_nodes,
                                        const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N,
                                        int64_t F, int64_t G, int64_t K,
                                        const void* x_user /* NULL, or the user-layout X [B][T][G][N] bf16 (N % 8 == 0): the items lay out the time steps of xs
                                                              the caller has not (edge-gated cell without time gates: no separate pass over X) */,
                                        void* stream);
int gcrnn_fused_node_forward_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, int64_t entries, double uniform_w, int img16);
int gcrnn_fused_node_forward_wide_bf16(const void* h0s, void* hs, const void* yx, const float* ngates, const float* gi, const float* gf,
                                       const void* wpackB, const float* bias, const int32_t* tile_nodes, const int32_t* tile_off,
                                       const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, void* Huser,
                                       int huser_last_only, void* stream);
/* The time gates' read-out finished in ONE launch (Utils/graphML.py:2364-2366, 2372-2374): gi[i] = sigmoid(sum_j parts[i][0][j] + lb_in),
 * gf[i] likewise with parts[i][1][..] and lb_f; parts [items][2][nparts] fp32 as gcrnn_fused_gate_pair_prepass_wide_bf16 leaves them (items = T B),
 * summed in a fixed order; lb_in / lb_f: device scalars (fp32) or NULL. */
int gcrnn_gate_readout_finish(const float* parts, int64_t nparts, const float* lb_in, const float* lb_f, float* gi, float* gf, int64_t items,
                              void* stream);
int gcrnn_fused_gate_pair_prepass_wide_bf16(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias2,
                                            const float* gw2, float* parts, void* cs_in, void* cs_f, const int32_t* tile_nodes,
                                            const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N,
                                            int64_t F, int64_t G, int64_t K, const int32_t* h0_zero_flag, const float* rank1_a, const float* rank1_b,
                                            void* stream);
/* gcrnn_fused_gate_pair_prepass_wide_bf16 for the NODE gates (Utils/graphML.py:2379-2393): both gate cells of every (t, b) in one launch, and
 * instead of a read-out the first stage of their F -> 1 graph filters GFL_node_* (:2303, 2318), taps first (:2387): tapf [2][F/32][3][64] x 16 B
 * = the taps' A fragments (three bf16 planes, p0 + p1 + p2 = w to 24 bits; lane 16 kg + tap: w_p[tap][32 cg + 8 kg .. + 7], taps >= ntaps
 * zero), taps_out [T*B][2][F/32][ntaps][N] fp32 = the partial dots per 32-feature chunk (the caller adds a gate's chunks, runs the one-channel
 * hops and the sigmoid). */
int gcrnn_fused_gate_pair_prepass_taps_wide_bf16(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias2,
                                                 const void* tapf, float* taps_out, int64_t ntaps, void* cs_in, void* cs_f,
                                                 const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                                 int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, const int32_t* h0_zero_flag,
                                                 const float* rank1_a, const float* rank1_b, void* stream);

/* Time-gate pre-pass (graphML.py:2357-2374): for every (t, b)
 *   sum over gate_out[t][b][0 .. F/16*8) = sum_{n,f} tanh( A_g(S) x_t + B_g(S) h0 + 2 b_g )[n][f] * gate_w[n][f]
 *   (gate_out holds the F/16 * 8 per-workgroup-wave partials as plain stores: summing them in a fixed order keeps the
 *    gate deterministic, which float atomics would not)
 * with the gate sub-cell's packed weights (gcrnn_fused_pack_weights of GFL_in / GFL_forget) and gate_w = the gate's
 * Linear(N*F -> 1) weight re-laid node-major [N][F] (fp32). All T*B items run in ONE launch because the reference's gates
 * read h0, never h_{t-1} (graphML.py:2362, 2370). The caller applies sigmoid(sum + c).
 * h0_zero_flag (or NULL): device int32; non-zero = h0 is all zeros (every training loop of the reference starts there,
 * train_rnn.py:256): the state half of the operand then contributes exactly nothing and its loads and MFMAs are skipped.
 * cs (or NULL): [T][B][NPad][F] bf16, receives the gate cell's state c_t = tanh(.) for the gate's BPTT (padded rows zero).
 * Launches are split over whole time steps where T*B*NPad*F*2 bytes exceed the 32-bit buffer offsets. */
int gcrnn_fused_gate_prepass_bf16(const void* xs, const void* h0, const void* wpack, const float* bias,
                                  const float* gate_w, float* gate_out, void* cs, const int32_t* tile_nodes,
                                  const int32_t* tile_off, const int32_t* ell_col, const float* ell_val,
                                  const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B, int64_t T,
                                  int64_t N, int64_t F, int64_t G, int64_t K, const int32_t* h0_zero_flag, double uniform_w,
                                  int img16 /* as in gcrnn_fused_backward_data_bf16 (forward plan) */, void* stream);

/* The same pre-pass, which ALSO lays out the input: x_user = X [B][T][G][N] bf16 in the reference's layout (graphML.py:2409: B x T x G x N;
 * G already padded to the kernels' 32 / 64 channels), xs = the sequence-major array [T][B][NPad][G] whose first
 * gcrnn_fused_gate_prepass_lays_out(...) time steps the caller laid out (gcrnn_pack_seq_major_steps); on return every step is laid
 * out. Each item of the sequence-resident kernel packs the operand of its workgroup's next item while its own hops run, as the
 * forward steps do for x_{t+1} -- gated cells no longer need a pass over X before their first pre-pass.
 * GCRNN_ERR_UNSUPPORTED where gcrnn_fused_gate_prepass_lays_out returns 0. */
int gcrnn_fused_gate_prepass_pack_bf16(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias,
                                       const float* gate_w, float* gate_out, void* cs, const int32_t* tile_nodes,
                                       const int32_t* tile_off, const int32_t* ell_col, const float* ell_val,
                                       const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B, int64_t T,
                                       int64_t N, int64_t F, int64_t G, int64_t K, const int32_t* h0_zero_flag, double uniform_w,
                                       int img16, void* stream);
/* 0 = not available for this shape / graph (weighted graph, N % 8 != 0, a batch the sequence-resident kernel does not take, LDS);
 * else the number of leading time steps of xs the caller lays out itself (the items of the first round of workgroups). */
int64_t gcrnn_fused_gate_prepass_lays_out(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                          double uniform_w, int img16);

/* Gate-cell pre-pass of a NODE gate at inference (graphML.py:2379-2393) with the first stage of the gate's F -> 1 GraphFilter fused into its
 * epilogue (taps first, :2387): taps_out[t*B + b][k][n] = < tanh(A_g(S) x_t + B_g(S) h0 + 2 b_g)[n, :], w_k >, k < ntaps <= 8, fp32 --
 * computed from the bf16-rounded states (the values cs would hold) on the matrix cores. tap_frags: the taps as A fragments of
 * v_mfma_f32_16x16x16_bf16, bf16 [F/16][3][64][4]: plane p of the three-way bf16 split of the fp32 taps (p0 + p1 + p2 = w to 24 bits),
 * lane l = 16 kg + tap: w_p[tap][16 chunk + 4 kg + e], taps >= ntaps zero. cs (or NULL): also store the states; x_user (or NULL): also lay
 * out X (as gcrnn_fused_gate_prepass_pack_bf16, same contract for xs). GCRNN_ERR_UNSUPPORTED where ..._taps_supported returns 0. */
int gcrnn_fused_gate_prepass_taps_bf16(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias,
                                       const void* tap_frags, float* taps_out, int64_t ntaps, void* cs, const int32_t* tile_nodes,
                                       const int32_t* tile_off, const int32_t* ell_col, const float* ell_val, const void* ell_val4,
                                       const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G,
                                       int64_t K, const int32_t* h0_zero_flag, double uniform_w, int img16, void* stream);
int gcrnn_fused_gate_prepass_taps_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                            double uniform_w, int img16, int with_pack, int64_t ntaps);

/* d loss / d (scalar time gate) of ONE filter of the time-gated cell (the gates multiply the filter outputs, graphML.py:2420-2421):
 *   sum over out[t][b][0 .. F/16*8) = sum_{f,n} ( W(S) z[t][b] + bias )[n][f] * dpre[t][b][n][f]
 * xs == NULL, G = 0: z = zs [T][B][NPad][F] bf16 sequence-major is that filter's operand (h_{t-1} for the state filter, x_t for
 * an input filter with G == F), wpack = gcrnn_fused_pack_weights of its taps as a state-only operand (G = 0);
 * xs != NULL: input filter with G != F: operand [0 | x_t] with xs [T][B][NPad][G], zs = ONE all-zero block [NPad][F] and
 * wpack = gcrnn_fused_pack_weights(A, zero state taps). bias [F] or NULL (added once), dpre = output of
 * gcrnn_fused_backward_data_bf16. All T*B items in one launch (split like the gate pre-pass). */
int gcrnn_fused_gate_grad_bf16(const void* zs, const void* xs, const void* dpre, const void* wpack, const float* bias, float* out,
                               const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col, const float* ell_val,
                               const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N,
                               int64_t F, int64_t G, int64_t K, double uniform_w /* as in gcrnn_fused_forward_bf16 */,
                               int img16 /* as in gcrnn_fused_backward_data_bf16 (forward plan) */, void* stream);

/* BPTT through a time gate's read-out gate = sigmoid(w . vec(c) + c0) (graphML.py:2364-2366), one pass, in place:
 *   cs [items][NPad][F] bf16: on entry the gate cell's states c (gcrnn_fused_gate_prepass_bf16), on return
 *   dpre_g = dlogit[item] * w[n][f] * (1 - c^2) (the operand of gcrnn_fused_backward_weight_bf16 with h_is_h0);
 *   dw_part [gcrnn_fused_gate_readout_slabs(items)][NPad*F] fp32: per-slab partial sums of dlogit[item] * c (plain stores;
 *   the caller adds the slabs in a fixed order); gate_w [N][F] fp32 node-major read-out weights; dlogit [items] fp32. */
int64_t gcrnn_fused_gate_readout_slabs(int64_t items);
/* flag int32[1] (device) = 1 when all `elements` bf16 values at src are +-0, else 0: the time-gated cell's "h0 is all zeros" flag (every training
 * loop of the reference starts from zeros, Modules/train_rnn.py:256; the gate kernels then skip the state half of their operand), decided on
 * the device. elements % 8 == 0, src 16-byte aligned. */
int gcrnn_all_zero_flag_bf16(const void* src, int64_t elements, int32_t* flag, void* stream);
int gcrnn_fused_gate_readout_backward_bf16(void* cs, const float* dlogit, const float* gate_w, float* dw_part, int64_t items,
                                           int64_t N, int64_t F, void* stream);

/* BPTT data gradient of the fused cell (adjoint of graphML.py:2420-2423), bf16 sequence-major arrays [T][B][NPad][F]:
 *   dpre[T-1] = dHs[T-1] * (1 - hs[T-1]^2);   for t = T-1 .. 1:
 *   dpre[t-1] = ( gf[t] sum_k (S)^k (dpre[t] B_k) + dHs[t-1] ) * (1 - hs[t-1]^2);   dh0 = gf[0] sum_k (S)^k (dpre[0] B_k)  (optional)
 * gf: [T][B] fp32 forget gates of the time-gated cell, or NULL (= 1).
 * dgf_parts (or NULL; needs h0s = h0 [B][NPad][F] bf16 sequence-major): [T][B][F/16*8] fp32 partials whose sum over the last
 * axis is <h_{t-1}, sum_k (S)^k (dpre[t] B_k)> = <B(S) h_{t-1}, dpre[t]> -- by the adjoint identity the bias-free part of
 * d loss / d gf[t][b], read off the chain this entry point evaluates anyway (the dh0 launch then always runs; dh0 may be NULL).
 * dHs = gradient of the loss w.r.t. every state, hs = the states of the forward. wpackT = gcrnn_fused_pack_weights of the
 * TRANSPOSED state taps (wB^T [F_in][Kst][F_out] passed as "wB", G = 0); the graph arrays are the ELL of CSR(S) (the adjoint
 * shift). One launch per step; same kernel as the forward with a different epilogue. */
int gcrnn_fused_backward_data_bf16(const void* dHs, const void* hs, void* dpre, void* dh0, const void* wpackT,
                                   const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col,
                                   const float* ell_val, const void* ell_val4, const void* ell_col4, int64_t entries,
                                   int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, const float* gf, const void* h0s,
                                   float* dgf_parts, double uniform_w,
                                   const void* dHuser_inline /* NULL, or dH [B][T][F][N] bf16 in the user layout: the launch that consumes dHs[t-1]
                                      also lays out dHs[t-2] (the caller packs steps T-2, T-1 only); gcrnn_fused_inline_pack_supported(N, F, F, ...) */,
                                   int img16 /* != 0: the graph arrays address a bf16 hop image (GraphOperator.fused_plan_img16(adjoint=True), uniform_w != 0) */,
                                   void* stream);

/* BPTT weight gradient of the fused cell (adjoint of the taps, graphML.py:134-135), all T*B items in ONE launch:
 *   dW[f'][k][j] += sum_{t,b,n} g[t][b] (S^k dpre[t][b])[n][f'] * z[t][b][n][j],   z = [h_{t-1} | x_t],   j < F: weight_B (g = gf),
 *   j >= F: weight_A (g = gi); gi / gf: [T][B] fp32 time gates or both NULL (= 1). h_is_h0 != 0: the state operand of every
 *   item is h0 (the gate sub-cells, graphML.py:2362, 2370; Huser may then be NULL).
 * dpre: output of gcrnn_fused_backward_data_bf16; Xuser [B][T][G][N], Huser [B][T][F][N] (the forward's output) and
 * h0user [B][F][N] are the bf16 USER-layout tensors (node-contiguous rows feed the matrix cores directly; needs N % 8 == 0);
 * dW fp32 [slots][F][K][F+G], slots = gcrnn_fused_wgrad_bf16_slots(B*T, F, K, entries, img16) (round 5: on the bf16-image plans a workgroup visits
 * an item for TWO 16-feature chunks of dpre, so an item has half the workgroups and twice the slots fill the chip; gcrnn_fused_wgrad_slots(B*T, F)
 * is the count of the fp32-accurate kernel, gcrnn_fused_backward_weight_f32): every workgroup slot stores ITS partial sum with plain
 * stores (the caller zero-fills the buffer and adds the slots in a fixed order: no atomics, two runs give the same bits);
 * graph arrays = LDS image of the ELL of CSR(S).
 * Returns GCRNN_ERR_UNSUPPORTED when the graph image does not fit in LDS next to the state. 
 * h_is_h0 bit 1 (value 2, r2): the graph arrays address a bf16 hop image (GraphOperator.fused_plan_img16(adjoint=True), uniform_w != 0). */
int64_t gcrnn_fused_wgrad_slots(int64_t items, int64_t F);
int64_t gcrnn_fused_wgrad_bf16_slots(int64_t items, int64_t F, int64_t K, int64_t entries, int img16);
int gcrnn_fused_backward_weight_bf16(const void* dpre, const void* Xuser, const void* Huser, const void* h0user, float* dW,
                                     float* dbsum /* [slots][F] partials of the bias gradient sum_{t,b} (gi + gf) sum_n dpre (2 sum dpre without gates), or NULL */,
                                     const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_val4,
                                     const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F,
                                     int64_t G, int64_t K, const float* gi, const float* gf, int h_is_h0,
                                     const int32_t* h0_zero_flag /* with h_is_h0 (or NULL): device int32, non-zero = h0 is all zeros */,
                                     double uniform_w /* != 0: uniform-weight image (gcrnn_ell_fill_z): both image halves in LDS, two barriers per tap */,
                                     const float* rank1_a, const float* rank1_b /* both or neither; with the bf16 hop image only: a rank-1-weighted graph
                                        S[m][n] = a[m] b[n] on the adjoint plan of its 0/1 pattern (uniform_w = 1) -- [NPad] fp32 factor tables of S^T */,
                                     void* stream);

/* ==== small-graph regime: the whole T-step recurrence of a sequence inside one workgroup, one launch ============
 * Replaces GGCRNNCell.forward (graphML.py:2336-2427, un-gated or time-gated with precomputed gates) when
 * K*(G+F)*N values plus weights and CSR fit in LDS (gcrnn_small_supported) -- the drivers' own configurations
 * (N = 50..80, G = 1, F = 20). User layout in and out, no transposes:
 *   X [B][T][G][N], h0 [B][F][N] -> H [B][T][F][N]; wA [F][Kin][G], wB [F][Kst][F], bias [F] or NULL (E = 1);
 *   gi / gf [T][B] time gates or both NULL; CSR(S^T) rowptr/col/val (val in the data dtype). dtype F32 or F64. */
int gcrnn_small_supported(int dtype, int64_t N, int64_t nnz, int64_t G, int64_t F, int64_t Kin, int64_t Kst);
int gcrnn_small_forward(int dtype, const void* X, const void* h0, const void* wA, const void* wB, const void* bias,
                        const void* gi, const void* gf, const int32_t* rowptr, const int32_t* col, const void* val,
                        void* H, int64_t B, int64_t T, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst,
                        int64_t nnz, void* stream);

/* BPTT of the small-graph cell in one launch (adjoint of GGCRNNCell.forward, graphML.py:2336-2427; replaces the autograd
 * graph PyTorch records for the T-step loop). H = forward output, dH = gradient w.r.t. every state [B][T][F][N];
 * CSR(S^T) (rowptr/col/val) and CSR(S) (arowptr/acol/aval). Per-sequence partial sums, to be added over the first two
 * dimensions by the caller in a fixed order:  pA [B][2][F][Kin][G], pB [B][2][F][Kst][F], pb [B][F];
 * dgi / dgf [T][B] (time-gated cells, else NULL); dh0 [B][F][N] or NULL. dX is not produced. */
int gcrnn_small_backward_supported(int dtype, int64_t N, int64_t nnz, int64_t G, int64_t F, int64_t Kin, int64_t Kst);
int gcrnn_small_backward(int dtype, const void* X, const void* h0, const void* H, const void* dH, const void* wA,
                         const void* wB, const void* bias, const void* gi, const void* gf, const int32_t* rowptr,
                         const int32_t* col, const void* val, const int32_t* arowptr, const int32_t* acol, const void* aval,
                         void* pA, void* pB, void* pb, void* dgi, void* dgf, void* dh0, int64_t B, int64_t T, int64_t N,
                         int64_t G, int64_t F, int64_t Kin, int64_t Kst, int64_t nnz, void* stream);

/* Same contract on the matrix cores for graphs whose DENSE N x N GSO fits in LDS (the drivers' N = 50..80): every hop,
 * tap, weight-gradient and adjoint product of a time step is a small GEMM on v_mfma_f64_16x16x4_f64 /
 * v_mfma_f32_16x16x4_f32 with operands read from LDS. Sdense = S itself, row-major [N][N] in the data dtype
 * (z S: out[c][n] = sum_m z[c][m] S[m][n]). Gates here are PER NODE: gi / gf multiply the input / state filter's output at
 * every node; element (b, t, n) is read at b * gate_stride_b + t * gate_stride_t + n * gate_stride_n, so [T][B] time gates
 * (strides 1, B, 0), [B][T][N] node gates (graphML.py:2379-2407) or their product (strides T N, N, 1) need no copy; both
 * NULL = un-gated. backward: pA [B][F][Kin][G], pB [B][F][Kst][F], pb [B][F] per-sequence partial sums (added over B by the
 * caller), dgi / dgf [B][T][N] (dense; a scalar gate's gradient is their sum over n), dh0 [B][F][N] or NULL. */
int gcrnn_small_dense_supported(int dtype, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, int backward,
                                int gated);
int gcrnn_small_dense_forward(int dtype, const void* X, const void* h0, const void* wA, const void* wB, const void* bias,
                              const void* gi, const void* gf, const void* Sdense, void* H, int64_t B, int64_t T, int64_t N,
                              int64_t G, int64_t F, int64_t Kin, int64_t Kst, int64_t gate_stride_b, int64_t gate_stride_t,
                              int64_t gate_stride_n, void* stream);
int gcrnn_small_dense_backward(int dtype, const void* X, const void* h0, const void* H, const void* dH, const void* wA,
                               const void* wB, const void* bias, const void* gi, const void* gf, const void* Sdense,
                               void* pA, void* pB, void* pb, void* dgi, void* dgf, void* dh0, int64_t B, int64_t T,
                               int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, int64_t gate_stride_b,
                               int64_t gate_stride_t, int64_t gate_stride_n, void* stream);

/* Time gates of the small-graph regime (GGCRNNCell time gating, graphML.py:2248-2278, 2357-2374), both gates in one launch:
 *   gate[g][t][b] = sigmoid( lw_g . vec_{F,N}( tanh(A_g(S) x_t + B_g(S) h0 + 2 b_g) ) + lb_g ),  g = 0 input, 1 forget.
 * Parameters stacked over the two gates: wA2 [2][F][Kin][G], wB2 [2][F][Kst][F], bias2 [2][F] or NULL, lw2 [2][F*N]
 * (nn.Linear weight, row-major over (f, n)), lb2 [2] or NULL. backward: dsum [2][T][B] = d loss / d (pre-sigmoid value);
 * per-sequence partial sums pA [B][2][F][Kin][G], pB [B][2][F][Kst][F], pb [B][2][F], plw [B][2][F*N], plb [B][2],
 * pdh0 [B][2][F][N] (or NULL). */
int gcrnn_small_gates_supported(int dtype, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, int backward);
int gcrnn_small_gates_forward(int dtype, const void* X, const void* h0, const void* wA2, const void* wB2, const void* bias2,
                              const void* lw2, const void* lb2, const void* Sdense, void* gate, int64_t B, int64_t T,
                              int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, void* stream);
int gcrnn_small_gates_backward(int dtype, const void* X, const void* h0, const void* wA2, const void* wB2, const void* bias2,
                               const void* lw2, const void* Sdense, const void* dsum, void* pA, void* pB, void* pb, void* plw,
                               void* plb, void* pdh0, int64_t B, int64_t T, int64_t N, int64_t G, int64_t F, int64_t Kin,
                               int64_t Kst, void* stream);

/* ==== per-node output head =========================================================================================
 * mlpType = 'multipMlp' of GatedGCRNNforRegression (architectures.py:1616-1627: one Linear(F -> O) applied to every node's
 * state in a Python loop over nodes), on the user layout: h [R][F][N] -> y [R][O][N], R = B * T, F <= 64, O <= 8, F32 / F64.
 * backward: dh [R][F][N] (or NULL), pw [gcrnn_node_linear_blocks(R, N)][O][F] and pb [blocks][O] partial sums of dw / db. */
int64_t gcrnn_node_linear_blocks(int64_t R, int64_t N);
int gcrnn_node_linear_forward(int dtype, const void* h, const void* w, const void* b, void* y, int64_t R, int64_t N, int64_t F,
                              int64_t O, void* stream);
int gcrnn_node_linear_backward(int dtype, const void* h, const void* w, const void* dy, void* dh, void* pw, void* pb, int64_t R,
                               int64_t N, int64_t F, int64_t O, void* stream);

/* The same head on bf16 activations (the fused cell's output): h, dy, y, dh are bf16 arrays, the parameters w [O][F], b [O]
 * are fp32 (wdtype GCRNN_F32: master weights) or bf16, accumulation and the partial sums pw [blocks][O][F], pb [blocks][O]
 * (blocks = gcrnn_node_linear_bf16_blocks(R, N); added by the caller in a fixed order) are fp32. N even, F <= 64, O <= 2. */
int gcrnn_node_linear_bf16_supported(int64_t N, int64_t F, int64_t O);
int64_t gcrnn_node_linear_bf16_blocks(int64_t R, int64_t N);
int gcrnn_node_linear_bf16_forward(int wdtype, const void* h, const void* w, const void* b, void* y, int64_t R, int64_t N,
                                   int64_t F, int64_t O, void* stream);
int gcrnn_node_linear_bf16_backward(int wdtype, const void* h, const void* w, const void* dy, void* dh, float* pw, float* pb,
                                    int64_t R, int64_t N, int64_t F, int64_t O, void* stream);

/* ==== training-loop loss ==========================================================================================
 * batchTimeL1Loss (Utils/miscTools.py:112-119 = nn.L1Loss: mean |x - y| over every entry) and its gradient in one pass.
 * x, y, grad: n contiguous elements of `dtype` (F32 / F64 / BF16), 16-byte aligned; grad (may be NULL) =
 * sign(x - y) * inv_n; partial: gcrnn_l1_loss_blocks(n) sums of |x - y| (fp32 for F32 / BF16, fp64 for F64) that the
 * caller adds up in a fixed order and scales by inv_n. */
int64_t gcrnn_l1_loss_blocks(int64_t n);
int gcrnn_l1_loss(int dtype, const void* x, const void* y, void* grad, void* partial, int64_t n, double inv_n, void* stream);
/* data[i] *= r[0] in place unless the DEVICE scalar r[0] (fp32; fp64 for fp64 data) equals 1 -- the chain rule through a scalar loss
 * whose gradient tensor gcrnn_l1_loss already wrote, without a second pass over it when the upstream gradient is 1. */
int gcrnn_scale_unless_one(int dtype, void* data, const void* r, int64_t n, void* stream);
/* batchTimeMSELoss, the drivers' metric (Utils/miscTools.py:121-130): x, y as [R][C] matrices (R = batch * time rows,
 * C = N * F columns; F32 / F64 / BF16): out[0] = mean_c sqrt(sum_r (x - y)^2) / sqrt(sum_r y^2). part: scratch of
 * gcrnn_batch_time_mse_slabs(R, C) * 2 * C accumulators (fp64 for F64, else fp32); out: one accumulator. Deterministic. */
int64_t gcrnn_batch_time_mse_slabs(int64_t R, int64_t C);
int gcrnn_batch_time_mse(int dtype, const void* x, const void* y, void* part, void* out, int64_t R, int64_t C, void* stream);

/* ==== optimiser ====================================================================================================
 * torch.optim.Adam (kStepPredGRNNs.py:158-161, stepped at train_rnn.py:276) over ONE flat parameter / gradient / moment
 * buffer of n elements (F32 or F64): the gradient buffer is the one the data-parallel all-reduce has just reduced.
 * g is multiplied by grad_scale on the fly. step_dev: device int64 step counter, incremented by the call before use
 * (capturable: no host value is baked into the launch). */
int gcrnn_adam_flat(int dtype, void* p, const void* g, void* m, void* v, int64_t n, double lr, double beta1, double beta2,
                    double eps, double grad_scale, int64_t* step_dev, void* stream);

/* ==== edge gate: graph attention on the CSR support of S + I ====================================================
 * Replaces graphAttention (graphML.py:521-627: dense B x N x N scores, mask, softmax, weighted sum) inside
 * GraphAttentional.forward (graphML.py:2099-2107). Node-major, T independent slices, dtype F32 / F64:
 *   Wx [T][N][B][F] = W u,  s1 / s2 [T][N][B] = a1 . Wx / a2 . Wx  (graphML.py:585-603),
 *   support row m: neighbours col[j], values val[j] = (S + I)[m][col[j]], |.| > 1e-9 (graphML.py:577, 611-613);
 *   transposed support: for column n, t_row[q] = m and t_pos[q] = index of edge (m, n) in col / val.
 *   alpha [T][nnz][B] (out; softmax over each row of LeakyReLU(s1[n] + s2[m]), kept for backward),
 *   y [T][N][B][F] (out) = sum_m alpha[m->n] val[m->n] Wx[m]  (graphML.py:625), before the nonlinearity.
 * backward: dy -> dWx (through the aggregation only), ds1, ds2; dz_scratch [T][nnz][B]. No atomics: deterministic. */
int gcrnn_attention_forward(int dtype, const int32_t* rowptr, const int32_t* col, const void* val, const int32_t* t_rowptr,
                            const int32_t* t_row, const int32_t* t_pos, const void* Wx, const void* s1, const void* s2,
                            void* alpha, void* y, int64_t T, int64_t N, int64_t B, int64_t F, int64_t nnz,
                            double negative_slope, void* stream);
int gcrnn_attention_backward(int dtype, const int32_t* rowptr, const int32_t* col, const void* val,
                             const int32_t* edge_row /* [nnz] row m of every edge */, const int32_t* t_rowptr, const int32_t* t_pos, const void* Wx, const void* s1, const void* s2,
                             const void* alpha, const void* dy, void* dWx, void* ds1, void* ds2, void* dz_scratch, int64_t T,
                             int64_t N, int64_t B, int64_t F, int64_t nnz, double negative_slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCRNN_H */
