// Edge gate: graph attention restricted to the CSR support of S + I (reference graphAttention, graphML.py:521-627,
// which materialises a dense B x N x N score tensor and masks it; here nothing of size N^2 exists).
//
// Node-major operands, T independent slices (the gates of all time steps are evaluated in one batched pass):
//   Wx [T][N][B][F]   W u            (graphML.py:585-586)
//   s1 [T][N][B]      a1 . Wx_n      s2 [T][N][B]   a2 . Wx_m        (graphML.py:591-603)
//   row m of the support has neighbours n_j with values v_j = (S + I)[m][n_j]:
//     e_j = LeakyReLU(s1[n_j] + s2[m]);  alpha_j = softmax_j(e_j)     (graphML.py:605-622; masked entries contribute exactly 0)
//     y[n] = sum_{m : n in row m} alpha[m -> n] v[m -> n] Wx[m]       (graphML.py:625)
// `alpha` ([T][nnz][B]) is kept for the backward pass. The transposed support (t_rowptr / t_row / t_pos) lists for
// every column n the rows m that reach it and the position of that edge in the row-ordered arrays.
// Every kernel is a gather (no atomics): results are deterministic.
#include "gcrnn_common.h"

namespace {

template <typename T> __device__ __forceinline__ T exp_t(T v);
template <> __device__ __forceinline__ float exp_t<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double exp_t<double>(double v) { return exp(v); }

// one thread per (t, m, b), b fastest: alpha of row m
template <typename T>
__global__ __launch_bounds__(256) void edge_alpha_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const T* __restrict__ s1, const T* __restrict__ s2,
                                                         T* __restrict__ alpha, int64_t total, int N, int B, int64_t nnz,
                                                         T slope) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx % B);
  const int64_t tm = idx / B;
  const int m = (int)(tm % N);
  const int64_t t = tm / N;
  const T* s1t = s1 + t * N * B + b;
  T* at = alpha + t * nnz * B + b;
  const T z2 = s2[idx];
  const int j0 = rowptr[m], j1 = rowptr[m + 1];
  T mx = -INFINITY;
  for (int j = j0; j < j1; ++j) {
    const T z = s1t[(int64_t)col[j] * B] + z2;
    const T e = z > T(0) ? z : slope * z;
    mx = e > mx ? e : mx;
  }
  T den = T(0);
  for (int j = j0; j < j1; ++j) {
    const T z = s1t[(int64_t)col[j] * B] + z2;
    const T e = z > T(0) ? z : slope * z;
    const T ex = exp_t<T>(e - mx);
    at[(int64_t)j * B] = ex;
    den += ex;
  }
  for (int j = j0; j < j1; ++j) at[(int64_t)j * B] = at[(int64_t)j * B] / den;
}

// one thread per (t, n, b, f), f fastest:  out[n] = sum_q alpha[pos_q] val[pos_q] src[row_q]
// forward: (lists = transposed support) y from Wx;   backward: (lists = the support itself, pos = identity) dWx from dy
template <typename T, bool IDENTITY_POS>
__global__ __launch_bounds__(256) void edge_aggregate_kernel(const int32_t* __restrict__ lptr, const int32_t* __restrict__ lidx,
                                                             const int32_t* __restrict__ lpos, const T* __restrict__ val,
                                                             const T* __restrict__ alpha, const T* __restrict__ src,
                                                             T* __restrict__ out, int64_t total, int N, int B, int F,
                                                             int64_t nnz) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int f = (int)(idx % F);
  const int64_t r = idx / F;
  const int b = (int)(r % B);
  const int64_t tn = r / B;
  const int n = (int)(tn % N);
  const int64_t t = tn / N;
  const T* at = alpha + t * nnz * B + b;
  const T* st = src + (t * N * B + b) * F + f;
  T acc = T(0);
  for (int q = lptr[n]; q < lptr[n + 1]; ++q) {
    const int p = IDENTITY_POS ? q : lpos[q];
    acc += at[(int64_t)p * B] * val[p] * st[(int64_t)lidx[q] * B * F];
  }
  out[idx] = acc;
}

// one thread per (t, edge j = (m -> n), b), b fastest: d alpha_j = v_j (Wx[m] . dy[n]) -> dz [T][nnz][B]
// (edge-parallel: T * nnz * B threads instead of a serial loop over the row's edges)
template <typename T>
__global__ __launch_bounds__(256) void edge_bwd_dot_kernel(const int32_t* __restrict__ erow, const int32_t* __restrict__ col,
                                                           const T* __restrict__ val, const T* __restrict__ Wx,
                                                           const T* __restrict__ dy, T* __restrict__ dz, int64_t total,
                                                           int N, int B, int F, int64_t nnz) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx % B);
  const int64_t tj = idx / B;
  const int64_t j = tj % nnz;
  const int64_t t = tj / nnz;
  const T* wx = Wx + ((t * N + erow[j]) * B + b) * F;
  const T* dyn = dy + ((t * N + col[j]) * B + b) * F;
  T d0 = T(0), d1 = T(0);
  int f = 0;
  for (; f + 2 <= F; f += 2) { d0 += wx[f] * dyn[f]; d1 += wx[f + 1] * dyn[f + 1]; }
  if (f < F) d0 += wx[f] * dyn[f];
  dz[idx] = (d0 + d1) * val[j];
}

// one thread per (t, m, b): d e_j = alpha_j (d alpha_j - sum_i alpha_i d alpha_i);  d z_j = d e_j LeakyReLU'(z_j) -> dz
// (in place);  d s2[m] = sum_j d z_j
template <typename T>
__global__ __launch_bounds__(256) void edge_bwd_row_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const T* __restrict__ s1, const T* __restrict__ s2,
                                                           const T* __restrict__ alpha, T* __restrict__ dz,
                                                           T* __restrict__ ds2, int64_t total, int N, int B, int64_t nnz,
                                                           T slope) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx % B);
  const int64_t tm = idx / B;
  const int m = (int)(tm % N);
  const int64_t t = tm / N;
  const T* s1t = s1 + t * N * B + b;
  const T* at = alpha + t * nnz * B + b;
  T* dzt = dz + t * nnz * B + b;
  const T z2 = s2[idx];
  const int j0 = rowptr[m], j1 = rowptr[m + 1];
  T S = T(0);
  for (int j = j0; j < j1; ++j) S += at[(int64_t)j * B] * dzt[(int64_t)j * B];
  T sum = T(0);
  for (int j = j0; j < j1; ++j) {
    const T de = at[(int64_t)j * B] * (dzt[(int64_t)j * B] - S);
    const T z = s1t[(int64_t)col[j] * B] + z2;
    const T g = de * (z > T(0) ? T(1) : slope);
    dzt[(int64_t)j * B] = g;
    sum += g;
  }
  ds2[idx] = sum;
}

// one thread per (t, n, b): d s1[n] = sum over the edges that reach n of d z
template <typename T>
__global__ __launch_bounds__(256) void edge_bwd_col_kernel(const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_pos,
                                                           const T* __restrict__ dz, T* __restrict__ ds1, int64_t total,
                                                           int N, int B, int64_t nnz) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx % B);
  const int64_t tn = idx / B;
  const int n = (int)(tn % N);
  const int64_t t = tn / N;
  const T* dzt = dz + t * nnz * B + b;
  T acc = T(0);
  for (int q = t_rowptr[n]; q < t_rowptr[n + 1]; ++q) acc += dzt[(int64_t)t_pos[q] * B];
  ds1[idx] = acc;
}

inline unsigned blocks_for(int64_t total) { return (unsigned)cdiv(total, 256); }

template <typename T>
int attention_forward_t(const int32_t* rowptr, const int32_t* col, const void* val, const int32_t* t_rowptr,
                        const int32_t* t_row, const int32_t* t_pos, const void* Wx, const void* s1, const void* s2,
                        void* alpha, void* y, int64_t Tn, int64_t N, int64_t B, int64_t F, int64_t nnz, double slope,
                        hipStream_t st) {
  GCRNN_PRE_LAUNCH();
  const int64_t rows = Tn * N * B;
  edge_alpha_kernel<T><<<blocks_for(rows), 256, 0, st>>>(rowptr, col, (const T*)s1, (const T*)s2, (T*)alpha, rows, (int)N,
                                                         (int)B, nnz, (T)slope);
  edge_aggregate_kernel<T, false><<<blocks_for(rows * F), 256, 0, st>>>(t_rowptr, t_row, t_pos, (const T*)val,
                                                                        (const T*)alpha, (const T*)Wx, (T*)y, rows * F,
                                                                        (int)N, (int)B, (int)F, nnz);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

template <typename T>
int attention_backward_t(const int32_t* rowptr, const int32_t* col, const void* val, const int32_t* erow,
                         const int32_t* t_rowptr, const int32_t* t_pos, const void* Wx, const void* s1, const void* s2, const void* alpha,
                         const void* dy, void* dWx, void* ds1, void* ds2, void* dz, int64_t Tn, int64_t N, int64_t B,
                         int64_t F, int64_t nnz, double slope, hipStream_t st) {
  GCRNN_PRE_LAUNCH();
  const int64_t rows = Tn * N * B;
  if (nnz > 0)
    edge_bwd_dot_kernel<T><<<blocks_for(Tn * nnz * B), 256, 0, st>>>(erow, col, (const T*)val, (const T*)Wx, (const T*)dy,
                                                                     (T*)dz, Tn * nnz * B, (int)N, (int)B, (int)F, nnz);
  edge_bwd_row_kernel<T><<<blocks_for(rows), 256, 0, st>>>(rowptr, col, (const T*)s1, (const T*)s2, (const T*)alpha,
                                                           (T*)dz, (T*)ds2, rows, (int)N, (int)B, nnz, (T)slope);
  edge_bwd_col_kernel<T><<<blocks_for(rows), 256, 0, st>>>(t_rowptr, t_pos, (const T*)dz, (T*)ds1, rows, (int)N, (int)B, nnz);
  edge_aggregate_kernel<T, true><<<blocks_for(rows * F), 256, 0, st>>>(rowptr, col, nullptr, (const T*)val, (const T*)alpha,
                                                                       (const T*)dy, (T*)dWx, rows * F, (int)N, (int)B,
                                                                       (int)F, nnz);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

bool attention_shape_ok(int64_t T, int64_t N, int64_t B, int64_t F, int64_t nnz) {
  if (T <= 0 || N <= 0 || B <= 0 || F <= 0 || nnz < 0) return false;
  if (N > 2147483647LL || B > 2147483647LL || F > 2147483647LL) return false;
  return cdiv(T * N * B * F, 256) <= 2147483647LL && cdiv(T * nnz * B, 256) <= 2147483647LL;
}

}  // namespace

extern "C" int gcrnn_attention_forward(int dtype, const int32_t* rowptr, const int32_t* col, const void* val,
                                       const int32_t* t_rowptr, const int32_t* t_row, const int32_t* t_pos, const void* Wx,
                                       const void* s1, const void* s2, void* alpha, void* y, int64_t T, int64_t N,
                                       int64_t B, int64_t F, int64_t nnz, double negative_slope, void* stream) {
  if (!rowptr || !t_rowptr || !Wx || !s1 || !s2 || !y) return GCRNN_ERR_NULL_POINTER;
  if (nnz > 0 && (!col || !val || !t_row || !t_pos || !alpha)) return GCRNN_ERR_NULL_POINTER;
  if (!attention_shape_ok(T, N, B, F, nnz)) return GCRNN_ERR_BAD_SHAPE;
  if (dtype == GCRNN_F32)
    return attention_forward_t<float>(rowptr, col, val, t_rowptr, t_row, t_pos, Wx, s1, s2, alpha, y, T, N, B, F, nnz,
                                      negative_slope, as_stream(stream));
  if (dtype == GCRNN_F64)
    return attention_forward_t<double>(rowptr, col, val, t_rowptr, t_row, t_pos, Wx, s1, s2, alpha, y, T, N, B, F, nnz,
                                       negative_slope, as_stream(stream));
  return GCRNN_ERR_BAD_DTYPE;
}

extern "C" int gcrnn_attention_backward(int dtype, const int32_t* rowptr, const int32_t* col, const void* val,
                                        const int32_t* edge_row, const int32_t* t_rowptr, const int32_t* t_pos, const void* Wx, const void* s1,
                                        const void* s2, const void* alpha, const void* dy, void* dWx, void* ds1, void* ds2,
                                        void* dz_scratch, int64_t T, int64_t N, int64_t B, int64_t F, int64_t nnz,
                                        double negative_slope, void* stream) {
  if (!rowptr || !t_rowptr || !Wx || !s1 || !s2 || !dy || !dWx || !ds1 || !ds2) return GCRNN_ERR_NULL_POINTER;
  if (nnz > 0 && (!col || !val || !edge_row || !t_pos || !alpha || !dz_scratch)) return GCRNN_ERR_NULL_POINTER;
  if (!attention_shape_ok(T, N, B, F, nnz)) return GCRNN_ERR_BAD_SHAPE;
  if (dtype == GCRNN_F32)
    return attention_backward_t<float>(rowptr, col, val, edge_row, t_rowptr, t_pos, Wx, s1, s2, alpha, dy, dWx, ds1, ds2, dz_scratch,
                                       T, N, B, F, nnz, negative_slope, as_stream(stream));
  if (dtype == GCRNN_F64)
    return attention_backward_t<double>(rowptr, col, val, edge_row, t_rowptr, t_pos, Wx, s1, s2, alpha, dy, dWx, ds1, ds2,
                                        dz_scratch, T, N, B, F, nnz, negative_slope, as_stream(stream));
  return GCRNN_ERR_BAD_DTYPE;
}
