// Per-node perceptron head of GatedGCRNNforRegression, mlpType = 'multipMlp' (reference architectures.py:1616-1627: the SAME
// Linear(F -> O) applied to every node's state, written there as a Python loop over the N nodes), directly on the user
// layout: H [R][F][N] (R = B * T; node-contiguous rows) -> Y [R][O][N]. No transposes, no [R*N][F] copy; the reduction
// over f streams F node-contiguous rows.
//   forward   y[r][o][n] = sum_f w[o][f] h[r][f][n] + b[o]
//   backward  dh[r][f][n] = sum_o w[o][f] dy[r][o][n];   dw[o][f] = sum_{r,n} dy[r][o][n] h[r][f][n];   db[o] = sum dy[r][o][n]
// dw / db: per-workgroup partial sums (no atomics), added by the caller in a fixed order.
#include "gcrnn_common.h"

namespace {

constexpr int HEAD_MAX_F = 64, HEAD_MAX_O = 8;

// one thread per (r, n): the thread's F inputs stay in registers for all O outputs
template <typename T>
__global__ __launch_bounds__(256) void node_linear_fwd_kernel(const T* __restrict__ h, const T* __restrict__ w,
                                                              const T* __restrict__ b, T* __restrict__ y, int64_t R, int N,
                                                              int F, int O) {
  __shared__ T ws[HEAD_MAX_O * HEAD_MAX_F + HEAD_MAX_O];
  for (int i = threadIdx.x; i < O * F; i += 256) ws[i] = w[i];
  for (int i = threadIdx.x; i < O; i += 256) ws[O * F + i] = b ? b[i] : T(0);
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * N) return;
  const int64_t r = idx / N;
  const int n = (int)(idx - r * N);
  const T* hp = h + r * F * N + n;
  T acc[HEAD_MAX_O];
#pragma unroll
  for (int o = 0; o < HEAD_MAX_O; ++o) acc[o] = (o < O) ? ws[O * F + o] : T(0);
  for (int f = 0; f < F; ++f) {
    const T hv = hp[(int64_t)f * N];
#pragma unroll
    for (int o = 0; o < HEAD_MAX_O; ++o)
      if (o < O) acc[o] += ws[o * F + f] * hv;
  }
  T* yp = y + r * O * N + n;
#pragma unroll
  for (int o = 0; o < HEAD_MAX_O; ++o)
    if (o < O) yp[(int64_t)o * N] = acc[o];
}

// one thread per (r, n) again: dh for all f, and the thread's contribution to dw / db, reduced over the workgroup
template <typename T>
__global__ __launch_bounds__(256) void node_linear_bwd_kernel(const T* __restrict__ h, const T* __restrict__ w,
                                                              const T* __restrict__ dy, T* __restrict__ dh,
                                                              T* __restrict__ pw,   // [blocks][O][F]
                                                              T* __restrict__ pb,   // [blocks][O]
                                                              int64_t R, int N, int F, int O) {
  __shared__ T ws[HEAD_MAX_O * HEAD_MAX_F];
  __shared__ T red[4];
  for (int i = threadIdx.x; i < O * F; i += 256) ws[i] = w[i];
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < R * N;
  const int64_t r = live ? idx / N : 0;
  const int n = live ? (int)(idx - r * N) : 0;
  T g[HEAD_MAX_O];
#pragma unroll
  for (int o = 0; o < HEAD_MAX_O; ++o) g[o] = (live && o < O) ? dy[(r * O + o) * N + n] : T(0);
  const T* hp = h + r * F * N + n;
  T* dp = dh ? dh + r * F * N + n : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int f = 0; f < F; ++f) {
    const T hv = live ? hp[(int64_t)f * N] : T(0);
    T d = T(0);
#pragma unroll
    for (int o = 0; o < HEAD_MAX_O; ++o)
      if (o < O) d += ws[o * F + f] * g[o];
    if (live && dp) dp[(int64_t)f * N] = d;
    for (int o = 0; o < O; ++o) {                    // dw[o][f]: block reduction of g[o] * hv
      T v = g[o] * hv;
      for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
      if (lane == 0) red[wave] = v;
      __syncthreads();
      if (threadIdx.x == 0) pw[((int64_t)blockIdx.x * O + o) * F + f] = red[0] + red[1] + red[2] + red[3];
      __syncthreads();
    }
  }
  for (int o = 0; o < O; ++o) {
    T v = g[o];
    for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) pb[(int64_t)blockIdx.x * O + o] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
}

bool head_shape_ok(int64_t R, int64_t N, int64_t F, int64_t O) {
  return R > 0 && N > 0 && F > 0 && O > 0 && F <= HEAD_MAX_F && O <= HEAD_MAX_O && N <= 2147483647LL &&
         cdiv(R * N, 256) <= 2147483647LL;
}

}  // namespace

extern "C" int64_t gcrnn_node_linear_blocks(int64_t R, int64_t N) { return cdiv((R > 0 ? R : 1) * (N > 0 ? N : 1), 256); }

extern "C" int gcrnn_node_linear_forward(int dtype, const void* h, const void* w, const void* b, void* y, int64_t R, int64_t N,
                                         int64_t F, int64_t O, void* stream) {
  if (!h || !w || !y) return GCRNN_ERR_NULL_POINTER;
  if (!head_shape_ok(R, N, F, O)) return GCRNN_ERR_UNSUPPORTED;
  const unsigned nb = (unsigned)gcrnn_node_linear_blocks(R, N);
  GCRNN_PRE_LAUNCH();
  if (dtype == GCRNN_F32)
    node_linear_fwd_kernel<float><<<nb, 256, 0, as_stream(stream)>>>((const float*)h, (const float*)w, (const float*)b, (float*)y, R, (int)N, (int)F, (int)O);
  else if (dtype == GCRNN_F64)
    node_linear_fwd_kernel<double><<<nb, 256, 0, as_stream(stream)>>>((const double*)h, (const double*)w, (const double*)b, (double*)y, R, (int)N, (int)F, (int)O);
  else
    return GCRNN_ERR_BAD_DTYPE;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_node_linear_backward(int dtype, const void* h, const void* w, const void* dy, void* dh, void* pw, void* pb,
                                          int64_t R, int64_t N, int64_t F, int64_t O, void* stream) {
  if (!h || !w || !dy || !pw || !pb) return GCRNN_ERR_NULL_POINTER;
  if (!head_shape_ok(R, N, F, O)) return GCRNN_ERR_UNSUPPORTED;
  const unsigned nb = (unsigned)gcrnn_node_linear_blocks(R, N);
  GCRNN_PRE_LAUNCH();
  if (dtype == GCRNN_F32)
    node_linear_bwd_kernel<float><<<nb, 256, 0, as_stream(stream)>>>((const float*)h, (const float*)w, (const float*)dy, (float*)dh, (float*)pw, (float*)pb, R, (int)N, (int)F, (int)O);
  else if (dtype == GCRNN_F64)
    node_linear_bwd_kernel<double><<<nb, 256, 0, as_stream(stream)>>>((const double*)h, (const double*)w, (const double*)dy, (double*)dh, (double*)pw, (double*)pb, R, (int)N, (int)F, (int)O);
  else
    return GCRNN_ERR_BAD_DTYPE;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}
