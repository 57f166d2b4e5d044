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

// ---- bf16 activations (the fused cell's output), fp32 or bf16 parameters, fp32 accumulation ----------------------------------
// One thread per (r, pair of adjacent nodes): every row access is a 4-byte word, a wave moves 256 contiguous bytes.
__device__ __forceinline__ float hbf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t hf2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
template <typename W> __device__ __forceinline__ float wload(const W* p, int i);
template <> __device__ __forceinline__ float wload<float>(const float* p, int i) { return p[i]; }
template <> __device__ __forceinline__ float wload<uint16_t>(const uint16_t* p, int i) { return hbf2f(p[i]); }

constexpr int HEAD16_MAX_O = 2;      // the drivers' head has O = 1; the backward keeps O * F accumulators per thread

template <typename W>
__global__ __launch_bounds__(256) void node_linear16_fwd_kernel(const uint16_t* __restrict__ h, const W* __restrict__ w,
                                                                const W* __restrict__ b, uint16_t* __restrict__ y, int64_t R,
                                                                int N, int F, int O) {
  __shared__ float ws[HEAD16_MAX_O * HEAD_MAX_F + HEAD16_MAX_O];
  for (int i = threadIdx.x; i < O * F; i += 256) ws[i] = wload<W>(w, i);
  for (int i = threadIdx.x; i < O; i += 256) ws[O * F + i] = b ? wload<W>(b, i) : 0.f;
  __syncthreads();
  const int half = N >> 1;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * half) return;
  const int64_t r = idx / half;
  const int n = 2 * (int)(idx - r * half);
  const uint16_t* hp = h + r * F * N + n;
  float a0[HEAD16_MAX_O], a1[HEAD16_MAX_O];
#pragma unroll
  for (int o = 0; o < HEAD16_MAX_O; ++o) a0[o] = a1[o] = (o < O) ? ws[O * F + o] : 0.f;
#pragma unroll 8
  for (int f = 0; f < F; ++f) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(hp + (int64_t)f * N);
    const float h0 = hbf2f((uint16_t)(v & 0xffffu)), h1 = hbf2f((uint16_t)(v >> 16));
#pragma unroll
    for (int o = 0; o < HEAD16_MAX_O; ++o)
      if (o < O) { a0[o] += ws[o * F + f] * h0; a1[o] += ws[o * F + f] * h1; }
  }
#pragma unroll
  for (int o = 0; o < HEAD16_MAX_O; ++o)
    if (o < O) *reinterpret_cast<uint32_t*>(y + (r * O + o) * N + n) = (uint32_t)hf2bf(a0[o]) | ((uint32_t)hf2bf(a1[o]) << 16);
}

// grid (node tiles of 512, slabs of rows r): a thread walks its slab with the O * F weight-gradient accumulators in registers;
// one workgroup reduction per (o, f) at the end, partial sums per workgroup (no atomics; the caller adds them in fixed order)
template <typename W, int O>
__global__ __launch_bounds__(256) void node_linear16_bwd_kernel(const uint16_t* __restrict__ h, const W* __restrict__ w,
                                                                const uint16_t* __restrict__ dy, uint16_t* __restrict__ dh,
                                                                float* __restrict__ pw,   // [blocks][O][F]
                                                                float* __restrict__ pb,   // [blocks][O]
                                                                int64_t R, int N, int F, int per_slab) {
  __shared__ float ws[O * HEAD_MAX_F];
  __shared__ float red[4];
  for (int i = threadIdx.x; i < O * F; i += 256) ws[i] = wload<W>(w, i);
  __syncthreads();
  const int n = 2 * (blockIdx.x * 256 + threadIdx.x);
  const bool live = n < N;
  const int64_t r0 = (int64_t)blockIdx.y * per_slab, r1 = (r0 + per_slab < R) ? r0 + per_slab : R;
  float aw[O][HEAD_MAX_F], ab[O];
#pragma unroll
  for (int o = 0; o < O; ++o) {
    ab[o] = 0.f;
#pragma unroll
    for (int f = 0; f < HEAD_MAX_F; ++f) aw[o][f] = 0.f;
  }
  if (live)
    for (int64_t r = r0; r < r1; ++r) {
      float g0[O], g1[O];
#pragma unroll
      for (int o = 0; o < O; ++o) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(dy + (r * O + o) * N + n);
        g0[o] = hbf2f((uint16_t)(v & 0xffffu)); g1[o] = hbf2f((uint16_t)(v >> 16));
        ab[o] += g0[o] + g1[o];
      }
      const uint16_t* hp = h + r * F * N + n;
      uint16_t* dp = dh ? dh + r * F * N + n : nullptr;
#pragma unroll
      for (int f = 0; f < HEAD_MAX_F; ++f) {
        if (f < F) {
          const uint32_t v = *reinterpret_cast<const uint32_t*>(hp + (int64_t)f * N);
          const float h0 = hbf2f((uint16_t)(v & 0xffffu)), h1 = hbf2f((uint16_t)(v >> 16));
          float d0 = 0.f, d1 = 0.f;
#pragma unroll
          for (int o = 0; o < O; ++o) {
            d0 += ws[o * F + f] * g0[o]; d1 += ws[o * F + f] * g1[o];
            aw[o][f] += g0[o] * h0 + g1[o] * h1;
          }
          if (dp) *reinterpret_cast<uint32_t*>(dp + (int64_t)f * N) = (uint32_t)hf2bf(d0) | ((uint32_t)hf2bf(d1) << 16);
        }
      }
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
  for (int o = 0; o < O; ++o) {
#pragma unroll
    for (int f = 0; f <= HEAD_MAX_F; ++f) {          // f == HEAD_MAX_F: the bias gradient
      if (f < F || f == HEAD_MAX_F) {
        float v = (f == HEAD_MAX_F) ? ab[o] : aw[o][f < HEAD_MAX_F ? f : 0];
        for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
          const float t = red[0] + red[1] + red[2] + red[3];
          if (f == HEAD_MAX_F) pb[blk * O + o] = t; else pw[(blk * O + o) * F + f] = t;
        }
        __syncthreads();
      }
    }
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

// bf16 activations: h, dy, y, dh are bf16 (uint16) arrays; parameters fp32 (wdtype GCRNN_F32) or bf16; partial sums fp32.
// N even, O <= 2 (the drivers' head has one output per node), F <= 64.
extern "C" int gcrnn_node_linear_bf16_supported(int64_t N, int64_t F, int64_t O) {
  return (N > 0 && N % 2 == 0 && F > 0 && F <= HEAD_MAX_F && O > 0 && O <= HEAD16_MAX_O) ? 1 : 0;
}

extern "C" int gcrnn_node_linear_bf16_forward(int wdtype, const void* h, const void* w, const void* b, void* y, int64_t R,
                                              int64_t N, int64_t F, int64_t O, void* stream) {
  if (!h || !w || !y) return GCRNN_ERR_NULL_POINTER;
  if (R <= 0 || !gcrnn_node_linear_bf16_supported(N, F, O) || cdiv(R * (N / 2), 256) > 2147483647LL) return GCRNN_ERR_UNSUPPORTED;
  const unsigned nb = (unsigned)cdiv(R * (N / 2), 256);
  GCRNN_PRE_LAUNCH();
  if (wdtype == GCRNN_F32)
    node_linear16_fwd_kernel<float><<<nb, 256, 0, as_stream(stream)>>>((const uint16_t*)h, (const float*)w, (const float*)b, (uint16_t*)y, R, (int)N, (int)F, (int)O);
  else if (wdtype == GCRNN_BF16)
    node_linear16_fwd_kernel<uint16_t><<<nb, 256, 0, as_stream(stream)>>>((const uint16_t*)h, (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)y, R, (int)N, (int)F, (int)O);
  else
    return GCRNN_ERR_BAD_DTYPE;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// number of workgroups (= rows of the partial-sum arrays pw [blocks][O][F], pb [blocks][O]) of the backward launch
static void head16_grid(int64_t R, int64_t N, int64_t* tiles, int64_t* slabs, int64_t* per_slab) {
  *tiles = cdiv(N / 2, 256);
  int64_t want = cdiv(1024, *tiles);                       // ~4 workgroups per CU
  if (want > R) want = R;
  *per_slab = cdiv(R, want);
  *slabs = cdiv(R, *per_slab);
}
extern "C" int64_t gcrnn_node_linear_bf16_blocks(int64_t R, int64_t N) {
  int64_t t, s, p;
  head16_grid(R > 0 ? R : 1, N > 1 ? N : 2, &t, &s, &p);
  return t * s;
}

extern "C" int gcrnn_node_linear_bf16_backward(int wdtype, const void* h, const void* w, const void* dy, void* dh, float* pw,
                                               float* pb, int64_t R, int64_t N, int64_t F, int64_t O, void* stream) {
  if (!h || !w || !dy || !pw || !pb) return GCRNN_ERR_NULL_POINTER;
  if (R <= 0 || !gcrnn_node_linear_bf16_supported(N, F, O)) return GCRNN_ERR_UNSUPPORTED;
  int64_t tiles, slabs, per_slab;
  head16_grid(R, N, &tiles, &slabs, &per_slab);
  if (slabs > 65535) return GCRNN_ERR_UNSUPPORTED;
  dim3 grid((unsigned)tiles, (unsigned)slabs);
  GCRNN_PRE_LAUNCH();
#define GCRNN_HEAD16(WT, OO) \
  node_linear16_bwd_kernel<WT, OO><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)h, (const WT*)w, (const uint16_t*)dy, \
                                                                         (uint16_t*)dh, pw, pb, R, (int)N, (int)F, (int)per_slab)
  if (wdtype == GCRNN_F32) { if (O == 1) GCRNN_HEAD16(float, 1); else GCRNN_HEAD16(float, 2); }
  else if (wdtype == GCRNN_BF16) { if (O == 1) GCRNN_HEAD16(uint16_t, 1); else GCRNN_HEAD16(uint16_t, 2); }
  else return GCRNN_ERR_BAD_DTYPE;
#undef GCRNN_HEAD16
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}
