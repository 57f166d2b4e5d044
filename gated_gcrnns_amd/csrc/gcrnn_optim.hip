// Optimiser step and metric of the k-step-prediction training loop as single passes.
//
// gcrnn_adam_flat: torch.optim.Adam (the drivers' optimiser: kStepPredGRNNs.py:158-161, train_rnn.py:276) over ONE flat
//   parameter buffer / ONE flat gradient buffer -- the buffer the data-parallel all-reduce has just reduced
//   (parallel.FlatGradAllReduce) -- instead of ~10 launches per parameter tensor:
//       m = b1 m + (1 - b1) g;   v = b2 v + (1 - b2) g^2;   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//   t is read from a DEVICE counter that the call increments first (hipGraph-capturable: no host value baked in).
// gcrnn_batch_time_mse: the drivers' metric batchTimeMSELoss (Utils/miscTools.py:121-130): for x, y as [R][C] matrices
//   (R = batch * time rows, C = N * F columns)  mean_c sqrt(sum_r (x - y)^2) / sqrt(sum_r y^2)  in two launches
//   (column partial sums over row slabs, then a fixed-order finish) instead of seven torch passes.
#include "gcrnn_common.h"

namespace {

__global__ void adam_tick_kernel(int64_t* step) { step[0] += 1; }

template <typename T>
__global__ __launch_bounds__(256) void adam_flat_kernel(T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v,
                                                        int64_t n, T lr, T b1, T b2, T eps, T gscale, const int64_t* __restrict__ step) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double t = (double)step[0];
  const T bc1 = (T)(1.0 - pow((double)b1, t)), bc2s = (T)sqrt(1.0 - pow((double)b2, t));
  const T gi = g[i] * gscale;
  const T mi = m[i] + (gi - m[i]) * (T(1) - b1);          // exp_avg.lerp_(grad, 1 - beta1)
  const T vi = v[i] * b2 + (T(1) - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const T denom = (T)sqrt((double)vi) / bc2s + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

// column partial sums of (x - y)^2 and y^2 over a slab of rows; thread = column (coalesced along the row)
template <typename T>
__device__ __forceinline__ double ldv(const T* p, int64_t i) { return (double)p[i]; }
template <>
__device__ __forceinline__ double ldv<uint16_t>(const uint16_t* p, int64_t i) { return (double)__uint_as_float((uint32_t)p[i] << 16); }

template <typename T, typename A>
__global__ __launch_bounds__(256) void mse_partial_kernel(const T* __restrict__ x, const T* __restrict__ y, A* __restrict__ part,
                                                          int64_t R, int64_t C, int64_t rows_per_slab) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = (r0 + rows_per_slab < R) ? r0 + rows_per_slab : R;
  A se = A(0), sy = A(0);
  for (int64_t r = r0; r < r1; ++r) {
    const A xv = (A)ldv<T>(x, r * C + c), yv = (A)ldv<T>(y, r * C + c);
    se += (xv - yv) * (xv - yv);
    sy += yv * yv;
  }
  part[((int64_t)blockIdx.y * 2 + 0) * C + c] = se;
  part[((int64_t)blockIdx.y * 2 + 1) * C + c] = sy;
}

template <typename A>
__global__ __launch_bounds__(256) void mse_finish_kernel(const A* __restrict__ part, A* __restrict__ out, int64_t C, int64_t slabs) {
  __shared__ A red[256];
  A acc = A(0);
  for (int64_t c = threadIdx.x; c < C; c += 256) {
    A se = A(0), sy = A(0);
    for (int64_t sl = 0; sl < slabs; ++sl) { se += part[(sl * 2 + 0) * C + c]; sy += part[(sl * 2 + 1) * C + c]; }
    acc += (A)sqrt((double)se) / (A)sqrt((double)sy);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (A)C;
}

}  // namespace

extern "C" int gcrnn_adam_flat(int dtype, void* p, const void* g, void* m, void* v, int64_t n, double lr, double beta1, double beta2,
                               double eps, double grad_scale, int64_t* step_dev, void* stream) {
  if (!p || !g || !m || !v || !step_dev) return GCRNN_ERR_NULL_POINTER;
  if (n <= 0) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
  GCRNN_PRE_LAUNCH();
  adam_tick_kernel<<<1, 1, 0, st>>>(step_dev);
  if (dtype == GCRNN_F32)
    adam_flat_kernel<float><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((float*)p, (const float*)g, (float*)m, (float*)v, n, (float)lr, (float)beta1,
                                                                      (float)beta2, (float)eps, (float)grad_scale, step_dev);
  else if (dtype == GCRNN_F64)
    adam_flat_kernel<double><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((double*)p, (const double*)g, (double*)m, (double*)v, n, lr, beta1, beta2,
                                                                       eps, grad_scale, step_dev);
  else
    return GCRNN_ERR_BAD_DTYPE;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// Row slabs of the metric's first pass for an [R][C] problem (part: [slabs][2][C] accumulators, fp64 for F64 else fp32).
extern "C" int64_t gcrnn_batch_time_mse_slabs(int64_t R, int64_t C) {
  const int64_t colblocks = cdiv(C > 0 ? C : 1, 256);
  int64_t slabs = cdiv(1024, colblocks);                       // ~1024 workgroups
  if (slabs > R) slabs = R > 0 ? R : 1;
  if (slabs > 4096) slabs = 4096;
  return slabs;
}

extern "C" int gcrnn_batch_time_mse(int dtype, const void* x, const void* y, void* part, void* out, int64_t R, int64_t C, void* stream) {
  if (!x || !y || !part || !out) return GCRNN_ERR_NULL_POINTER;
  if (R <= 0 || C <= 0) return GCRNN_ERR_BAD_SHAPE;
  const int64_t slabs = gcrnn_batch_time_mse_slabs(R, C), rps = cdiv(R, slabs);
  const dim3 grid((unsigned)cdiv(C, 256), (unsigned)cdiv(R, rps));
  hipStream_t st = as_stream(stream);
  GCRNN_PRE_LAUNCH();
  if (dtype == GCRNN_F64) {
    mse_partial_kernel<double, double><<<grid, 256, 0, st>>>((const double*)x, (const double*)y, (double*)part, R, C, rps);
    mse_finish_kernel<double><<<1, 256, 0, st>>>((const double*)part, (double*)out, C, (int64_t)grid.y);
  } else if (dtype == GCRNN_F32) {
    mse_partial_kernel<float, float><<<grid, 256, 0, st>>>((const float*)x, (const float*)y, (float*)part, R, C, rps);
    mse_finish_kernel<float><<<1, 256, 0, st>>>((const float*)part, (float*)out, C, (int64_t)grid.y);
  } else if (dtype == GCRNN_BF16) {
    mse_partial_kernel<uint16_t, float><<<grid, 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)y, (float*)part, R, C, rps);
    mse_finish_kernel<float><<<1, 256, 0, st>>>((const float*)part, (float*)out, C, (int64_t)grid.y);
  } else {
    return GCRNN_ERR_BAD_DTYPE;
  }
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}
