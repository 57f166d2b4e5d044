// Training-loop loss of the k-step prediction drivers: batchTimeL1Loss (reference Utils/miscTools.py:112-119 =
// nn.L1Loss, mean |x - y| over every entry) as ONE pass that also emits the gradient. PyTorch's autograd needs
// sub, abs, mean forward and sign, scale backward -- five passes over tensors the size of the whole state sequence.
//   partial[block] = sum over the block's elements of |x - y|        (summed by the caller in a fixed order)
//   grad[i]        = sign(x[i] - y[i]) * inv_n                        (optional; sign(0) = 0 like torch.sign)
#include "gcrnn_common.h"

namespace {

template <typename T> struct LossAcc { typedef float type; };
template <> struct LossAcc<double> { typedef double type; };

__device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ double ld(const double* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ld(const uint16_t* p, int64_t i) { return __uint_as_float((uint32_t)p[i] << 16); }
__device__ __forceinline__ void st(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st(double* p, int64_t i, double v) { p[i] = v; }
__device__ __forceinline__ void st(uint16_t* p, int64_t i, float v) {
  uint32_t a = __float_as_uint(v);
  a += 0x7fffu + ((a >> 16) & 1u);
  p[i] = (uint16_t)(a >> 16);
}

// 256 threads, ELEMS consecutive elements per thread and trip (16 bytes for every dtype), grid-stride over chunks
template <typename T, typename A, int ELEMS>
__global__ __launch_bounds__(256) void l1_loss_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ grad,
                                                      A* __restrict__ partial, int64_t n, A inv_n) {
  __shared__ A red[4];
  A acc = A(0);
  const int64_t stride = (int64_t)gridDim.x * 256 * ELEMS;
  int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ELEMS;
  // two independent 16-byte chunks per thread and trip while both are whole: twice the loads in flight
  for (; base + stride + ELEMS <= n; base += 2 * stride) {
    T xa[ELEMS], ya[ELEMS], xb[ELEMS], yb[ELEMS], ga[ELEMS], gb[ELEMS];
    *reinterpret_cast<uint4*>(xa) = *reinterpret_cast<const uint4*>(x + base);
    *reinterpret_cast<uint4*>(ya) = *reinterpret_cast<const uint4*>(y + base);
    *reinterpret_cast<uint4*>(xb) = *reinterpret_cast<const uint4*>(x + base + stride);
    *reinterpret_cast<uint4*>(yb) = *reinterpret_cast<const uint4*>(y + base + stride);
#pragma unroll
    for (int e = 0; e < ELEMS; ++e) {
      const A da = (A)ld(xa, e) - (A)ld(ya, e), db = (A)ld(xb, e) - (A)ld(yb, e);
      acc += (da < A(0) ? -da : da) + (db < A(0) ? -db : db);
      if (grad) {
        st(ga, e, da > A(0) ? inv_n : (da < A(0) ? -inv_n : A(0)));
        st(gb, e, db > A(0) ? inv_n : (db < A(0) ? -inv_n : A(0)));
      }
    }
    if (grad) {
      *reinterpret_cast<uint4*>(grad + base) = *reinterpret_cast<const uint4*>(ga);
      *reinterpret_cast<uint4*>(grad + base + stride) = *reinterpret_cast<const uint4*>(gb);
    }
  }
  for (; base < n; base += stride) {
    if (base + ELEMS <= n) {
      T xv[ELEMS], yv[ELEMS];
      *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(x + base);
      *reinterpret_cast<uint4*>(yv) = *reinterpret_cast<const uint4*>(y + base);
      T gv[ELEMS];
#pragma unroll
      for (int e = 0; e < ELEMS; ++e) {
        const A d = (A)ld(xv, e) - (A)ld(yv, e);
        acc += d < A(0) ? -d : d;
        if (grad) st(gv, e, d > A(0) ? inv_n : (d < A(0) ? -inv_n : A(0)));
      }
      if (grad) *reinterpret_cast<uint4*>(grad + base) = *reinterpret_cast<const uint4*>(gv);
    } else {
      for (int64_t i = base; i < n; ++i) {
        const A d = (A)ld(x, i) - (A)ld(y, i);
        acc += d < A(0) ? -d : d;
        if (grad) st(grad, i, d > A(0) ? inv_n : (d < A(0) ? -inv_n : A(0)));
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T, typename A, int ELEMS>
int l1_launch(const void* x, const void* y, void* grad, void* partial, int64_t n, int64_t nblocks, double inv_n,
              hipStream_t st_) {
  GCRNN_PRE_LAUNCH();
  l1_loss_kernel<T, A, ELEMS><<<(unsigned)nblocks, 256, 0, st_>>>((const T*)x, (const T*)y, (T*)grad, (A*)partial, n, (A)inv_n);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// data[i] *= r[0] unless r[0] == 1 (then every workgroup returns after one scalar load): the upstream gradient of a scalar loss
// is almost always exactly 1, and a full pass over a gradient the size of the state sequence just to multiply by it is 3 % of a
// training step. r is a device scalar of the accumulation type.
template <typename T, typename A, int ELEMS>
__global__ __launch_bounds__(256) void scale_unless_one_kernel(T* __restrict__ data, const A* __restrict__ r, int64_t n) {
  const A rv = r[0];
  if (rv == A(1)) return;
  const int64_t stride = (int64_t)gridDim.x * 256 * ELEMS;
  for (int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ELEMS; base < n; base += stride) {
    if (base + ELEMS <= n) {
      T v[ELEMS];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(data + base);
#pragma unroll
      for (int e = 0; e < ELEMS; ++e) st(v, e, (A)ld(v, e) * rv);
      *reinterpret_cast<uint4*>(data + base) = *reinterpret_cast<const uint4*>(v);
    } else {
      for (int64_t i = base; i < n; ++i) st(data, i, (A)ld(data, i) * rv);
    }
  }
}

template <typename T, typename A, int ELEMS>
int scale_launch(void* data, const void* r, int64_t n, int64_t nblocks, hipStream_t st_) {
  GCRNN_PRE_LAUNCH();
  scale_unless_one_kernel<T, A, ELEMS><<<(unsigned)nblocks, 256, 0, st_>>>((T*)data, (const A*)r, n);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

}  // namespace

// Number of partial sums (= workgroups) gcrnn_l1_loss writes for n elements.
extern "C" int64_t gcrnn_l1_loss_blocks(int64_t n) {
  const int64_t chunks = cdiv(n > 0 ? n : 1, 256 * 8);
  return chunks < 2048 ? chunks : 2048;
}

extern "C" int gcrnn_l1_loss(int dtype, const void* x, const void* y, void* grad, void* partial, int64_t n, double inv_n,
                             void* stream) {
  if (!x || !y || !partial) return GCRNN_ERR_NULL_POINTER;
  if (n <= 0) return GCRNN_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(grad)) & 15)
    return GCRNN_ERR_UNSUPPORTED;                          // 16-byte vector accesses
  const int64_t nb = gcrnn_l1_loss_blocks(n);
  if (dtype == GCRNN_F32) return l1_launch<float, float, 4>(x, y, grad, partial, n, nb, inv_n, as_stream(stream));
  if (dtype == GCRNN_F64) return l1_launch<double, double, 2>(x, y, grad, partial, n, nb, inv_n, as_stream(stream));
  if (dtype == GCRNN_BF16) return l1_launch<uint16_t, float, 8>(x, y, grad, partial, n, nb, inv_n, as_stream(stream));
  return GCRNN_ERR_BAD_DTYPE;
}

// data[i] *= r[0] in place, skipped entirely when the device scalar r[0] (fp32; fp64 for fp64 data) equals 1: the backward of a
// scalar loss whose gradient tensor gcrnn_l1_loss has already written (d loss / d x = upstream * sign(x - y) / n, miscTools.py:112-119
// under autograd) without a second pass over it in the common case upstream == 1.
extern "C" int gcrnn_scale_unless_one(int dtype, void* data, const void* r, int64_t n, void* stream) {
  if (!data || !r) return GCRNN_ERR_NULL_POINTER;
  if (n <= 0) return GCRNN_ERR_BAD_SHAPE;
  if (reinterpret_cast<uintptr_t>(data) & 15) return GCRNN_ERR_UNSUPPORTED;
  const int64_t nb = gcrnn_l1_loss_blocks(n);
  if (dtype == GCRNN_F32) return scale_launch<float, float, 4>(data, r, n, nb, as_stream(stream));
  if (dtype == GCRNN_F64) return scale_launch<double, double, 2>(data, r, n, nb, as_stream(stream));
  if (dtype == GCRNN_BF16) return scale_launch<uint16_t, float, 8>(data, r, n, nb, as_stream(stream));
  return GCRNN_ERR_BAD_DTYPE;
}
