// Shared pieces of the small-graph matrix-core kernels (gcrnn_small_mfma.hip, gcrnn_small_gates.hip): MFMA traits for
// fp64 / fp32, the conflict-free LDS row stride and the batched branch-free tile multiply-accumulate.
#pragma once
#include "gcrnn_common.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mf;
template <> struct Mf<double> {
  typedef d4 acc;
  static __device__ __forceinline__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
  // tanh(v) = sign(v) (1 - t) / (1 + t), t = exp(-2 |v|): no overflow, absolute error ~1e-16 -- a third of the
  // instructions of the library tanh, which sits on the critical path of every time step
  static __device__ __forceinline__ double tanh_(double v) {
    const double t = exp(-2.0 * fabs(v));
    return copysign((1.0 - t) / (1.0 + t), v);
  }
};
template <> struct Mf<float> {
  typedef f4 acc;
  static __device__ __forceinline__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
  static __device__ __forceinline__ float tanh_(float v) { return tanhf(v); }
};

// smallest stride >= n (elements) whose byte size is an odd multiple of 16 modulo 256 and a multiple of 4 elements' worth
// of padding beyond the last k-step: n4 = n rounded up to 4 is covered.
template <typename T>
__host__ __device__ inline int lds_stride(int n) {
  int s = (n + 3) & ~3;
  const int unit = 16 / (int)sizeof(T);            // elements per 16 bytes: 2 (fp64) or 4 (fp32)
  while (((s / unit) & 1) == 0 || s % unit) ++s;   // s * sizeof(T) = 16 * odd
  return s;
}

// acc += sum over `ksteps` k-steps of A(i, k) B(k, j): ap / bp point at this lane's element of k-step 0 and advance by
// astep / bstep elements per k-step. Branch-free (masked lanes point at a row of zeros) and batched by four so that
// eight LDS reads are in flight before the dependent MFMA chain consumes them.
template <typename T>
__device__ __forceinline__ typename Mf<T>::acc tile_mac(typename Mf<T>::acc acc, const T* ap, int astep, const T* bp,
                                                        int bstep, int ksteps, T ascale = T(1), T bscale = T(1)) {
  int s = 0;
  for (; s + 4 <= ksteps; s += 4) {
    const T a0 = ap[0], a1 = ap[astep], a2 = ap[2 * astep], a3 = ap[3 * astep];
    const T b0 = bp[0], b1 = bp[bstep], b2 = bp[2 * bstep], b3 = bp[3 * bstep];
    ap += 4 * astep; bp += 4 * bstep;
    acc = Mf<T>::mma(a0 * ascale, b0 * bscale, acc);
    acc = Mf<T>::mma(a1 * ascale, b1 * bscale, acc);
    acc = Mf<T>::mma(a2 * ascale, b2 * bscale, acc);
    acc = Mf<T>::mma(a3 * ascale, b3 * bscale, acc);
  }
  for (; s < ksteps; ++s) {
    acc = Mf<T>::mma(ap[0] * ascale, bp[0] * bscale, acc);
    ap += astep; bp += bstep;
  }
  return acc;
}

// acc += sum_k A(i, k) (B(k, j) g(k)): both operands step by 4 elements per k-step (transposed B), and the B operand is
// scaled by a per-k gate g read through gp (gp steps with k too). `bfirst`: the first pointer is the B operand.
template <typename T>
__device__ __forceinline__ typename Mf<T>::acc tile_mac_gated(typename Mf<T>::acc acc, const T* bp, const T* gp, const T* ap,
                                                              int astep, int ksteps, bool bfirst) {
  (void)bfirst;
  int s = 0;
  for (; s + 4 <= ksteps; s += 4) {
    const T b0 = bp[0] * gp[0], b1 = bp[4] * gp[4], b2 = bp[8] * gp[8], b3 = bp[12] * gp[12];
    const T a0 = ap[0], a1 = ap[astep], a2 = ap[2 * astep], a3 = ap[3 * astep];
    bp += 16; gp += 16; ap += 4 * astep;
    acc = Mf<T>::mma(a0, b0, acc);
    acc = Mf<T>::mma(a1, b1, acc);
    acc = Mf<T>::mma(a2, b2, acc);
    acc = Mf<T>::mma(a3, b3, acc);
  }
  for (; s < ksteps; ++s) {
    acc = Mf<T>::mma(ap[0], bp[0] * gp[0], acc);
    bp += 4; gp += 4; ap += astep;
  }
  return acc;
}

}  // namespace
