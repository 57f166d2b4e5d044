// Filter taps of a Horner-form GCRNN step on the fp32 / fp64 matrix cores (exact IEEE products and sums, no reduced-precision
// operands): the 1e-5 (fp32) / 1e-11 (fp64) parity modes of the streaming path (GGCRNNCell._forward_horner).
//     u_k[r][:] = z_h[r][:] B_k^T + z_x[r][:] A_k^T       k = blockIdx.y,   r = (node, sequence) row of the node-major layout
// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64:  D^T tile [16 f x 16 rows] += W_k [16 f x 4 c] * z^T [4 c x 16 rows].
// The contraction index may be visited in any order as long as both operands agree, so a lane's B operands are simply the
// elements of its 16-byte row loads (lane (row, q) holds c = 4 V j + V q + r of load j, V = elements per 16 bytes) and the
// A operands are gathered once per workgroup in that same order and stay in registers (one tap per workgroup: K x fewer
// weight registers, the rows are re-read per tap from L2 / Infinity Cache). fp32 D leaves a lane with 4 consecutive output
// features of one row (one 16-byte store); the fp64 D layout (row = lane/16 + 4 r) is made contiguous by permuting which
// output feature each A row carries. Replaces the library GEMMs of round 1 on this path (reference: graphML.py:134-135).
#include "gcrnn_common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) double f64x4;

namespace {

template <typename T> struct Mf;
template <> struct Mf<float> {
  static constexpr int V = 4;
  typedef f32x4 acc_t;
  __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ int feat(int m) { return m; }                   // D row m*? : lane fq holds rows 4 fq + r
};
template <> struct Mf<double> {
  static constexpr int V = 2;
  typedef f64x4 acc_t;
  __device__ static __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ int feat(int m) { return 4 * (m & 3) + (m >> 2); }   // lane fq holds rows fq + 4 r -> features 4 fq + r
};

// FT: 16-feature output tiles per wave pass (blockIdx.z walks F / (16 FT)); NL: 16-byte loads per row over [z_h | z_x].
template <typename T, int FT, int NL>
__global__ __launch_bounds__(256) void taps_mfma_kernel(const T* __restrict__ zh, const T* __restrict__ zx, const T* __restrict__ wB,
                                                        const T* __restrict__ wA, T* __restrict__ out0, T* __restrict__ outrest,
                                                        int64_t R, int F, int Ch, int Cx, int Kst, int Kin) {
  constexpr int V = Mf<T>::V;
  typedef typename Mf<T>::acc_t acc_t;
  const int tap = blockIdx.y, f0 = blockIdx.z * (16 * FT);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int nlh = Ch / (4 * V);                      // loads that come from z_h
  // A operands of this (tap, feature block), in the order the row loads deliver the contraction index
  T afr[FT][NL][V];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    const int f = f0 + 16 * ft + Mf<T>::feat(m);
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
      for (int r = 0; r < V; ++r) {
        const int c = 4 * V * j + V * q + r;           // index into [z_h | z_x]
        T w = T(0);
        if (j < nlh) { if (tap < Kst) w = wB[((int64_t)f * Kst + tap) * Ch + c]; }
        else         { if (tap < Kin) w = wA[((int64_t)f * Kin + tap) * Cx + (c - Ch)]; }
        afr[ft][j][r] = w;
      }
  }
  T* dst = (tap == 0) ? out0 : outrest + (int64_t)(tap - 1) * R * F;
  const int64_t tiles = (R + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row = tile * 16 + m;
    const bool ok = row < R;
    uint4 zv[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      zv[j] = uint4{0u, 0u, 0u, 0u};
      if (ok) zv[j] = (j < nlh) ? *reinterpret_cast<const uint4*>(zh + row * Ch + 4 * V * j + V * q)
                                : *reinterpret_cast<const uint4*>(zx + row * Cx + 4 * V * (j - nlh) + V * q);
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      acc_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const T* ze = reinterpret_cast<const T*>(&zv[j]);
#pragma unroll
        for (int r = 0; r < V; ++r) acc = Mf<T>::mfma(afr[ft][j][r], ze[r], acc);
      }
      if (ok) {
        T* p = dst + row * F + f0 + 16 * ft + 4 * q;       // 4 consecutive output features of this row
        if (V == 4) {
          *reinterpret_cast<f32x4*>(p) = *reinterpret_cast<const f32x4*>(&acc);
        } else {
          const double* a = reinterpret_cast<const double*>(&acc);
          *reinterpret_cast<double2*>(p) = double2{a[0], a[1]};
          *reinterpret_cast<double2*>(p + 2) = double2{a[2], a[3]};
        }
      }
    }
  }
}

template <typename T, int FT, int NL>
int launch_nl(const void* zh, const void* zx, const void* wB, const void* wA, void* out0, void* outrest, int64_t R, int F, int Ch, int Cx,
              int Kst, int Kin, hipStream_t st) {
  const int K = Kst > Kin ? Kst : Kin;
  const int64_t tiles = (R + 15) / 16;
  int64_t gx = (tiles + 3) / 4;
  // grid-stride over the row tiles: a workgroup fills its weight registers once (FT * NL * V scattered 4 / 8-byte loads per
  // lane) and must amortise that over many tiles -- about two workgroups per CU in all, K * (F / 16 FT) of them per x index
  int64_t cap = (2 * 256) / ((int64_t)K * (F / (16 * FT)));
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  GCRNN_PRE_LAUNCH();
  taps_mfma_kernel<T, FT, NL><<<dim3((unsigned)gx, (unsigned)K, (unsigned)(F / (16 * FT))), 256, 0, st>>>(
      (const T*)zh, (const T*)zx, (const T*)wB, (const T*)wA, (T*)out0, (T*)outrest, R, F, Ch, Cx, Kst, Kin);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

template <typename T>
int launch_t(const void* zh, const void* zx, const void* wB, const void* wA, void* out0, void* outrest, int64_t R, int F, int Ch, int Cx,
             int Kst, int Kin, hipStream_t st) {
  constexpr int V = Mf<T>::V;
  const int NL = (Ch + Cx) / (4 * V);
  // weight registers per lane = FT * NL * V (* 2 for fp64): keep them under ~128
  const int ft = (F % 32 == 0 && NL * V * (sizeof(T) / 4) * 2 <= 128) ? 2 : 1;
#define GCRNN_TM_CASE(FTV, NLV) \
  if (ft == FTV && NL == NLV) return launch_nl<T, FTV, NLV>(zh, zx, wB, wA, out0, outrest, R, F, Ch, Cx, Kst, Kin, st);
  GCRNN_TM_CASE(2, 1) GCRNN_TM_CASE(2, 2) GCRNN_TM_CASE(2, 4) GCRNN_TM_CASE(2, 8) GCRNN_TM_CASE(2, 16)
  GCRNN_TM_CASE(1, 1) GCRNN_TM_CASE(1, 2) GCRNN_TM_CASE(1, 4) GCRNN_TM_CASE(1, 8) GCRNN_TM_CASE(1, 16)
#undef GCRNN_TM_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

}  // namespace

// F % 16 == 0; Ch, Cx multiples of 16 (fp32) / 8 (fp64) elements with Ch + Cx in {1, 2, 4, 8, 16} x that; Cx may be 0.
extern "C" int gcrnn_taps_mfma_supported(int dtype, int64_t F, int64_t Ch, int64_t Cx) {
  const int V = dtype == GCRNN_F32 ? 4 : (dtype == GCRNN_F64 ? 2 : 0);
  if (!V || F <= 0 || F % 16 || Ch <= 0 || Cx < 0 || Ch % (4 * V) || Cx % (4 * V)) return 0;
  const int64_t nl = (Ch + Cx) / (4 * V);
  return nl == 1 || nl == 2 || nl == 4 || nl == 8 || nl == 16;
}

// zh [R][Ch], zx [R][Cx] (null when Cx == 0); wB [F][Kst][Ch], wA [F][Kin][Cx] (the reference's F x 1 x K x C taps); out0 [R][F]
// receives tap 0, outrest [K-1][R][F] taps 1 .. K-1 (K = max(Kin, Kst); taps beyond a filter's own count contribute zero).
extern "C" int gcrnn_taps_mfma_forward(int dtype, const void* zh, const void* zx, const void* wB, const void* wA, void* out0,
                                       void* outrest, int64_t R, int64_t F, int64_t Ch, int64_t Cx, int64_t Kst, int64_t Kin,
                                       void* stream) {
  if (!zh || !wB || !out0 || (Cx > 0 && (!zx || !wA))) return GCRNN_ERR_NULL_POINTER;
  if (R <= 0 || Kst <= 0 || (Cx > 0 && Kin <= 0) || !gcrnn_taps_mfma_supported(dtype, F, Ch, Cx)) return GCRNN_ERR_BAD_SHAPE;
  const int64_t K = (Cx > 0 && Kin > Kst) ? Kin : Kst;
  if (K > 1 && !outrest) return GCRNN_ERR_NULL_POINTER;
  if (K > 65535) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
  if (dtype == GCRNN_F32)
    return launch_t<float>(zh, zx, wB, wA, out0, outrest, R, (int)F, (int)Ch, (int)Cx, (int)Kst, (int)(Cx > 0 ? Kin : 0), st);
  return launch_t<double>(zh, zx, wB, wA, out0, outrest, R, (int)F, (int)Ch, (int)Cx, (int)Kst, (int)(Cx > 0 ? Kin : 0), st);
}
