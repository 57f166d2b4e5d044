// Filter taps of a Horner-form GCRNN step on the matrix cores, bf16 rows / fp32 accumulation, any number of rows:
//     u_k[r][:] = z_h[r][:] B_k^T + z_x[r][:] A_k^T         k = 0 .. K-1,   r = (node, sequence) row of the node-major layout
// All K taps of a step in ONE pass over [z_h | z_x] (the operand rows are read once, K outputs are written): the streaming
// path for graphs beyond LDS (BASELINE configs[4]) runs  acc = u_{K-1};  acc = P acc + u_k  on these (graphML.py:118-135 in
// Horner form). Replaces the per-tap library GEMMs of round 1.
//   D^T tile = W_k(chunk)[16 x 32 s] * z^T[32 s x 16 rows]  with v_mfma_f32_16x16x32_bf16: A = weight fragments from LDS (the
//   layout gcrnn_fused_pack_weights writes), B = 16-byte row loads (8 consecutive features of one row), D leaves each lane with
//   4 consecutive output features of one row = one 8-byte bf16 store.
#include "gcrnn_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

__device__ __forceinline__ uint32_t pk2(float a, float b) {
  return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)b) << 16);
}

template <int HS, int XS>
__global__ __launch_bounds__(256) void taps_bf16_kernel(const uint16_t* __restrict__ zh, const uint16_t* __restrict__ zx,
                                                        const uint4* __restrict__ wpack, uint16_t* __restrict__ out0,
                                                        uint16_t* __restrict__ outrest, int64_t R, int K) {
  constexpr int KS = HS + XS, F = 32 * HS, G = 32 * XS, NCH = F / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* wl = reinterpret_cast<uint4*>(smem);                  // [NCH][K][KS][64] x 16 B
  const int nfrag = NCH * K * KS * 64;
  for (int i = threadIdx.x; i < nfrag; i += 256) wl[i] = wpack[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t tiles = (R + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row = tile * 16 + r;
    const bool ok = row < R;
    bf16x8 bfr[KS];
#pragma unroll
    for (int s = 0; s < HS; ++s)
      bfr[s] = ok ? __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(zh + row * F + 32 * s + 8 * q))
                  : __builtin_bit_cast(bf16x8, uint4{0u, 0u, 0u, 0u});
#pragma unroll
    for (int s = 0; s < XS; ++s)
      bfr[HS + s] = ok ? __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(zx + row * G + 32 * s + 8 * q))
                       : __builtin_bit_cast(bf16x8, uint4{0u, 0u, 0u, 0u});
    for (int tap = 0; tap < K; ++tap) {
      uint16_t* dst = (tap == 0) ? out0 : outrest + (int64_t)(tap - 1) * R * F;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, wl[((c * K + tap) * KS + s) * 64 + lane]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[s], acc, 0, 0, 0);
        }
        if (ok) *reinterpret_cast<uint2*>(dst + row * F + c * 16 + q * 4) = uint2{pk2(acc[0], acc[1]), pk2(acc[2], acc[3])};
      }
    }
  }
}

template <int HS, int XS>
int taps_launch(const void* zh, const void* zx, const void* wpack, void* out0, void* outrest, int64_t R, int K, hipStream_t st) {
  constexpr int KS = HS + XS, NCH = 2 * HS;
  const size_t lds = (size_t)NCH * K * KS * 1024;
  if (lds > 160 * 1024) return GCRNN_ERR_UNSUPPORTED;
  auto kern = taps_bf16_kernel<HS, XS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  const int64_t tiles = (R + 15) / 16;
  int64_t grid = (tiles + 3) / 4;
  const int64_t cap = 256 * (lds > 40 * 1024 ? 2 : 8);       // grid-stride: the weight image is staged once per workgroup
  if (grid > cap) grid = cap;
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)grid, 256, lds, st>>>((const uint16_t*)zh, (const uint16_t*)zx, (const uint4*)wpack, (uint16_t*)out0, (uint16_t*)outrest, R, K);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

}  // namespace

extern "C" int gcrnn_taps_bf16_supported(int64_t F, int64_t G, int64_t K) {
  const bool shape = (F == 32 || F == 64) && (G == 0 || G == 32 || G == 64) && !(F == 32 && G == 64);
  return shape && K >= 1 && (F / 16) * K * ((F + G) / 32) * 1024 <= 160 * 1024;
}

// zh [R][F], zx [R][G] (null when G == 0), wpack = gcrnn_fused_pack_weights(A, B) ([F/16][K][(F+G)/32][64] x 16 B, bf16),
// out0 [R][F] receives tap 0, outrest [K-1][R][F] taps 1 .. K-1 (may be null when K == 1); all bf16, 16-byte aligned.
extern "C" int gcrnn_taps_bf16_forward(const void* zh, const void* zx, const void* wpack, void* out0, void* outrest, int64_t R,
                                       int64_t F, int64_t G, int64_t K, void* stream) {
  if (!zh || !wpack || !out0 || (G > 0 && !zx) || (K > 1 && !outrest)) return GCRNN_ERR_NULL_POINTER;
  if (R <= 0 || !gcrnn_taps_bf16_supported(F, G, K)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
  if (F == 64 && G == 64) return taps_launch<2, 2>(zh, zx, wpack, out0, outrest, R, (int)K, st);
  if (F == 64 && G == 32) return taps_launch<2, 1>(zh, zx, wpack, out0, outrest, R, (int)K, st);
  if (F == 64 && G == 0) return taps_launch<2, 0>(zh, zx, wpack, out0, outrest, R, (int)K, st);
  if (F == 32 && G == 32) return taps_launch<1, 1>(zh, zx, wpack, out0, outrest, R, (int)K, st);
  if (F == 32 && G == 0) return taps_launch<1, 0>(zh, zx, wpack, out0, outrest, R, (int)K, st);
  return GCRNN_ERR_UNSUPPORTED;
}
