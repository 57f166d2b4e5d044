// Fused GCRNN path for gfx950: layout kernels, weight packing, the C entry points of the step kernel (its template lives in
// gcrnn_fused_step.h, its instantiations in gcrnn_fused_step_k*.hip), the gate read-out backward and the BPTT seed.
#include "gcrnn_fused_step.h"

GCRNN_STEP_FOR_K5(GCRNN_STEP_DECLARE)
GCRNN_STEP_FOR_K4(GCRNN_STEP_DECLARE)
GCRNN_STEP_FOR_K3(GCRNN_STEP_DECLARE)
GCRNN_STEP_FOR_K2(GCRNN_STEP_DECLARE)

// ------------------------------------------------------------------------------------------
// layout: user [B][T][C][N]  <->  sequence-major [T][B][NP][C], node positions renumbered by perm,
// rows N..NP-1 zero. E = element size carrier (uint16_t for bf16, uint32_t for f32).
// ------------------------------------------------------------------------------------------
template <typename E, bool PACK>
__global__ __launch_bounds__(256) void seq_layout_kernel(const E* __restrict__ src, E* __restrict__ dst, int B, int Tn,
                                                         int C, int N, int NPad, const int32_t* __restrict__ perm, int z0 = 0) {
  __shared__ E tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int bt = (int)blockIdx.z + z0, b = bt / Tn, t = bt - b * Tn;      // (z0: launches of at most 65535 items each, the grid's z limit)
  const int64_t ubase = ((int64_t)(b * Tn + t) * C) * N;          // user [C][N] block
  const int64_t sbase = ((int64_t)(t * B + b) * NPad) * C;        // seq-major [NPad][C] block
  if (PACK) {
    const int n = n0 + tx;
    const int nsrc = (n < N) ? (perm ? perm[n] : n) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i;
      tile[ty + 8 * i][tx] = (c < C && n < N) ? src[ubase + (int64_t)c * N + nsrc] : E(0);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nn = n0 + ty + 8 * i, c = c0 + tx;
      if (nn < NPad && c < C) dst[sbase + (int64_t)nn * C + c] = tile[tx][ty + 8 * i];   // rows >= N get the zeros
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nn = n0 + ty + 8 * i, c = c0 + tx;
      tile[ty + 8 * i][tx] = (nn < N && c < C) ? src[sbase + (int64_t)nn * C + c] : E(0);
    }
    __syncthreads();
    const int n = n0 + tx;
    const int ndst = (n < N) ? (perm ? perm[n] : n) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i;
      if (c < C && n < N) dst[ubase + (int64_t)c * N + ndst] = tile[tx][ty + 8 * i];
    }
  }
}

// Fast path for 16-bit elements, no permutation, even N and C: 64 (c) x 64 (n) tiles, every thread moves one
// 4-byte pair per row, so both the user side (rows of n) and the sequence-major side (rows of c) are touched
// in 128-byte segments.
template <bool PACK>
__global__ __launch_bounds__(256) void seq_layout16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                           int B, int Tn, int C, int N, int NPad, int Cs = 0, int z0 = 0) {
  // PACK with Cs < C (Cs = channels of the USER tensor, C = channels of the sequence-major one): channels >= Cs are written as
  // zeros -- the reference drivers' G = 1 input (kStepPredGRNNs.py:220) reaches the kernels' 32-channel operand without a padded
  // copy of X in the user layout. Cs = 0: the same channel count on both sides.
  __shared__ uint16_t tile[64][66];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int bt = (int)blockIdx.z + z0, b = bt / Tn, t = bt - b * Tn;
  const int Cu = (PACK && Cs > 0) ? Cs : C;
  const int64_t ubase = ((int64_t)(b * Tn + t) * Cu) * N;
  const int64_t sbase = ((int64_t)(t * B + b) * NPad) * C;
  if (PACK) {
    const int n = n0 + 2 * tx;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + ty + 8 * i;
      uint32_t v = 0;
      if (c < Cu && n < N) v = *reinterpret_cast<const uint32_t*>(src + ubase + (int64_t)c * N + n);
      *reinterpret_cast<uint32_t*>(&tile[ty + 8 * i][2 * tx]) = v;
    }
    __syncthreads();
    const int c = c0 + 2 * tx;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int nl = ty + 8 * i, nn = n0 + nl;
      if (nn < NPad && c < C) {
        const uint32_t v = (uint32_t)tile[2 * tx][nl] | ((uint32_t)tile[2 * tx + 1][nl] << 16);
        *reinterpret_cast<uint32_t*>(dst + sbase + (int64_t)nn * C + c) = v;     // rows >= N receive the zeros
      }
    }
  } else {
    const int c = c0 + 2 * tx;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int nl = ty + 8 * i, nn = n0 + nl;
      uint32_t v = 0;
      if (nn < N && c < C) v = *reinterpret_cast<const uint32_t*>(src + sbase + (int64_t)nn * C + c);
      tile[2 * tx][nl] = (uint16_t)(v & 0xffffu);
      tile[2 * tx + 1][nl] = (uint16_t)(v >> 16);
    }
    __syncthreads();
    const int n = n0 + 2 * tx;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int cc = c0 + ty + 8 * i;
      if (cc < C && n < N)
        *reinterpret_cast<uint32_t*>(dst + ubase + (int64_t)cc * N + n) = *reinterpret_cast<const uint32_t*>(&tile[ty + 8 * i][2 * tx]);
    }
  }
}

// Pack of a range of time steps with a small LDS footprint (4.2 KiB): meant to run on a SECOND stream beside the step
// kernels, whose one workgroup per CU leaves only a few KiB of LDS (and ~40 VGPRs per lane) free -- the packs of steps
// t+1.. then hide behind the recurrence of steps ..t instead of preceding it. 32 (c) x 64 (n) tiles: 128-byte reads along n,
// 64-byte writes along c (two workgroups complete each 128-byte row). bf16, no permutation, even N and C.
__global__ __launch_bounds__(256) void seq_pack16_small_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                               int B, int Tn, int C, int N, int NPad, int t0, int nt, int ntx,
                                                               int nty, int ntiles) {
  __shared__ uint16_t tile[32][66];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int cp = threadIdx.x & 15, nr = threadIdx.x >> 4;        // write side: 16 column pairs x 16 rows per pass
  // a bounded grid walks the tiles: the kernel is meant to trickle along beside the step kernels, not to race them for HBM
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int bx = tl % ntx, rest = tl / ntx, by = rest % nty, bt = rest / nty;
    const int n0 = bx * 64, c0 = by * 32;
    const int b = bt / nt, t = t0 + (bt - b * nt);
    const int64_t ubase = ((int64_t)(b * Tn + t) * C) * N;
    const int64_t sbase = ((int64_t)(t * B + b) * NPad) * C;
    const int n = n0 + 2 * tx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i;
      uint32_t v = 0;
      if (c < C && n < N) v = *reinterpret_cast<const uint32_t*>(src + ubase + (int64_t)c * N + n);
      *reinterpret_cast<uint32_t*>(&tile[ty + 8 * i][2 * tx]) = v;
    }
    __syncthreads();
    const int c = c0 + 2 * cp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = nr + 16 * i, nn = n0 + nl;
      if (nn < NPad && c < C) {
        const uint32_t v = (uint32_t)tile[2 * cp][nl] | ((uint32_t)tile[2 * cp + 1][nl] << 16);
        *reinterpret_cast<uint32_t*>(dst + sbase + (int64_t)nn * C + c) = v;     // rows >= N receive the zeros
      }
    }
    __syncthreads();
  }
}

// steps [t0, t1) of the bf16 pack only, on the small-footprint kernel above; max_blocks > 0 bounds the grid
extern "C" int gcrnn_pack_seq_major_steps(const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N, int64_t NPad,
                                          int64_t t0, int64_t t1, int64_t max_blocks, void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || C <= 0 || N <= 0 || NPad < N || t0 < 0 || t1 > T || t0 >= t1) return GCRNN_ERR_BAD_SHAPE;
  if ((N % 2) || (C % 2) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3)) return GCRNN_ERR_UNSUPPORTED;
  const int64_t ntx = cdiv(NPad, 64), nty = cdiv(C, 32), ntiles = ntx * nty * B * (t1 - t0);
  if (ntiles > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  int64_t grid = ntiles;
  if (max_blocks > 0 && grid > max_blocks) grid = max_blocks;
  GCRNN_PRE_LAUNCH();
  seq_pack16_small_kernel<<<(unsigned)grid, 256, 0, as_stream(stream)>>>((const uint16_t*)src, (uint16_t*)dst, (int)B, (int)T, (int)C,
                                                                          (int)N, (int)NPad, (int)t0, (int)(t1 - t0), (int)ntx,
                                                                          (int)nty, (int)ntiles);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// flag[0] = 1 when every bf16 element of src is +-0, else 0 (flag pre-set to 1 by the host call; a workgroup that meets a non-zero clears it)
__global__ __launch_bounds__(256) void all_zero_flag_kernel(const uint4* __restrict__ src, int64_t n16, int32_t* __restrict__ flag) {
  uint32_t any = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    const uint4 v = src[i];
    any |= (v.x | v.y | v.z | v.w) & 0x7fff7fffu;
  }
  if (__builtin_amdgcn_ballot_w64(any != 0) != 0 && (threadIdx.x & 63) == 0) *flag = 0;      // (plain store of the same value by every finder: no atomics needed)
}

// The time-gated cell's "h0 is all zeros" flag (every training loop of the reference starts from zeros, Modules/train_rnn.py:256; the gate
// kernels then skip the state half of their operand -- exact), decided on the device without a host round trip: flag int32[1] = 1 when all
// `elements` bf16 values at src are +-0. elements % 8 == 0, 16-byte aligned src.
extern "C" int gcrnn_all_zero_flag_bf16(const void* src, int64_t elements, int32_t* flag, void* stream) {
  if (!src || !flag) return GCRNN_ERR_NULL_POINTER;
  if (elements <= 0 || elements % 8 || (reinterpret_cast<uintptr_t>(src) & 15)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
  GCRNN_PRE_LAUNCH();
  const int64_t n16 = elements / 8;
  int64_t grid = cdiv(n16, 256 * 8);
  if (grid > 2048) grid = 2048;
  if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flag), 1, 1, st) != hipSuccess) return GCRNN_ERR_LAUNCH;
  all_zero_flag_kernel<<<(unsigned)grid, 256, 0, st>>>((const uint4*)src, n16, flag);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

template <bool PACK>
static int seq_layout_launch(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                             int64_t NPad, const int32_t* perm, void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || C <= 0 || N <= 0 || NPad < N || B * T > 2147483647LL || cdiv(C, 32) > 65535) return GCRNN_ERR_BAD_SHAPE;
  if (dtype != GCRNN_BF16 && dtype != GCRNN_F32) return GCRNN_ERR_BAD_DTYPE;
  GCRNN_PRE_LAUNCH();
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0;
  const bool fast = dtype == GCRNN_BF16 && !perm && (N % 2 == 0) && (C % 2 == 0) && aligned;
  for (int64_t z0 = 0; z0 < B * T; z0 += 65535) {      // the grid's z extent holds 65535 items: more (B = 2048 at T = 32) run as several launches
    const unsigned nz = (unsigned)(B * T - z0 < 65535 ? B * T - z0 : 65535);
    if (fast) {
      dim3 grid((unsigned)cdiv(PACK ? NPad : N, 64), (unsigned)cdiv(C, 64), nz);
      seq_layout16_kernel<PACK><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)src, (uint16_t*)dst, (int)B, (int)T,
                                                                      (int)C, (int)N, (int)NPad, 0, (int)z0);
      continue;
    }
    dim3 grid((unsigned)cdiv(PACK ? NPad : N, 32), (unsigned)cdiv(C, 32), nz);
    if (dtype == GCRNN_BF16)
      seq_layout_kernel<uint16_t, PACK><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)src, (uint16_t*)dst, (int)B,
                                                                               (int)T, (int)C, (int)N, (int)NPad, perm, (int)z0);
    else
      seq_layout_kernel<uint32_t, PACK><<<grid, 256, 0, as_stream(stream)>>>((const uint32_t*)src, (uint32_t*)dst, (int)B,
                                                                               (int)T, (int)C, (int)N, (int)NPad, perm, (int)z0);
  }
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_pack_seq_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                                    int64_t NPad, const int32_t* perm, void* stream) {
  return seq_layout_launch<true>(dtype, src, dst, B, T, C, N, NPad, perm, stream);
}
// bf16 pack with channel padding: user [B][T][Cs][N] -> sequence-major [T][B][NPad][C], C >= Cs even, channels >= Cs zero.
// Replaces "zero-pad X to the kernels' input width in the user layout, then pack" (ops.fused_pad_operands materialised a 32x larger
// copy of the drivers' one-channel X, kStepPredGRNNs.py:220). N even, 4-byte aligned arrays.
extern "C" int gcrnn_pack_seq_major_padded(const void* src, void* dst, int64_t B, int64_t T, int64_t Cs, int64_t C, int64_t N,
                                           int64_t NPad, void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || Cs <= 0 || C < Cs || N <= 0 || NPad < N || B * T > 2147483647LL || cdiv(C, 64) > 65535) return GCRNN_ERR_BAD_SHAPE;
  if ((N % 2) || (C % 2) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3)) return GCRNN_ERR_UNSUPPORTED;
  GCRNN_PRE_LAUNCH();
  for (int64_t z0 = 0; z0 < B * T; z0 += 65535) {
    dim3 grid((unsigned)cdiv(NPad, 64), (unsigned)cdiv(C, 64), (unsigned)(B * T - z0 < 65535 ? B * T - z0 : 65535));
    seq_layout16_kernel<true><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)src, (uint16_t*)dst, (int)B, (int)T, (int)C, (int)N,
                                                                   (int)NPad, (int)Cs, (int)z0);
  }
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_unpack_seq_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C,
                                      int64_t N, int64_t NPad, const int32_t* perm, void* stream) {
  return seq_layout_launch<false>(dtype, src, dst, B, T, C, N, NPad, perm, stream);
}

// ------------------------------------------------------------------------------------------
// weights -> per-lane MFMA A fragments (bf16):  wpack[chunk][tap][kstep][lane][8]
//   row  f' = chunk*16 + (lane & 15);  k = 32*kstep + 8*(lane >> 4) + j  over the concatenated [h | x] features
// ------------------------------------------------------------------------------------------
template <typename W>
__global__ void pack_weights_kernel(const W* __restrict__ wA, const W* __restrict__ wB, uint16_t* __restrict__ out,
                                    int F, int G, int Kin, int Kst, int K) {
  const int KS = (F + G) / 32;
  const int64_t total = (int64_t)(F / FC) * K * KS * 64 * 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 7, lane = (idx >> 3) & 63;
  int64_t rest = idx >> 9;
  const int s = rest % KS; rest /= KS;
  const int tap = rest % K;
  const int chunk = rest / K;
  const int f = chunk * FC + (lane & 15);
  const int feat = 32 * s + 8 * (lane >> 4) + j;
  float v = 0.f;
  if (feat < F) { if (tap < Kst) v = (float)wB[((int64_t)f * Kst + tap) * F + feat]; }
  else          { if (tap < Kin) v = (float)wA[((int64_t)f * Kin + tap) * G + (feat - F)]; }
  out[idx] = f2bf(v);
}

extern "C" int gcrnn_fused_pack_weights(int wdtype, const void* wA, const void* wB, void* wpack, int64_t F, int64_t G,
                                        int64_t Kin, int64_t Kst, void* stream) {
  if (!wA || !wB || !wpack) return GCRNN_ERR_NULL_POINTER;
  if (F <= 0 || G < 0 || F % FC || (F + G) % 32 || F % 8 || Kin <= 0 || Kst <= 0) return GCRNN_ERR_BAD_SHAPE;
  const int K = (int)(Kin > Kst ? Kin : Kst);
  const int64_t total = (F / FC) * K * ((F + G) / 32) * 64 * 8;
  GCRNN_PRE_LAUNCH();
  if (wdtype == GCRNN_F32)
    pack_weights_kernel<float><<<(unsigned)cdiv(total, 256), 256, 0, as_stream(stream)>>>(
        (const float*)wA, (const float*)wB, (uint16_t*)wpack, (int)F, (int)G, (int)Kin, (int)Kst, K);
  else if (wdtype == GCRNN_BF16)
    pack_weights_kernel<__hip_bfloat16><<<(unsigned)cdiv(total, 256), 256, 0, as_stream(stream)>>>(
        (const __hip_bfloat16*)wA, (const __hip_bfloat16*)wB, (uint16_t*)wpack, (int)F, (int)G, (int)Kin, (int)Kst, K);
  else
    return GCRNN_ERR_BAD_DTYPE;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}


static int fused_dispatch(int mode, const void* xs, const void* h0, void* hs, const void* wpack, const float* bias,
                          const float* gi, const float* gf, const float* gate_w, float* gate_out, const FusedGraphArgs& ga,
                          int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, hipStream_t st,
                          const void* bw_dHs = nullptr, const void* bw_hs = nullptr, void* bw_dh0 = nullptr,
                          void* huser = nullptr, const void* bw_h0 = nullptr, void* const* step_events = nullptr,
                          int huser_last_only = 0, const int32_t* hzero_flag = nullptr) {
#define GCRNN_FUSED_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) \
    return fused_launch_t<KK, HH, XX>(mode, xs, h0, hs, wpack, bias, gi, gf, gate_w, gate_out, ga, B, T, N, st, bw_dHs, \
                                      bw_hs, bw_h0, bw_dh0, huser, step_events, huser_last_only, hzero_flag);
  GCRNN_FUSED_CASE(5, 2, 2)
  GCRNN_FUSED_CASE(4, 2, 2)
  GCRNN_FUSED_CASE(3, 2, 2)
  GCRNN_FUSED_CASE(2, 2, 2)
  GCRNN_FUSED_CASE(5, 1, 1)
  GCRNN_FUSED_CASE(4, 1, 1)
  GCRNN_FUSED_CASE(3, 1, 1)
  GCRNN_FUSED_CASE(2, 1, 1)
  GCRNN_FUSED_CASE(5, 2, 1)      // F = 64 with up to 32 input features (the drivers' G = 1, zero-padded to 32 by the caller)
  GCRNN_FUSED_CASE(4, 2, 1)
  GCRNN_FUSED_CASE(3, 2, 1)
  GCRNN_FUSED_CASE(2, 2, 1)
  GCRNN_FUSED_CASE(5, 2, 0)      // BPTT data-gradient steps and gate-gradient passes: ONE operand array (dpre, x or h alone)
  GCRNN_FUSED_CASE(4, 2, 0)
  GCRNN_FUSED_CASE(3, 2, 0)
  GCRNN_FUSED_CASE(2, 2, 0)
  GCRNN_FUSED_CASE(5, 1, 0)
  GCRNN_FUSED_CASE(4, 1, 0)
  GCRNN_FUSED_CASE(3, 1, 0)
  GCRNN_FUSED_CASE(2, 1, 0)
#undef GCRNN_FUSED_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

extern "C" int gcrnn_fused_forward_bf16(const void* xs, const void* h0, void* hs, const void* wpack, const float* bias,
                                        const float* gi, const float* gf, const int32_t* tile_nodes,
                                        const int32_t* tile_off, const int32_t* ell_col, const float* ell_val,
                                        const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B,
                                        int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, void* Huser,
                                        int huser_last_only, void* const* step_events, double uniform_w, const void* Xuser_inline,
                                        const float* head_w, float* head_part, void* stream) {
  if (!xs || !h0 || !hs || !wpack || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (Xuser_inline && (gi || step_events || (reinterpret_cast<uintptr_t>(Xuser_inline) & 15))) return GCRNN_ERR_BAD_SHAPE;
  if ((head_w == nullptr) != (head_part == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B > (1 << 24) || entries < 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  if (B * (NP * (F > G ? F : G) * 2) > 2147483647LL || T * F * N > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;   // 32-bit buffer offsets
  if (Huser && (N % 8 != 0 || (reinterpret_cast<uintptr_t>(Huser) & 15))) return GCRNN_ERR_BAD_SHAPE;
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, (huser_last_only >> 1) & 1};
  return fused_dispatch(gi ? 1 : 0, xs, h0, hs, wpack, bias, gi, gf, head_w, head_part, ga, B, T, N, F, G, K, as_stream(stream),
                        Xuser_inline, nullptr, nullptr, Huser, nullptr, step_events, huser_last_only & 1);
}

// Can gcrnn_fused_forward_bf16 take Xuser_inline for this problem (un-gated cell, uniform-weight graph image that leaves LDS
// room for the [G][NPad / (F/16)] input tile, N % 8 == 0)?
extern "C" int gcrnn_fused_inline_pack_supported(int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries, double uniform_w) {
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
  if (uniform_w == 0.0 || N % 8 || N > NP || G <= 0 || G % 32 || (F != 32 && F != 64) || entries % 4) return 0;
  const size_t base = (size_t)NP * FC * 4 + (size_t)K * ((F + G) / 32) * 1024;
  const size_t need = base + (size_t)entries * 32 + (size_t)G * (NP / (F / FC)) * 2;
  return need <= 160 * 1024;
#else
  return 0;
#endif
}

// Will gcrnn_fused_forward_bf16 (backward = 0) / gcrnn_fused_backward_data_bf16 (backward != 0) run this problem on the sequence-
// resident persistent kernel (gcrnn_fused_seq.h: ONE launch for all T steps, one workgroup per sequence) rather than on T launches
// of the chunk-parallel step kernel? Un-gated cell without a fused head, uniform-weight bf16-image plan (img16), a batch that fills
// whole rounds of the chip (cost model in gcrnn_fused_seq.h; GCRNN_SEQ_KERNEL=0 / GCRNN_SEQ_MIN_B override it). Returns the number
// of time steps one launch covers: T (persistent), 1 (the same kernel launched per step: GCRNN_SEQ_PERSIST=0) or 0 (chunk-parallel).
extern "C" int64_t gcrnn_fused_seq_steps_per_launch(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                                    double uniform_w, int img16, int inline_pack, int backward) {
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
  if (uniform_w == 0.0 || !img16 || N <= 0 || N > NP || B <= 0 || T <= 0 || entries % 4 || !gcrnn_fused_supported(N, F, G > 0 ? (G > 32 ? 64 : 32) : F, K)) return 0;
  if (F != 32 && F != 64) return 0;
  const int nch = (int)(F / FC);
  if (!fused_seq_wanted(B, nch)) return 0;
  const int64_t ks = backward ? F / 32 : (F + G) / 32, pkrows = backward ? F : G;
  (void)pkrows;      // (the inline-pack tile is the second hop image)
  if (!fused_seq_lds_bytes(K, ks, entries)) return 0;
  return fused_seq_persistent() ? (backward ? (T > 1 ? T - 1 : 1) : T) : 1;
#else
  return 0;
#endif
}

extern "C" int gcrnn_fused_gate_prepass_bf16(const void* xs, const void* h0, const void* wpack, const float* bias,
                                             const float* gate_w, float* gate_out, void* cs, const int32_t* tile_nodes,
                                             const int32_t* tile_off, const int32_t* ell_col, const float* ell_val,
                                             const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B,
                                             int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, const int32_t* h0_zero_flag,
                                             double uniform_w, int img16, void* stream) {
  if (!xs || !h0 || !wpack || !gate_w || !gate_out || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B * T > (1 << 24) || entries < 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, img16 ? 1 : 0};
  return fused_dispatch(2, xs, h0, cs, wpack, bias, nullptr, nullptr, gate_w, gate_out, ga, B, T, N, F, G, K, as_stream(stream),
                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, h0_zero_flag);
}

// The gate pre-pass that also lays out X: x_user = the user-layout input [B][T][G][N] bf16 (channels already padded to the kernels'
// 32 / 64), xs = the sequence-major array [T][B][NP][G] with its first gcrnn_fused_gate_prepass_lays_out(...) time steps laid out by
// the caller; on return every step is. Each item of the sequence-resident kernel lays out the operand of its workgroup's next item
// (as the forward steps lay out x_{t+1}), so the separate pass over X (0.5 ms at B = 256, T = 32) goes away for gated cells.
extern "C" int gcrnn_fused_gate_prepass_pack_bf16(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias,
                                                  const float* gate_w, float* gate_out, void* cs, const int32_t* tile_nodes,
                                                  const int32_t* tile_off, const int32_t* ell_col, const float* ell_val,
                                                  const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B,
                                                  int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, const int32_t* h0_zero_flag,
                                                  double uniform_w, int img16, void* stream) {
  if (!x_user || !xs || !h0 || !wpack || !gate_w || !gate_out || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B * T > (1 << 24) || entries < 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, img16 ? 1 : 0};
  return fused_dispatch(2, xs, h0, cs, wpack, bias, nullptr, nullptr, gate_w, gate_out, ga, B, T, N, F, G, K, as_stream(stream),
                        x_user, nullptr, nullptr, nullptr, nullptr, nullptr, 0, h0_zero_flag);
}

// The gate-cell pre-pass of a NODE gate at inference (graphML.py:2379-2393): tanh(A_g(S) x_t + B_g(S) h0 + 2 b_g) for every (t, b), and -- fused
// into its epilogue -- the per-tap dot products s_k[n] = <c[n, :], w_k> that start the gate's F -> 1 GraphFilter (taps first, :2387).
// tap_frags: the taps as MFMA A fragments, [F/16][3][64] x 8 B (three bf16 planes of the fp32 taps: lane l = 16 kg + tap holds
// w_plane[tap][16 chunk + 4 kg + e], e = 0..3; taps >= ntaps zero); taps_out [T*B][ntaps][N] fp32. cs (or NULL) as in the plain
// pre-pass; x_user (or NULL) as in gcrnn_fused_gate_prepass_pack_bf16. Sequence-resident kernel only: GCRNN_ERR_UNSUPPORTED where
// gcrnn_fused_gate_prepass_taps_supported returns 0.
extern "C" int gcrnn_fused_gate_prepass_taps_bf16(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias,
                                                  const void* tap_frags, float* taps_out, int64_t ntaps, void* cs,
                                                  const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col,
                                                  const float* ell_val, const void* ell_val4, const void* ell_col4, int64_t entries,
                                                  int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K,
                                                  const int32_t* h0_zero_flag, double uniform_w, int img16, void* stream) {
  if (!xs || !h0 || !wpack || !tap_frags || !taps_out || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B * T > (1 << 24) || entries < 0 || entries % 4 || ntaps < 1 || ntaps > 8) return GCRNN_ERR_BAD_SHAPE;
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, img16 ? 1 : 0};
  return fused_dispatch(2, xs, h0, cs, wpack, bias, nullptr, nullptr, nullptr, nullptr, ga, B, T, N, F, G, K, as_stream(stream),
                        x_user, tap_frags, taps_out, nullptr, nullptr, nullptr, (int)ntaps, h0_zero_flag);
}

// 1 when gcrnn_fused_gate_prepass_taps_bf16 takes this shape (with_pack: together with the layout of X)
extern "C" int gcrnn_fused_gate_prepass_taps_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                                       double uniform_w, int img16, int with_pack, int64_t ntaps) {
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
  if (uniform_w == 0.0 || !img16 || N <= 0 || N > NP || B <= 0 || T <= 0 || entries % 4 || (G != 32 && G != 64) || (F != 32 && F != 64)) return 0;
  if (ntaps < 1 || ntaps > 8 || !gcrnn_fused_supported(N, F, G, K) || (with_pack && N % 8)) return 0;
  const int nch = (int)(F / FC);
  if (!fused_seq_wanted(B * T, nch)) return 0;
  if (!fused_seq_lds_bytes(K, (F + G) / 32, entries, (size_t)ntaps * NP * 4 + (size_t)nch * 3 * 512)) return 0;
  const int64_t row_bytes = (int64_t)NP * (F > G ? F : G) * 2;
  if ((2147483647LL / row_bytes) / B < T) return 0;
  return 1;
#else
  return 0;
#endif
}

// 0: gcrnn_fused_gate_prepass_pack_bf16 is not available for this shape (chunk-parallel kernel, weighted graph, N % 8, LDS); else the
// number of leading time steps of xs the caller must lay out itself (the items of the first round of workgroups).
extern "C" int64_t gcrnn_fused_gate_prepass_lays_out(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                                     double uniform_w, int img16) {
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
  if (uniform_w == 0.0 || !img16 || N <= 0 || N > NP || N % 8 || B <= 0 || T <= 0 || entries % 4 || (G != 32 && G != 64) || (F != 32 && F != 64)) return 0;
  if (!gcrnn_fused_supported(N, F, G, K)) return 0;
  const int nch = (int)(F / FC);
  if (!fused_seq_wanted(B * T, nch)) return 0;
  if (!fused_seq_lds_bytes(K, (F + G) / 32, entries) || T * G * N > 2147483647LL) return 0;
  const int64_t row_bytes = (int64_t)NP * (F > G ? F : G) * 2;
  if ((2147483647LL / row_bytes) / B < T) return 0;                  // (the launch would be split over time: not with the pack)
  const int64_t first = B * T < GCRNN_SEQ_MAX_GRID ? B * T : GCRNN_SEQ_MAX_GRID;      // (the items of the first round of workgroups: the same constant as the launch)
  return (first + B - 1) / B;
#else
  return 0;
#endif
}

// d loss / d (scalar time gate) of one filter of the gated cell:  out[t*B+b][partials] summed = sum_{f,n} (W(S) z + b) . dpre
// xs == null (G = 0): z = zs [T][B][NP][F] bf16 sequence-major is that filter's operand (h_{t-1} for the state filter, or x_t
// for an input filter with G == F), wpack its taps packed as a state-only operand (gcrnn_fused_pack_weights with G = 0).
// xs != null: input filter with G != F input features: operand [0 | x_t], xs [T][B][NP][G], zs = ONE all-zero block [NP][F],
// wpack = gcrnn_fused_pack_weights(A, zero state taps). bias [F] or null (added once), dpre: [T][B][NP][F] bf16.
// Reference: the gates multiply the two filter outputs, graphML.py:2420-2421.
extern "C" int gcrnn_fused_gate_grad_bf16(const void* zs, const void* xs, const void* dpre, const void* wpack, const float* bias,
                                          float* out, const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col,
                                          const float* ell_val, const void* ell_val4, const void* ell_col4, int64_t entries,
                                          int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, double uniform_w, int img16, void* stream) {
  if (!zs || !dpre || !wpack || !out || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if ((xs == nullptr) != (G == 0)) return GCRNN_ERR_BAD_SHAPE;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B * T > (1 << 24) || entries < 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, img16 ? 1 : 0};
  return fused_dispatch(4, xs, zs, nullptr, wpack, bias, nullptr, nullptr, nullptr, out, ga, B, T, N, F, G, K, as_stream(stream), dpre);
}


// Filter output of every (t, b) item in one launch: out[t][b][n][:] = (W(S) z)[n][:] + bias, bf16 sequence-major -- the input
// filter A(S)x_t + b of the node-gated cell for all steps at once (it does not depend on the recurrence; graphML.py:2402-2403).
// Operand conventions of gcrnn_fused_gate_grad_bf16: xs == null (G = 0): zs [T][B][NP][F] is the operand (an input with G == F
// packed like a state), wpack its taps as a state-only operand; xs != null: operand [0 | x_t], zs = ONE all-zero block [NP][F].
extern "C" int gcrnn_fused_filter_output_bf16(const void* zs, const void* xs, const void* wpack, const float* bias, void* out,
                                              const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col,
                                              const float* ell_val, const void* ell_val4, const void* ell_col4, int64_t entries,
                                              int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, double uniform_w, int img16, void* stream) {
  if (!zs || !wpack || !out || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if ((xs == nullptr) != (G == 0)) return GCRNN_ERR_BAD_SHAPE;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B * T > (1 << 24) || entries < 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, img16 ? 1 : 0};
  return fused_dispatch(5, xs, zs, out, wpack, bias, nullptr, nullptr, nullptr, nullptr, ga, B, T, N, F, G, K, as_stream(stream));
}

// Node-gated cell (graphML.py:2379-2407), T launches:  h_t = tanh(gi ni_t . Yx_t + gf nf_t . (B(S)h_{t-1} + b)).
// h0s [B][NP][F], hs [T][B][NP][F] (out), yx [T][B][NP][F] = A(S)x_t + b from gcrnn_fused_filter_output_bf16 (all bf16 sequence-major);
// ngates fp32 [T][2][B][N]: per-node input gates then forget gates of every step; gi / gf fp32 [T][B] scalar time gates or both NULL;
// wpackB = the state taps packed as a state-only operand (gcrnn_fused_pack_weights with G = 0); yh_out (or NULL) [T][B][NP][F]
// receives B(S)h_{t-1} + b (kept for the BPTT); Huser as in gcrnn_fused_forward_bf16.
extern "C" int gcrnn_fused_node_forward_bf16(const void* h0s, void* hs, const void* yx, const float* ngates, const float* gi,
                                             const float* gf, const void* wpackB, const float* bias, void* yh_out,
                                             const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col,
                                             const float* ell_val, const void* ell_val4, const void* ell_col4, int64_t entries,
                                             int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, void* Huser, int huser_last_only,
                                             double uniform_w, void* stream) {
  if (!h0s || !hs || !yx || !ngates || !wpackB || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B > (1 << 24) || entries < 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  if (B * (NP * F * 2) > 2147483647LL || T * F * N > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (Huser && (N % 8 != 0 || (reinterpret_cast<uintptr_t>(Huser) & 15))) return GCRNN_ERR_BAD_SHAPE;
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, (huser_last_only >> 1) & 1};
  return fused_dispatch(6, nullptr, h0s, hs, wpackB, bias, gi, gf, ngates, nullptr, ga, B, T, N, F, 0, K, as_stream(stream), yx, nullptr,
                        yh_out, Huser, nullptr, nullptr, huser_last_only & 1);
}

// dpre[i] = dH[i] * (1 - h[i]^2) on bf16 arrays (the seed of the BPTT chain, t = T-1)
__global__ void bwd_seed_kernel(const uint16_t* __restrict__ dH, const uint16_t* __restrict__ h, uint16_t* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= n) return;
  const uint32_t g = *reinterpret_cast<const uint32_t*>(dH + i), hv = *reinterpret_cast<const uint32_t*>(h + i);
  const float g0 = bf2f((uint16_t)(g & 0xffffu)), g1 = bf2f((uint16_t)(g >> 16));
  const float h0 = bf2f((uint16_t)(hv & 0xffffu)), h1 = bf2f((uint16_t)(hv >> 16));
  *reinterpret_cast<uint32_t*>(out + i) = pack2bf(g0 * (1.f - h0 * h0), g1 * (1.f - h1 * h1));
}

// out[i][n][:] = in[i][n][:] * g[i][n]  (bf16 rows of F, fp32 per-node factors; rows >= N stay zero)
__global__ void scale_rows_kernel(const uint16_t* __restrict__ in, const float* __restrict__ g, uint16_t* __restrict__ out, int64_t items,
                                  int N, int NPad, int F) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread = one feature pair of one row
  const int fp = F / 2;
  if (idx >= items * NPad * fp) return;
  const int64_t row = idx / fp;
  const int n = (int)(row % NPad);
  const int64_t item = row / NPad;
  uint32_t o = 0;
  if (n < N) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(in + idx * 2);
    const float gn = g[item * N + n];
    o = pack2bf(bf2f((uint16_t)(v & 0xffffu)) * gn, bf2f((uint16_t)(v >> 16)) * gn);
  }
  *reinterpret_cast<uint32_t*>(out + idx * 2) = o;
}

// BPTT data-gradient chain of the NODE-gated cell (graphML.py:2402-2407 under autograd): dpre_t = (dH_t + rec_t)(1 - h_t^2),
// rec_{t-1} = sum_k S^k ((gf nf)_t . dpre_t B_k). dHs, hs, dpre (out), dyh (out: (gf nf) . dpre, the chain's operands):
// [T][B][NPad][F] bf16 sequence-major; ngf fp32 [T][B][N] = gf_t[b] nf_t[b][n]; wpackT / graph arrays as in
// gcrnn_fused_backward_data_bf16. T launches of the same step kernel.
extern "C" int gcrnn_fused_node_backward_data_bf16(const void* dHs, const void* hs, void* dpre, void* dyh, const float* ngf,
                                                   const void* wpackT, const int32_t* tile_nodes, const int32_t* tile_off,
                                                   const int32_t* ell_col, const float* ell_val, const void* ell_val4,
                                                   const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F,
                                                   int64_t K, double uniform_w, const void* dHuser_inline, int img16, void* stream) {
  if (!dHs || !hs || !dpre || !dyh || !ngf || !wpackT || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if (dHuser_inline && (reinterpret_cast<uintptr_t>(dHuser_inline) & 15)) return GCRNN_ERR_BAD_SHAPE;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B > (1 << 24) || entries < 0 || entries % 4 || F % 2) return GCRNN_ERR_BAD_SHAPE;
  const int64_t step = B * NP * F;
  GCRNN_PRE_LAUNCH();
  bwd_seed_kernel<<<(unsigned)cdiv(step / 2, 256), 256, 0, as_stream(stream)>>>(
      (const uint16_t*)dHs + (T - 1) * step, (const uint16_t*)hs + (T - 1) * step, (uint16_t*)dpre + (T - 1) * step, step);
  scale_rows_kernel<<<(unsigned)cdiv(step / 2, 256), 256, 0, as_stream(stream)>>>(
      (const uint16_t*)dpre + (T - 1) * step, ngf + (T - 1) * B * N, (uint16_t*)dyh + (T - 1) * step, B, (int)N, NP, (int)F);
  GCRNN_CHECK_LAUNCH();
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, img16 ? 1 : 0};
  return fused_dispatch(7, dyh, nullptr, dpre, wpackT, nullptr, nullptr, nullptr, ngf, nullptr, ga, B, T, N, F, 0, K, as_stream(stream), dHs, hs,
                        nullptr, nullptr, dHuser_inline);
}

// ONE step of the BPTT data chain with explicit arrays (the edge-gated cell interleaves it with the attention backward):
//   dpre_prev = (sum_k S^k (operand W_k) + dH_prev) (1 - h_prev^2);  operand, dH_prev, h_prev, dpre_prev: [B][NPad][F] bf16
// sequence-major, wpackT = the transposed taps packed as a state-only operand (as in gcrnn_fused_backward_data_bf16).
// dHuser_next / dHs_next (or both NULL): inline pack -- the user-layout block dH[0][t-2] (T = sequence length: its item stride is
// T F N) is laid out into dHs_next = dHs[t-2] by this launch (uniform-weight graphs, gcrnn_fused_inline_pack_supported).
// bwd_seed_kernel's formula dpre = dH (1 - h^2) for the last step is gcrnn_fused_backward_seed_bf16.
extern "C" int gcrnn_fused_backward_step_bf16(const void* operand, const void* dH_prev, const void* h_prev, void* dpre_prev,
                                              const void* wpackT, const int32_t* tile_nodes, const int32_t* tile_off,
                                              const int32_t* ell_col, const float* ell_val, const void* ell_val4, const void* ell_col4,
                                              int64_t entries, int64_t B, int64_t N, int64_t F, int64_t K, double uniform_w,
                                              const void* dHuser_next, void* dHs_next, int64_t T, int img16, void* stream) {
  if (!operand || !dH_prev || !h_prev || !dpre_prev || !wpackT || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || N <= 0 || N > NP || B > (1 << 24) || entries < 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  if ((dHuser_next == nullptr) != (dHs_next == nullptr) || (dHuser_next && (T <= 0 || (reinterpret_cast<uintptr_t>(dHuser_next) & 15)))) return GCRNN_ERR_BAD_SHAPE;
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, img16 ? 1 : 0};
  return fused_dispatch(8, dHuser_next, operand, dpre_prev, wpackT, nullptr, nullptr, nullptr, nullptr, nullptr, ga, B, dHuser_next ? T : 1, N, F, 0, K,
                        as_stream(stream), dH_prev, h_prev, dHs_next);
}

extern "C" int gcrnn_fused_backward_seed_bf16(const void* dH, const void* h, void* dpre, int64_t elements, void* stream) {
  if (!dH || !h || !dpre) return GCRNN_ERR_NULL_POINTER;
  if (elements <= 0 || elements % 2) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  bwd_seed_kernel<<<(unsigned)cdiv(elements / 2, 256), 256, 0, as_stream(stream)>>>((const uint16_t*)dH, (const uint16_t*)h, (uint16_t*)dpre, elements);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_fused_backward_data_bf16(const void* dHs, const void* hs, void* dpre, void* dh0, const void* wpackT,
                                              const int32_t* tile_nodes, const int32_t* tile_off, const int32_t* ell_col,
                                              const float* ell_val, const void* ell_val4, const void* ell_col4,
                                              int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t K,
                                              const float* gf, const void* h0s, float* dgf_parts, double uniform_w,
                                              const void* dHuser_inline, int img16, void* stream) {
  if (dgf_parts && !h0s) return GCRNN_ERR_NULL_POINTER;
  if (dHuser_inline && (reinterpret_cast<uintptr_t>(dHuser_inline) & 15)) return GCRNN_ERR_BAD_SHAPE;
  if (!dHs || !hs || !dpre || !wpackT || !tile_nodes || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B > (1 << 24) || entries < 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  const int64_t step = B * NP * F;
  GCRNN_PRE_LAUNCH();
  bwd_seed_kernel<<<(unsigned)cdiv(step / 2, 256), 256, 0, as_stream(stream)>>>(
      (const uint16_t*)dHs + (T - 1) * step, (const uint16_t*)hs + (T - 1) * step, (uint16_t*)dpre + (T - 1) * step, step);
  GCRNN_CHECK_LAUNCH();
  const FusedGraphArgs ga{tile_nodes, tile_off, ell_col, ell_val, ell_val4, ell_col4, entries, (float)uniform_w, img16 ? 1 : 0};
  return fused_dispatch(3, dHuser_inline, nullptr, dpre, wpackT, nullptr, nullptr, gf, nullptr, dgf_parts, ga, B, T, N, F, 0, K,
                        as_stream(stream), dHs, hs, dh0, nullptr, h0s);
}

// ------------------------------------------------------------------------------------------
// BPTT through a time gate's read-out (graphML.py:2364-2366): gate = sigmoid(w . vec(c) + c0), c = tanh(pre_g) stored by the
// gate pre-pass. One pass over c [items][NPad][F] (bf16, in place):
//     dpre_g[item][n][f] = dlogit[item] * w[n][f] * (1 - c^2)          (overwrites c; feeds the weight-gradient kernel)
//     dw[n][f]          += dlogit[item] * c[item][n][f]                (per item-slab partial sums, plain stores)
// A thread owns one pair of adjacent (n, f) columns and walks its slab's items; consecutive threads touch consecutive 4-byte
// words, so every wave reads and writes whole 256-byte lines.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_readout_bwd_kernel(uint16_t* __restrict__ cs, const float* __restrict__ dlogit,
                                                               const float* __restrict__ gate_w, float* __restrict__ dw_part,
                                                               int items, int cols /* NPad*F */, int valid /* N*F */, int per_slab) {
  const int col = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (col >= cols) return;
  const int it0 = blockIdx.y * per_slab, it1 = min(items, it0 + per_slab);
  float w0 = 0.f, w1 = 0.f;
  if (col < valid) { w0 = gate_w[col]; w1 = gate_w[col + 1]; }      // rows >= N: c is zero there and stays zero
  float a0 = 0.f, a1 = 0.f;
  uint32_t* p = reinterpret_cast<uint32_t*>(cs + (int64_t)it0 * cols + col);
  const int64_t stride = cols / 2;
#pragma unroll 4
  for (int it = it0; it < it1; ++it, p += stride) {
    const uint32_t v = *p;
    const float c0 = bf2f((uint16_t)(v & 0xffffu)), c1 = bf2f((uint16_t)(v >> 16));
    const float dl = dlogit[it];
    a0 += dl * c0;
    a1 += dl * c1;
    *p = pack2bf(dl * w0 * (1.f - c0 * c0), dl * w1 * (1.f - c1 * c1));
  }
  *reinterpret_cast<float2*>(dw_part + (int64_t)blockIdx.y * cols + col) = float2{a0, a1};
}

extern "C" int64_t gcrnn_fused_gate_readout_slabs(int64_t items) { return items < 64 ? 1 : (items < 1024 ? 8 : 32); }

extern "C" int gcrnn_fused_gate_readout_backward_bf16(void* cs, const float* dlogit, const float* gate_w, float* dw_part,
                                                      int64_t items, int64_t N, int64_t F, void* stream) {
  if (!cs || !dlogit || !gate_w || !dw_part) return GCRNN_ERR_NULL_POINTER;
  if (items <= 0 || items > (1 << 24) || N <= 0 || N > NP || F <= 0 || F % 2) return GCRNN_ERR_BAD_SHAPE;
  const int64_t slabs = gcrnn_fused_gate_readout_slabs(items);
  const int cols = (int)(NP * F);
  GCRNN_PRE_LAUNCH();
  dim3 grid((unsigned)cdiv(cols / 2, 256), (unsigned)slabs);
  gate_readout_bwd_kernel<<<grid, 256, 0, as_stream(stream)>>>((uint16_t*)cs, dlogit, gate_w, dw_part, (int)items, cols,
                                                              (int)(N * F), (int)cdiv(items, slabs));
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_fused_supported(int64_t N, int64_t F, int64_t G, int64_t K) {
  if (N <= 0 || N > NP) return 0;
  const bool big = (F == 64 && (G == 64 || G == 32) && K >= 2 && K <= 5);
  const bool small = (F == 32 && G == 32 && K >= 2 && K <= 5);
  return (big || small) ? 1 : 0;
}

extern "C" int64_t gcrnn_fused_padded_nodes(void) { return NP; }
// waves per workgroup of the step kernels / of the weight-gradient kernel: the ELL tiles of a plan are dealt out over that
// many waves (storage tile w * (NP/16/waves) + i belongs to wave w)
extern "C" int64_t gcrnn_fused_step_waves(void) { return SWAVES; }
extern "C" int64_t gcrnn_fused_wgrad_waves(void) { return WAVES; }
