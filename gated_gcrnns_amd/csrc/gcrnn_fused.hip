// Fused GCRNN time step for gfx950 (CDNA4): the flagship path (N <= 1024 nodes, bf16 storage, fp32 accumulate).
//
// One launch = one time step t for a whole batch:
//     h_t = tanh( gi * (A(S) x_t + b) + gf * (B(S) h_{t-1} + b) )                (reference graphML.py:2420-2423)
// One workgroup (512 threads = 8 waves, 2 per SIMD) owns one (sequence b, 16-feature output chunk c).
//
// Algebra.  With P = S^T acting on node-major rows and W_k = [B_k | A_k] (F x (F+G)) the step is
//     pre = sum_k P^k ([h|x] W_k^T)  + 2b          (taps and shifts commute: they act on different axes)
// evaluated in Horner form   acc = u_{K-1};  acc = P acc + u_{K-2}; ... ; acc = P acc + u_0,
// u_k = [h|x] W_k^T restricted to the chunk's 16 output features. x and h share one accumulator chain, so
// only (K-1) hops over 16 channels are needed per chunk (the reference does 2(K-1) hops over G+F channels).
//
// Phase 1 (MFMA): u_k^T tile = W_k(chunk) [16 x 128] * [h|x]^T [128 x 16 nodes] with v_mfma_f32_16x16x32_bf16.
//     A operand = weight fragments, pre-arranged per lane in LDS (one ds_read_b128 each);
//     B operand = 8 consecutive bf16 features of one node, a 16-byte global load from the node-major row;
//     D: lane holds 4 consecutive output features of node (lane & 15) -> exactly one 16-byte LDS slot.
//     u_{K-1} goes to LDS, u_0..u_{K-2} stay in registers (8 tiles x (K-1) x 4 fp32 per lane).
// Phase 2 (LDS gather): K-1 hops acc'[n] = sum_m P[n,m] acc[m] + u_k[n] on a fp32 [1024][16] image in LDS
//     (64-byte rows), double-buffered, one barrier per hop. The graph comes as degree-sorted sliced ELL
//     (16 nodes per slice, entries [e][16]), so the neighbour loop is wave-uniform and its (col,val)
//     loads are 128-byte coalesced; each gather is one ds_read_b128 + 4 FMAs per lane.
// Epilogue: + bias, tanh, bf16 store of the chunk into the node-major state h_t[b][n][c*16 .. +15].
//
// HBM traffic per (sequence, step): read x_t and h_{t-1} (each N*64*2 B; the 4 chunk workgroups of a sequence
// are placed on one XCD so that three of the four reads hit its L2), write h_t: the compulsory
// T*s*N*(G+2F) of SURVEY.md section 8d.
#include "gcrnn_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
constexpr int FC = 16;          // output features per workgroup
constexpr int WAVES = 8;
constexpr int TILES = 8;        // node tiles (16 nodes) per wave
constexpr int NP = WAVES * TILES * 16;   // 1024 padded nodes
}

__device__ __forceinline__ uint16_t f2bf(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN-safe
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// ------------------------------------------------------------------------------------------
// layout: user [B][T][C][N]  <->  sequence-major [T][B][NP][C], node positions renumbered by perm,
// rows N..NP-1 zero. E = element size carrier (uint16_t for bf16, uint32_t for f32).
// ------------------------------------------------------------------------------------------
template <typename E, bool PACK>
__global__ __launch_bounds__(256) void seq_layout_kernel(const E* __restrict__ src, E* __restrict__ dst, int B, int Tn,
                                                         int C, int N, int NPad, const int32_t* __restrict__ perm) {
  __shared__ E tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int bt = blockIdx.z, b = bt / Tn, t = bt - b * Tn;
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

template <bool PACK>
static int seq_layout_launch(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                             int64_t NPad, const int32_t* perm, void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || C <= 0 || N <= 0 || NPad < N || B * T > 65535 || cdiv(C, 32) > 65535) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  dim3 grid((unsigned)cdiv(PACK ? NPad : N, 32), (unsigned)cdiv(C, 32), (unsigned)(B * T));
  if (dtype == GCRNN_BF16)
    seq_layout_kernel<uint16_t, PACK><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)src, (uint16_t*)dst, (int)B,
                                                                             (int)T, (int)C, (int)N, (int)NPad, perm);
  else if (dtype == GCRNN_F32)
    seq_layout_kernel<uint32_t, PACK><<<grid, 256, 0, as_stream(stream)>>>((const uint32_t*)src, (uint32_t*)dst, (int)B,
                                                                             (int)T, (int)C, (int)N, (int)NPad, perm);
  else
    return GCRNN_ERR_BAD_DTYPE;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_pack_seq_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                                    int64_t NPad, const int32_t* perm, void* stream) {
  return seq_layout_launch<true>(dtype, src, dst, B, T, C, N, NPad, perm, stream);
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
  if (F <= 0 || G <= 0 || F % FC || (F + G) % 32 || F % 8 || Kin <= 0 || Kst <= 0) return GCRNN_ERR_BAD_SHAPE;
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

// ------------------------------------------------------------------------------------------
// the fused step
// ------------------------------------------------------------------------------------------
template <int K, int HS, int XS, bool GATED>
__global__ __launch_bounds__(512, 2) void fused_step_kernel(
    const uint16_t* __restrict__ xt,      // [B][NP][G]   bf16
    const uint16_t* __restrict__ hprev,   // [B][NP][F]   bf16
    uint16_t* __restrict__ hout,          // [B][NP][F]   bf16
    const uint4* __restrict__ wpack,      // [F/16][K][KS][64] x 16 B
    const float* __restrict__ bias,       // [F] or null
    const float* __restrict__ gi,         // [B] (GATED)
    const float* __restrict__ gf,         // [B] (GATED)
    const int32_t* __restrict__ tile_off, // [NP/16 + 1], in entries
    const int32_t* __restrict__ ell_col,  // [entries][16] neighbour position
    const float* __restrict__ ell_val,    // [entries][16]
    int B, int N) {
  constexpr int KS = HS + XS;
  constexpr int F = 32 * HS, G = 32 * XS;
  constexpr int NCH = F / FC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* buf0 = reinterpret_cast<float*>(smem);
  float* buf1 = buf0 + NP * FC;
  uint4* wl = reinterpret_cast<uint4*>(buf1 + NP * FC);

  // XCD-aware placement: the NCH chunk workgroups of one sequence get block ids that are equal mod 8,
  // i.e. one XCD under round-robin dispatch (speed only; nothing depends on it).
  const int L = blockIdx.x;
  const int grp = L / (8 * NCH), rem = L - grp * (8 * NCH);
  const int chunk = rem >> 3, b = grp * 8 + (rem & 7);
  if (b >= B) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  for (int i = tid; i < K * KS * 64; i += 512) wl[i] = wpack[(int64_t)chunk * K * KS * 64 + i];
  __syncthreads();

  const uint16_t* hb = hprev + (int64_t)b * NP * F;
  const uint16_t* xb = xt + (int64_t)b * NP * G;
  float gin = 1.f, gfo = 1.f;
  if (GATED) { gin = gi[b]; gfo = gf[b]; }

  f32x4 u[TILES][K > 1 ? K - 1 : 1];

  // ---- phase 1: taps on the matrix cores ------------------------------------------------------
#pragma unroll
  for (int i = 0; i < TILES; ++i) {
    const int node = (i * WAVES + wave) * 16 + r;
    bf16x8 bfrag[KS];
#pragma unroll
    for (int s = 0; s < HS; ++s)
      bfrag[s] = *reinterpret_cast<const bf16x8*>(hb + (int64_t)node * F + 32 * s + 8 * q);
#pragma unroll
    for (int s = 0; s < XS; ++s)
      bfrag[HS + s] = *reinterpret_cast<const bf16x8*>(xb + (int64_t)node * G + 32 * s + 8 * q);
#pragma unroll
    for (int tap = 0; tap < K; ++tap) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (GATED) {
        f32x4 accx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < HS; ++s) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, wl[(tap * KS + s) * 64 + lane]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfrag[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int s = HS; s < KS; ++s) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, wl[(tap * KS + s) * 64 + lane]);
          accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfrag[s], accx, 0, 0, 0);
        }
        acc = gfo * acc + gin * accx;
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, wl[(tap * KS + s) * 64 + lane]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfrag[s], acc, 0, 0, 0);
        }
      }
      if (tap == K - 1) *reinterpret_cast<f32x4*>(buf0 + node * FC + q * 4) = acc;
      else u[i][tap] = acc;
    }
  }
  __syncthreads();

  // ---- phase 2: Horner hops in LDS ---------------------------------------------------------------
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    const float bs = gin + gfo;     // the one bias is added by both filters (graphML.py:2420-2421)
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[c] = bs * bias[chunk * FC + q * 4 + c];
  }
  const char* cur = reinterpret_cast<const char*>(buf0);
  char* nxt = reinterpret_cast<char*>(buf1);
  const int qoff = q * 16;
#pragma unroll
  for (int j = 1; j < K; ++j) {
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      const int tile = i * WAVES + wave;
      const int node = tile * 16 + r;
      const int beg = __builtin_amdgcn_readfirstlane(tile_off[tile]);
      const int end = __builtin_amdgcn_readfirstlane(tile_off[tile + 1]);
      f32x4 acc = u[i][K - 1 - j];
      for (int e = beg; e < end; e += 4) {      // entry counts are padded to multiples of 4
        int cc[4]; float vv[4]; f32x4 xv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { cc[p] = ell_col[(e + p) * 16 + r]; vv[p] = ell_val[(e + p) * 16 + r]; }
#pragma unroll
        for (int p = 0; p < 4; ++p) xv[p] = *reinterpret_cast<const f32x4*>(cur + cc[p] * (FC * 4) + qoff);
#pragma unroll
        for (int p = 0; p < 4; ++p) acc += vv[p] * xv[p];
      }
      if (j == K - 1) {
        uint2 pk;
        if (node < N) {
          const float o0 = tanhf(acc[0] + bsum[0]), o1 = tanhf(acc[1] + bsum[1]);
          const float o2 = tanhf(acc[2] + bsum[2]), o3 = tanhf(acc[3] + bsum[3]);
          pk.x = (uint32_t)f2bf(o0) | ((uint32_t)f2bf(o1) << 16);
          pk.y = (uint32_t)f2bf(o2) | ((uint32_t)f2bf(o3) << 16);
        } else {
          pk.x = 0u; pk.y = 0u;          // padded rows stay zero
        }
        *reinterpret_cast<uint2*>(hout + ((int64_t)b * NP + node) * F + chunk * FC + q * 4) = pk;
      } else {
        *reinterpret_cast<f32x4*>(nxt + node * (FC * 4) + qoff) = acc;
      }
    }
    if (j < K - 1) {
      __syncthreads();
      const char* t = cur; cur = nxt; nxt = const_cast<char*>(t);
    }
  }
}

template <int K, int HS, int XS>
static int fused_forward_t(const void* xs, const void* h0, void* hs, const void* wpack, const float* bias,
                           const float* gi, const float* gf, const int32_t* tile_off, const int32_t* ell_col,
                           const float* ell_val, int64_t B, int64_t T, int64_t N, hipStream_t st) {
  constexpr int F = 32 * HS, G = 32 * XS, KS = HS + XS;
  const size_t lds = (size_t)2 * NP * FC * 4 + (size_t)K * KS * 64 * 16;
  const bool gated = gi != nullptr;
  auto kern = gated ? fused_step_kernel<K, HS, XS, true> : fused_step_kernel<K, HS, XS, false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  const int NCH = F / FC;
  const unsigned grid = (unsigned)(cdiv(B, 8) * 8 * NCH);
  const uint16_t* x = (const uint16_t*)xs;
  uint16_t* h = (uint16_t*)hs;
  const int64_t xstep = B * NP * G, hstep = B * NP * F;
  GCRNN_PRE_LAUNCH();
  for (int64_t t = 0; t < T; ++t) {
    const uint16_t* hp = (t == 0) ? (const uint16_t*)h0 : h + (t - 1) * hstep;
    kern<<<grid, 512, lds, st>>>(x + t * xstep, hp, h + t * hstep, (const uint4*)wpack, bias, gated ? gi + t * B : nullptr,
                                 gated ? gf + t * B : nullptr, tile_off, ell_col, ell_val, (int)B, (int)N);
  }
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_fused_forward_bf16(const void* xs, const void* h0, void* hs, const void* wpack, const float* bias,
                                        const float* gi, const float* gf, const int32_t* tile_off,
                                        const int32_t* ell_col, const float* ell_val, int64_t B, int64_t T, int64_t N,
                                        int64_t F, int64_t G, int64_t K, void* stream) {
  if (!xs || !h0 || !hs || !wpack || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B > (1 << 24)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_FUSED_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) \
    return fused_forward_t<KK, HH, XX>(xs, h0, hs, wpack, bias, gi, gf, tile_off, ell_col, ell_val, B, T, N, st);
  GCRNN_FUSED_CASE(5, 2, 2)
  GCRNN_FUSED_CASE(4, 2, 2)
  GCRNN_FUSED_CASE(3, 2, 2)
  GCRNN_FUSED_CASE(2, 2, 2)
  GCRNN_FUSED_CASE(5, 1, 1)
  GCRNN_FUSED_CASE(3, 1, 1)
  GCRNN_FUSED_CASE(2, 1, 1)
#undef GCRNN_FUSED_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

extern "C" int gcrnn_fused_supported(int64_t N, int64_t F, int64_t G, int64_t K) {
  if (N <= 0 || N > NP) return 0;
  const bool big = (F == 64 && G == 64 && K >= 2 && K <= 5);
  const bool small = (F == 32 && G == 32 && (K == 2 || K == 3 || K == 5));
  return (big || small) ? 1 : 0;
}

extern "C" int64_t gcrnn_fused_padded_nodes(void) { return NP; }
