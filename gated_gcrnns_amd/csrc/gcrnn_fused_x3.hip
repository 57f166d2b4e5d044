// fp32-accurate fused GCRNN step on the bf16 matrix cores ("x3"): the 1e-5 parity mode of the north_star at fused-kernel speed.
//
// Same decomposition as fused_step_kernel (gcrnn_fused_step.h): one workgroup = (sequence, 16-feature chunk), taps on the matrix
// cores, K-1 Horner hops on an fp32 LDS state image, tanh epilogue -- the reference's h_t = tanh(A(S)x_t + b + B(S)h_{t-1} + b)
// (Utils/graphML.py:2420-2423, un-gated cell). What changes is the operand format: every fp32 operand v is carried as THREE bf16
// planes  v = v1 + v2 + v3,  v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2)  (3 x 8 significant bits = fp32's 24), for the
// state, the input AND the taps. A product z.w is evaluated as the six partial products whose weight is at least 2^-16 of the
// leading one -- z1w1, z1w2, z1w3, z2w1, z2w2, z3w1 -- on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: bf16 x bf16 products
// are exact in fp32, so the result is an fp32 dot product to within a few ulp, at 6/16 of the cost of the fp32 MFMA
// (v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate). Hops, bias, tanh are fp32 as before; the state goes back to HBM as
// three planes (6 bytes per element) and, for the caller, as fp32 in the user layout H[b][t][f][:].
// Uniform-weight graphs only (the weight planes take 3 x the LDS of the bf16 kernel: 64 KiB state + 60 KiB taps + 32 B x entries).
#include "gcrnn_fused_step.h"

namespace {

__device__ __forceinline__ void split3(float v, uint16_t& a, uint16_t& b, uint16_t& c) {
  a = f2bf(v);
  const float r1 = v - bf2f(a);           // exact
  b = f2bf(r1);
  const float r2 = r1 - bf2f(b);          // exact
  c = f2bf(r2);
}

// user fp32 [B][T][C][N]  ->  three bf16 planes, sequence-major [T][3][B][NPad][C], rows N..NPad-1 zero
__global__ __launch_bounds__(256) void seq_pack_x3_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int B, int Tn, int C,
                                                          int N, int NPad) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int bt = blockIdx.z, b = bt / Tn, t = bt - b * Tn;
  const int64_t ubase = ((int64_t)(b * Tn + t) * C) * N;
  const int n = n0 + tx;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    tile[ty + 8 * i][tx] = (c < C && n < N) ? src[ubase + (int64_t)c * N + n] : 0.f;
  }
  __syncthreads();
  const int cp = threadIdx.x & 15, nr = threadIdx.x >> 4;             // 16 column pairs x 16 rows per pass
  const int c = c0 + 2 * cp;
  const int64_t plane = (int64_t)B * NPad * C;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int nl = nr + 16 * i, nn = n0 + nl;
    if (nn < NPad && c < C) {                                          // C is even (host check): whole pairs
      uint16_t a0, a1, a2, b0, b1, b2;
      split3(tile[2 * cp][nl], a0, a1, a2);
      split3(tile[2 * cp + 1][nl], b0, b1, b2);
      uint16_t* d = dst + ((int64_t)t * 3 * B + b) * NPad * C + (int64_t)nn * C + c;
      *reinterpret_cast<uint32_t*>(d) = (uint32_t)a0 | ((uint32_t)b0 << 16);
      *reinterpret_cast<uint32_t*>(d + plane) = (uint32_t)a1 | ((uint32_t)b1 << 16);
      *reinterpret_cast<uint32_t*>(d + 2 * plane) = (uint32_t)a2 | ((uint32_t)b2 << 16);
    }
  }
}

// taps -> three planes of per-lane MFMA A fragments: out[p][chunk][tap][kstep][lane][8] (layout of pack_weights_kernel per plane)
__global__ void pack_weights_x3_kernel(const float* __restrict__ wA, const float* __restrict__ wB, uint16_t* __restrict__ out, int F, int G,
                                       int Kin, int Kst, int K) {
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
  if (feat < F) { if (tap < Kst) v = wB[((int64_t)f * Kst + tap) * F + feat]; }
  else          { if (tap < Kin) v = wA[((int64_t)f * Kin + tap) * G + (feat - F)]; }
  uint16_t a, b, c;
  split3(v, a, b, c);
  out[idx] = a; out[idx + total] = b; out[idx + 2 * total] = c;
}

template <int K, int HS, int XS>
__global__ __launch_bounds__(STHREADS) void fused_step_x3_kernel(
    const uint16_t* __restrict__ xt3,        // [3][B][NP][G]  planes of x_t
    const uint16_t* __restrict__ hp3,        // [3][B][NP][F]  planes of h_{t-1}
    uint16_t* __restrict__ ho3,              // [3][B][NP][F]  planes of h_t
    const uint4* __restrict__ wpack3,        // [3][F/16][K][KS][64] x 16 B
    const float* __restrict__ bias,          // [F] or null
    const int32_t* __restrict__ tile_nodes, const int32_t* __restrict__ tile_off, const uint2* __restrict__ ell_col4,
    float* __restrict__ Huser,               // user-layout fp32 output block of this step, H[.][t][F][N] (or null)
    int64_t ubstride,                        // elements between consecutive sequences of Huser
    int entries, int B, int N, float uni_w) {
  static_assert(GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8, "x3 runs on the one-block asm hop stream");
  constexpr int KS = HS + XS, F = 32 * HS, G = 32 * XS, NCH = F / FC, HT = STILES;
  constexpr int WPL = K * KS * 64;            // uint4 fragments per weight plane and chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* state = reinterpret_cast<float*>(smem);
  uint4* wl = reinterpret_cast<uint4*>(smem + NP * FC * 4);                       // [3][K][KS][64]
  uint2* lcol4 = reinterpret_cast<uint2*>(smem + NP * FC * 4 + 3 * WPL * 16);

  const int L = blockIdx.x;
  const int grp = L / (8 * NCH), rem = L - grp * (8 * NCH);
  const int chunk = rem >> 3, b0 = grp * 8 + (rem & 7);
  const int seq_slots = (gridDim.x / (8 * NCH)) * 8;
  if (b0 >= B) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

#pragma unroll
  for (int p = 0; p < 3; ++p)
    for (int i = tid; i < WPL; i += STHREADS) wl[p * WPL + i] = wpack3[((int64_t)p * NCH + chunk) * WPL + i];
  {
    const int n = (entries >> 2) * 16;
    for (int i = tid; i < n; i += STHREADS) lcol4[i] = ell_col4[i];
  }
  int tbeg[STILES], tend[STILES], woff[STILES];
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
    tbeg[i] = tile_off[wave * STILES + i];
    tend[i] = tile_off[wave * STILES + i + 1];
    woff[i] = tile_nodes[(wave * STILES + i) * 16 + r] ^ (q << 4);            // node << 16 | row << 6 | swz << 4, this lane's quad
  }
  float bvec[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) bvec[c] = 2.f * bias[chunk * FC + q * 4 + c];   // the one bias enters through both filters (graphML.py:2420-2421)
  }
  __syncthreads();
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  if (lds0 != 0) __builtin_trap();
  const uint32_t qx = (uint32_t)(q * 16);
  const uint32_t lds_col = lds0 + NP * FC * 4 + 3 * WPL * 16;
  const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(hp3), 0, 3 * B * (NP * F * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xt3), 0, 3 * B * (NP * G * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(ho3, 0, 3 * B * (NP * F * 2), 0x00020000);

  for (int b = b0; b < B; b += seq_slots) {
    // ---- phase 1: taps, six partial products per operand pair, two tiles per weight fragment ------------------------------
    // 12 (tile pair, plane) stages; the fragments of stage g+1 are requested before the MFMAs of stage g (two register sets),
    // so one L2 latency is exposed per sequence instead of one per stage.
    f32x4 u[STILES][K - 1];
    bf16x8 fr[2][2][KS];                     // [set][tile of the pair][k-step]
    auto load_stage = [&](int g, int set) {
      const int i = 2 * (g / 3), p = g % 3;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int w = woff[i + j];
        asm volatile("" : "+v"(w));
        const int roh = (w >> 16) * (F * 2) + 16 * q, rox = (w >> 16) * (G * 2) + 16 * q;
#pragma unroll
        for (int s = 0; s < HS; ++s)
          fr[set][j][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, roh + 64 * s, (p * B + b) * (NP * F * 2), 0));
#pragma unroll
        for (int s = 0; s < XS; ++s)
          fr[set][j][HS + s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, rox + 64 * s, (p * B + b) * (NP * G * 2), 0));
      }
    };
    load_stage(0, 0);
    f32x4 acc[K][2];
#pragma unroll
    for (int g = 0; g < 3 * (STILES / 2); ++g) {
      const int i = 2 * (g / 3), p = g % 3, set = g & 1;
      if (p == 0) {
#pragma unroll
        for (int tap = 0; tap < K; ++tap) { acc[tap][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[tap][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
      if (g + 1 < 3 * (STILES / 2)) load_stage(g + 1, set ^ 1);
      __builtin_amdgcn_sched_barrier(0);       // keep the next stage's loads ahead of this stage's MFMAs
#pragma unroll
      for (int tap = 0; tap < K; ++tap)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int wp = 0; wp < 3 - p; ++wp) {             // operand plane p meets weight planes 0 .. 2-p
            const bf16x8 a = __builtin_bit_cast(bf16x8, wl[wp * WPL + (tap * KS + s) * 64 + lane]);
            acc[tap][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fr[set][0][s], acc[tap][0], 0, 0, 0);
            acc[tap][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fr[set][1][s], acc[tap][1], 0, 0, 0);
          }
      if (p == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int tap = 0; tap < K - 1; ++tap) u[i + j][tap] = acc[tap][j];
          int wv = woff[i + j];
          asm volatile("" : "+v"(wv));
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(state) + (wv & 0xffff)) = acc[K - 1][j];
        }
      }
    }
    __syncthreads();
    // ---- phase 2: Horner hops on the fp32 state image (uniform-weight asm stream) -------------------------------------------
#pragma unroll
    for (int j = 1; j < K; ++j) {
#define GCRNN_X3_INIT(i) u[i][K - 1 - j]
#define GCRNN_X3_STORE(i, a) u[i][K - 1 - j] = a
      GCRNN_HOP_ASM_UNI_STREAM(GCRNN_X3_INIT, GCRNN_X3_STORE);
#undef GCRNN_X3_INIT
#undef GCRNN_X3_STORE
      if (j < K - 1) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(state) + (wv & 0xffff)) = u[i][K - 1 - j];
        }
        __syncthreads();
      }
    }
    // ---- epilogue: bias, tanh, three planes of h_t; fp32 user-layout copy through an LDS transpose in two node halves ------
    constexpr int LASTU = (K > 1) ? 0 : 0;
#pragma unroll
    for (int i = 0; i < STILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      const int node = wv >> 16;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (node < N) {
        const f32x4 a = u[i][LASTU];
        o = f32x4{fast_tanh(a[0] + bvec[0]), fast_tanh(a[1] + bvec[1]), fast_tanh(a[2] + bvec[2]), fast_tanh(a[3] + bvec[3])};
      }
      uint16_t p0[4], p1[4], p2[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) split3(o[c], p0[c], p1[c], p2[c]);
      const int eoff = node * (F * 2) + (chunk * FC + q * 4) * 2;
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)p0[0] | ((uint32_t)p0[1] << 16), (uint32_t)p0[2] | ((uint32_t)p0[3] << 16)}, rsrc_o, eoff, (0 * B + b) * (NP * F * 2), 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)p1[0] | ((uint32_t)p1[1] << 16), (uint32_t)p1[2] | ((uint32_t)p1[3] << 16)}, rsrc_o, eoff, (1 * B + b) * (NP * F * 2), 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)p2[0] | ((uint32_t)p2[1] << 16), (uint32_t)p2[2] | ((uint32_t)p2[3] << 16)}, rsrc_o, eoff, (2 * B + b) * (NP * F * 2), 0);
      u[i][LASTU] = o;
    }
    if (Huser) {
      constexpr int RS = 512 * 4 + 16;                    // row stride of the transposed half tile [16 f][512 nodes] fp32
      char* tst = reinterpret_cast<char*>(state);
      float* ub = Huser + (int64_t)b * ubstride + (int64_t)(chunk * FC) * N;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        __syncthreads();                                  // the last hop's reads (hf = 0) / the previous half's row reads are done
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int node = wv >> 16;
          if ((node >> 9) == hf) {
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<float*>(tst + (q * 4 + c) * RS + (node & 511) * 4) = u[i][LASTU][c];
          }
        }
        __syncthreads();
        const int nhalf = (N - 512 * hf) < 512 ? (N - 512 * hf) : 512;      // valid nodes of this half (N % 4 == 0: whole 16-byte segments)
        const int segs = nhalf > 0 ? (nhalf >> 2) : 0;
        for (int idx = tid; idx < FC * segs; idx += STHREADS) {
          const int f = idx / segs, sg = idx - f * segs;
          *reinterpret_cast<float4*>(ub + (int64_t)f * N + 512 * hf + sg * 4) = *reinterpret_cast<const float4*>(tst + f * RS + sg * 16);
        }
      }
    }
    __syncthreads();      // `state` is free again before the next sequence's taps land in it
  }
}

template <int K, int HS, int XS>
int x3_launch(const void* xs3, const void* h03, void* hs3, const void* wpack3, const float* bias, const int32_t* tile_nodes,
              const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, float uni_w, float* Huser,
              int last_only, hipStream_t st) {
  constexpr int F = 32 * HS, G = 32 * XS, KS = HS + XS, NCH = F / FC;
  const size_t lds = (size_t)NP * FC * 4 + (size_t)3 * K * KS * 1024 + (size_t)entries * 32;
  if (lds > 160 * 1024) return GCRNN_ERR_UNSUPPORTED;
  auto kern = fused_step_x3_kernel<K, HS, XS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  int64_t slots = cdiv(B, 8) * 8;
  const int64_t max_slots = (256 / NCH) / 8 * 8 > 0 ? (256 / NCH) / 8 * 8 : 8;
  if (slots > max_slots) slots = max_slots;
  const int64_t xstep = 3 * B * NP * G, hstep = 3 * B * NP * F;
  const uint16_t* x = (const uint16_t*)xs3;
  uint16_t* h = (uint16_t*)hs3;
  GCRNN_PRE_LAUNCH();
  for (int64_t t = 0; t < T; ++t) {
    const uint16_t* hp = (t == 0) ? (const uint16_t*)h03 : h + (t - 1) * hstep;
    float* hu = !Huser ? nullptr : (!last_only ? Huser + t * F * N : (t == T - 1 ? Huser : nullptr));
    kern<<<(unsigned)(slots * NCH), STHREADS, lds, st>>>(x + t * xstep, hp, h + t * hstep, (const uint4*)wpack3, bias, tile_nodes, tile_off,
                                                        (const uint2*)ell_col4, hu, (int64_t)(last_only ? 1 : T) * F * N, (int)entries, (int)B,
                                                        (int)N, uni_w);
  }
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

}  // namespace

// Shapes the fp32-accurate fused path is built for: F = G = 64 / 32 and F = 64 with G <= 32 (padded to 32), K in 2..5,
// N <= 1024 with N % 4 == 0, a uniform-weight graph image of `entries` ELL entries that fits next to the state and three tap planes.
extern "C" int gcrnn_fused_x3_supported(int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries) {
  if (N <= 0 || N > NP || N % 4 || entries < 0 || entries % 4 || K < 2 || K > 5) return 0;
  if (!((F == 64 && (G == 64 || G == 32)) || (F == 32 && G == 32))) return 0;
  return (int64_t)NP * FC * 4 + 3 * K * ((F + G) / 32) * 1024 + entries * 32 <= 160 * 1024;
}

// user fp32 [B][T][C][N] -> bf16 planes [T][3][B][NPad][C] (C even)
extern "C" int gcrnn_pack_seq_major_x3(const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N, int64_t NPad, void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || C <= 0 || C % 2 || N <= 0 || NPad < N || B * T > 65535 || cdiv(C, 32) > 65535) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  seq_pack_x3_kernel<<<dim3((unsigned)cdiv(NPad, 32), (unsigned)cdiv(C, 32), (unsigned)(B * T)), 256, 0, as_stream(stream)>>>(
      (const float*)src, (uint16_t*)dst, (int)B, (int)T, (int)C, (int)N, (int)NPad);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// fp32 taps wA [F][Kin][G], wB [F][Kst][F] -> wpack3 [3][F/16][K][(F+G)/32][64][8] bf16 planes (K = max(Kin, Kst))
extern "C" int gcrnn_fused_pack_weights_x3(const void* wA, const void* wB, void* wpack3, int64_t F, int64_t G, int64_t Kin, int64_t Kst,
                                           void* stream) {
  if (!wA || !wB || !wpack3) return GCRNN_ERR_NULL_POINTER;
  if (F <= 0 || G <= 0 || F % FC || (F + G) % 32 || Kin <= 0 || Kst <= 0) return GCRNN_ERR_BAD_SHAPE;
  const int K = (int)(Kin > Kst ? Kin : Kst);
  const int64_t total = (F / FC) * K * ((F + G) / 32) * 64 * 8;
  GCRNN_PRE_LAUNCH();
  pack_weights_x3_kernel<<<(unsigned)cdiv(total, 256), 256, 0, as_stream(stream)>>>((const float*)wA, (const float*)wB, (uint16_t*)wpack3,
                                                                                     (int)F, (int)G, (int)Kin, (int)Kst, K);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// T launches of the x3 step kernel: xs3 [T][3][B][NPad][G], h03 [3][B][NPad][F], hs3 [T][3][B][NPad][F] (all bf16 planes),
// wpack3 from gcrnn_fused_pack_weights_x3, bias fp32 [F] or NULL, graph arrays of a UNIFORM plan (gcrnn_ell_fill_z; uniform_w = the
// one weight), Huser fp32 [B][T][F][N] (or [B][1][F][N] with last_only) or NULL.
extern "C" int gcrnn_fused_forward_x3(const void* xs3, const void* h03, void* hs3, const void* wpack3, const float* bias,
                                      const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                      int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, double uniform_w, void* Huser,
                                      int last_only, void* stream) {
  if (!xs3 || !h03 || !hs3 || !wpack3 || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || uniform_w == 0.0 || !gcrnn_fused_x3_supported(N, F, G, K, entries)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * (F > G ? F : G) * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;         // 32-bit buffer offsets
  if (Huser && (reinterpret_cast<uintptr_t>(Huser) & 15)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_X3_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) \
    return x3_launch<KK, HH, XX>(xs3, h03, hs3, wpack3, bias, tile_nodes, tile_off, ell_col4, entries, B, T, N, (float)uniform_w, (float*)Huser, last_only, st);
  GCRNN_X3_CASE(5, 2, 2) GCRNN_X3_CASE(4, 2, 2) GCRNN_X3_CASE(3, 2, 2) GCRNN_X3_CASE(2, 2, 2)
  GCRNN_X3_CASE(5, 2, 1) GCRNN_X3_CASE(4, 2, 1) GCRNN_X3_CASE(3, 2, 1) GCRNN_X3_CASE(2, 2, 1)
  GCRNN_X3_CASE(5, 1, 1) GCRNN_X3_CASE(4, 1, 1) GCRNN_X3_CASE(3, 1, 1) GCRNN_X3_CASE(2, 1, 1)
#undef GCRNN_X3_CASE
  return GCRNN_ERR_UNSUPPORTED;
}
