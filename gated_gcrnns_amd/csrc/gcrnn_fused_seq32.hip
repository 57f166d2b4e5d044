// 32-feature-chunk sequence-resident forward (gcrnn_fused_seq32.h): weight packing, dispatch, C entry points.
#include "gcrnn_fused_step.h"
#include "gcrnn_fused_seq32.h"

// weights -> per-lane MFMA A fragments (bf16) of the wide kernel:  wpack[chunk32][tap][half][kstep][lane][8]
//   MFMA row m = lane & 15 of (chunk c, half h) is output feature 32 c + 8 (m >> 2) + 4 h + (m & 3) -- lane (r, q) of the D tile then holds
//   features 32 c + 8 q .. + 7 of its node (gcrnn_fused_seq32.h); k = 32 kstep + 8 (lane >> 4) + j over the concatenated [h | x] features.
//   Tap k is scaled by w^k (w = the graph's one weight): the hops then sum the 0/1 pattern, no multiply (Horner t_j = w^j v_j).
//   Fout output rows over F state and G input features: Fout = F for a cell, 2 F for the two time gates' sub-cells stacked (gate pair pre-pass).
template <typename W>
__global__ void pack_weights_wide_kernel(const W* __restrict__ wA, const W* __restrict__ wB, uint16_t* __restrict__ out,
                                         int Fout, int F, int G, int Kin, int Kst, int K, float w) {
  const int KS = (F + G) / 32;
  const int64_t total = (int64_t)(Fout / 32) * K * 2 * KS * 64 * 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 7, lane = (idx >> 3) & 63;
  int64_t rest = idx >> 9;
  const int s = rest % KS; rest /= KS;
  const int h = rest & 1; rest >>= 1;
  const int tap = rest % K;
  const int chunk = rest / K;
  const int m = lane & 15;
  const int f = chunk * 32 + 8 * (m >> 2) + 4 * h + (m & 3);
  const int feat = 32 * s + 8 * (lane >> 4) + j;
  float v = 0.f;
  if (feat < F) { if (tap < Kst) v = (float)wB[((int64_t)f * Kst + tap) * F + feat]; }
  else          { if (tap < Kin) v = (float)wA[((int64_t)f * Kin + tap) * G + (feat - F)]; }
  float sc = 1.f;
  for (int k = 0; k < tap; ++k) sc *= w;
  out[idx] = f2bf(v * sc);
}

extern "C" int gcrnn_fused_pack_weights_wide(int wdtype, const void* wA, const void* wB, void* wpack, int64_t Fout, int64_t F, int64_t G,
                                             int64_t Kin, int64_t Kst, double uniform_w, void* stream) {
  if (!wA || !wB || !wpack) return GCRNN_ERR_NULL_POINTER;
  if (F <= 0 || G < 0 || F % 32 || (F + G) % 32 || Fout <= 0 || Fout % 32 || Kin <= 0 || Kst <= 0 || uniform_w == 0.0) return GCRNN_ERR_BAD_SHAPE;
  const int K = (int)(Kin > Kst ? Kin : Kst);
  const int64_t total = (Fout / 32) * K * 2 * ((F + G) / 32) * 64 * 8;
  GCRNN_PRE_LAUNCH();
  if (wdtype == GCRNN_F32)
    pack_weights_wide_kernel<float><<<(unsigned)cdiv(total, 256), 256, 0, as_stream(stream)>>>(
        (const float*)wA, (const float*)wB, (uint16_t*)wpack, (int)Fout, (int)F, (int)G, (int)Kin, (int)Kst, K, (float)uniform_w);
  else if (wdtype == GCRNN_BF16)
    pack_weights_wide_kernel<__hip_bfloat16><<<(unsigned)cdiv(total, 256), 256, 0, as_stream(stream)>>>(
        (const __hip_bfloat16*)wA, (const __hip_bfloat16*)wB, (uint16_t*)wpack, (int)Fout, (int)F, (int)G, (int)Kin, (int)Kst, K, (float)uniform_w);
  else
    return GCRNN_ERR_BAD_DTYPE;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// Split sequences (one launch per step, F/32 workgroups per sequence): batches between a quarter and half of the chip's CUs, where the
// chunk-parallel kernel needs two of its 64-sequence rounds and the persistent kernel leaves half of the CUs idle
// (profiles/r04_small_batches.txt). GCRNN_SEQ32_SPLIT=0 switches it off, =1 forces it for any B <= 128 (tests).
static bool seq32_split_wanted(int64_t B, int64_t F) {
  const char* off = getenv("GCRNN_SEQ32");
  if (off && off[0] == '0') return false;
  const char* off16 = getenv("GCRNN_SEQ_KERNEL");
  if (off16 && off16[0] == '0') return false;
  if (F != 64 || B > 128) return false;
  const char* sp = getenv("GCRNN_SEQ32_SPLIT");
  if (sp) return sp[0] != '0';
  return B > 64;
}

// GCRNN_SEQ32=0 keeps the 16-feature kernels (A/B); GCRNN_SEQ32_MIN_B=n overrides the batch rule (tests)
static bool seq32_wanted(int64_t B) {
  const char* off = getenv("GCRNN_SEQ32");
  if (off && off[0] == '0') return false;
  const char* mb = getenv("GCRNN_SEQ32_MIN_B");
  if (mb) return B >= (atoi(mb) < 1 ? 1 : atoi(mb));
  const char* off16 = getenv("GCRNN_SEQ_KERNEL");      // (no sequence-resident kernel at all: A/B against the chunk-parallel kernel)
  if (off16 && off16[0] == '0') return false;
  // one workgroup per sequence against the chunk-parallel kernel's one per (sequence, 16-feature chunk): measured at B = 256, K = 5 a round
  // of 256 sequences costs this kernel 0.62 x the time of their 1024 chunk items there (1.70 vs 2.75 ms per forward), so it wins as soon
  // as the batch needs three of the chunk-parallel kernel's rounds: B >= 129 (profiles/r04_small_batches.txt). GCRNN_SEQ_MIN_B, the
  // 16-feature kernel's test override, does not move problems onto this one.
  const double rounds_seq = (double)((B + 255) / 256), rounds_chunk = (double)((B * 4 + 255) / 256);
  return rounds_seq * 0.62 * 4 < rounds_chunk;
}

// The hand-allocated-hop kernel (round 5, gcrnn_fused_seq32p.h: pinned operand / accumulator tuples, tap MFMAs at the stream's tile exits, the
// next operand requested inside the last hop) carries the un-gated persistent forward where it measures faster: the NATIVE layout
// (sequence-major in and out: 1.365 ms against 1.407 ms per forward at the bench size, same box). With the inline pack and the user-layout
// copy it is level with round 4's kernel (1.705 / 1.70 ms), which keeps those. GCRNN_SEQ32P=1 forces it for every un-gated launch, =0 switches
// it off (same-box A/B: profiles/r05_p32_ab.txt).
int gcrnn_seq32p_forward(const Seq32Args& sa, int K, int HS, int XS, bool inline_pack, size_t lds, hipStream_t st);
static bool seq32p_wanted(bool native) {
  const char* e = getenv("GCRNN_SEQ32P");
  if (e) return e[0] != '0';
  return native;
}

template <int K, int HS, int XS>
static size_t seq32_lds(int64_t entries, bool inline_pack, bool r1 = false) { return Seq32Map<K, HS, XS>::lds_bytes(entries, inline_pack, r1); }

static size_t seq32_lds_chain(int64_t F, int64_t K, int64_t entries, bool inline_pack, bool r1 = false) {
#define GCRNN_SEQ32_CASE(KK, HH) if (K == KK && F == 32 * HH) return seq32_lds<KK, HH, 0>(entries, inline_pack, r1);
  GCRNN_SEQ32_CASE(5, 2) GCRNN_SEQ32_CASE(4, 2) GCRNN_SEQ32_CASE(3, 2) GCRNN_SEQ32_CASE(2, 2)
  GCRNN_SEQ32_CASE(5, 1) GCRNN_SEQ32_CASE(4, 1) GCRNN_SEQ32_CASE(3, 1) GCRNN_SEQ32_CASE(2, 1)
#undef GCRNN_SEQ32_CASE
  return 0;
}

static size_t seq32_lds_for(int64_t F, int64_t G, int64_t K, int64_t entries, bool inline_pack, bool r1 = false) {
#define GCRNN_SEQ32_CASE(KK, HH, XX) if (K == KK && F == 32 * HH && G == 32 * XX) return seq32_lds<KK, HH, XX>(entries, inline_pack, r1);
  GCRNN_SEQ32_CASE(5, 2, 2) GCRNN_SEQ32_CASE(4, 2, 2) GCRNN_SEQ32_CASE(3, 2, 2) GCRNN_SEQ32_CASE(2, 2, 2)
  GCRNN_SEQ32_CASE(5, 2, 1) GCRNN_SEQ32_CASE(4, 2, 1) GCRNN_SEQ32_CASE(3, 2, 1) GCRNN_SEQ32_CASE(2, 2, 1)
  GCRNN_SEQ32_CASE(5, 1, 1) GCRNN_SEQ32_CASE(4, 1, 1) GCRNN_SEQ32_CASE(3, 1, 1) GCRNN_SEQ32_CASE(2, 1, 1)
#undef GCRNN_SEQ32_CASE
  return 0;
}

// 1 when gcrnn_fused_forward_wide_bf16 takes this problem (un-gated cell, uniform-weight bf16-image plan, a batch that fills whole rounds
// of the chip, LDS room); inline_pack: with the layout of X inside the launch (N % 8 == 0, T > 2: the caller lays out x_0 and x_1)
extern "C" int gcrnn_fused_forward_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                                  double uniform_w, int img16 /* bit 1: rank-1-weighted graph (two factor tables in LDS) */, int inline_pack) {
  if (uniform_w == 0.0 || !img16 || N <= 0 || N > NP || B <= 0 || T <= 0 || entries <= 0 || entries % 4) return 0;
  if (inline_pack && (N % 8 || T * G * N > 2147483647LL)) return 0;
  if (B * (NP * (F > G ? F : G) * 2) > 2147483647LL || T * F * N > 2147483647LL) return 0;
  if (!seq32_wanted(B) && !((img16 & 2) == 0 && seq32_split_wanted(B, F))) return 0;      // (rank-1 graphs have no split variant)
  return seq32_lds_for(F, G, K, entries, inline_pack != 0, (img16 & 2) != 0) ? 1 : 0;
}

template <int K, int HS, int XS, int VAR, int MODE = 0, bool GATED = false, bool R1 = false>
static int seq32_launch_v(const Seq32Args& sa, size_t lds, hipStream_t st) {
  auto sk = fused_seq32_kernel<K, HS, XS, VAR, MODE, GATED, R1>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  sk<<<(unsigned)(sa.B < gcrnn_persistent_grid() ? sa.B : gcrnn_persistent_grid()), STHREADS, lds, st>>>(sa);      // one workgroup per CU (count read from the device once)
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// Split sequences (65 <= B <= 128, F = 64): one launch per time step, F/32 workgroups per sequence; the host walks the steps and hands every
// launch its step's arrays (operand, output, user-layout block, the next step's input to lay out)
template <int K, int HS, int XS, int VAR, bool GATED = false>
static int seq32_launch_split(const Seq32Args& sa0, size_t lds, int64_t T, int64_t F, int64_t G, int64_t N, hipStream_t st) {
  if constexpr (HS > 1) {
    auto sk = fused_seq32_kernel<K, HS, XS, VAR, 0, GATED, false, true>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GCRNN_ERR_LAUNCH;
    const int64_t xstep = (int64_t)sa0.B * NP * G, hstep = (int64_t)sa0.B * NP * F;
    const unsigned grid = (unsigned)((int64_t)sa0.B * HS < gcrnn_persistent_grid() ? (int64_t)sa0.B * HS : gcrnn_persistent_grid() / HS * HS);
    GCRNN_PRE_LAUNCH();
    for (int64_t t = 0; t < T; ++t) {
      Seq32Args s1 = sa0;
      s1.nsteps = 1;
      s1.x0 = sa0.x0 + t * xstep;
      s1.hfirst = (t == 0) ? sa0.hfirst : sa0.out0 + (t - 1) * hstep;
      s1.out0 = sa0.out0 + t * hstep;
      s1.a1 = !sa0.a1 ? nullptr : (!sa0.a1_last_only ? sa0.a1 + t * F * N : (t == T - 1 ? sa0.a1 : nullptr));
      s1.a1_last_only = 0;
      if (GATED) { s1.gi0 = sa0.gi0 + t * sa0.gstride; s1.gf0 = sa0.gf0 + t * sa0.gstride; }
      const bool pk = sa0.pk_src0 && t + 1 < T;      // this launch lays out x_{t+1} (the caller laid out x_0)
      s1.pk_src0 = pk ? sa0.pk_src0 + (t + 1) * G * N : nullptr;
      s1.pk_dst0 = pk ? sa0.pk_dst0 + (t + 1) * xstep : nullptr;
      sk<<<grid, STHREADS, lds, st>>>(s1);
    }
    GCRNN_CHECK_LAUNCH();
    return GCRNN_OK;
  } else {
    return GCRNN_ERR_UNSUPPORTED;
  }
}

template <int K, int HS, int XS>
static int seq32_launch(const Seq32Args& sa, bool inline_pack, hipStream_t st, bool split = false, int64_t T = 0, int64_t N = 0) {
  const size_t lds = seq32_lds<K, HS, XS>(sa.entries, inline_pack, sa.r1a != nullptr);
  if (!lds) return GCRNN_ERR_UNSUPPORTED;
  if (split) {
    if (sa.r1a) return GCRNN_ERR_UNSUPPORTED;
    if (sa.gi0) {      // time-gated recurrence (the gate pre-pass has laid out X)
      if (inline_pack) return GCRNN_ERR_BAD_SHAPE;
      if (sa.a1) return seq32_launch_split<K, HS, XS, 2, true>(sa, lds, T, 32 * HS, 32 * XS, N, st);
      return seq32_launch_split<K, HS, XS, 0, true>(sa, lds, T, 32 * HS, 32 * XS, N, st);
    }
    const int var = (inline_pack ? 1 : 0) | (sa.a1 ? 2 : 0);
    switch (var) {
      case 0: return seq32_launch_split<K, HS, XS, 0>(sa, lds, T, 32 * HS, 32 * XS, N, st);
      case 1: return seq32_launch_split<K, HS, XS, 1>(sa, lds, T, 32 * HS, 32 * XS, N, st);
      case 2: return seq32_launch_split<K, HS, XS, 2>(sa, lds, T, 32 * HS, 32 * XS, N, st);
      default: return seq32_launch_split<K, HS, XS, 3>(sa, lds, T, 32 * HS, 32 * XS, N, st);
    }
  }
  if (sa.r1a) {      // rank-1-weighted graph: un-gated forward, as the module issues it (3) or sequence-major in and out (0); time-gated recurrence
    if (sa.gi0) {
      if (inline_pack) return GCRNN_ERR_BAD_SHAPE;
      if (sa.a1) return seq32_launch_v<K, HS, XS, 2, 0, true, true>(sa, lds, st);
      return seq32_launch_v<K, HS, XS, 0, 0, true, true>(sa, lds, st);
    }
    if (inline_pack && sa.a1) return seq32_launch_v<K, HS, XS, 3, 0, false, true>(sa, lds, st);
    if (!inline_pack && sa.a1) return seq32_launch_v<K, HS, XS, 2, 0, false, true>(sa, lds, st);
    if (!inline_pack && !sa.a1) return seq32_launch_v<K, HS, XS, 0, 0, false, true>(sa, lds, st);
    return seq32_launch_v<K, HS, XS, 1, 0, false, true>(sa, lds, st);
  }
  if (sa.gi0) {      // time-gated recurrence: the gate pre-pass has laid out X
    if (inline_pack) return GCRNN_ERR_BAD_SHAPE;
    if (sa.a1) return seq32_launch_v<K, HS, XS, 2, 0, true>(sa, lds, st);
    return seq32_launch_v<K, HS, XS, 0, 0, true>(sa, lds, st);
  }
  if (seq32p_wanted(!inline_pack && !sa.a1)) return gcrnn_seq32p_forward(sa, K, HS, XS, inline_pack, lds, st);      // the hand-allocated hop (gcrnn_fused_seq32p.h)
  const int var = (inline_pack ? 1 : 0) | (sa.a1 ? 2 : 0);
  switch (var) {
    case 0: return seq32_launch_v<K, HS, XS, 0>(sa, lds, st);
    case 1: return seq32_launch_v<K, HS, XS, 1>(sa, lds, st);
    case 2: return seq32_launch_v<K, HS, XS, 2>(sa, lds, st);
    default: return seq32_launch_v<K, HS, XS, 3>(sa, lds, st);
  }
}

template <int K, int HS, int XS>
static int seq32_launch_pair(const Seq32Args& sa, bool inline_pack, hipStream_t st) {
  const size_t lds = seq32_lds<K, HS, XS>(sa.entries, inline_pack, sa.r1a != nullptr);
  if (!lds) return GCRNN_ERR_UNSUPPORTED;
  if (sa.r1a) return inline_pack ? seq32_launch_v<K, HS, XS, 1, 1, false, true>(sa, lds, st) : seq32_launch_v<K, HS, XS, 0, 1, false, true>(sa, lds, st);
  return inline_pack ? seq32_launch_v<K, HS, XS, 1, 1>(sa, lds, st) : seq32_launch_v<K, HS, XS, 0, 1>(sa, lds, st);
}

// Whole forward as ONE launch of the wide sequence-resident kernel (reference Utils/graphML.py:2351-2427): un-gated, or -- gi / gf [T][B]
// fp32, the scalar time gates of every step (graphML.py:2357-2374; they read (x_t, h0), so all are known before step 0) -- time-gated.
// xs [T][B][NP][G] bf16 sequence-major (every step laid out, or -- with Xuser_inline = the user-layout X [B][T][G][N] -- steps 0 and 1 only:
// step t lays out x_{t+2}), h0 [B][NP][F], hs [T][B][NP][F] (out), wpack from gcrnn_fused_pack_weights_wide, bias [F] fp32 or NULL,
// plan arrays of the bf16-image plan; Huser [B][T or 1][F][N] bf16 or NULL (huser_last_only: the last step only).
extern "C" int gcrnn_fused_forward_wide_bf16(const void* xs, const void* h0, void* hs, const void* wpack, const float* bias,
                                             const float* gi, const float* gf, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                             int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, void* Huser,
                                             int huser_last_only, const void* Xuser_inline, const float* rank1_a, const float* rank1_b,
                                             void* stream) {
  if (!xs || !h0 || !hs || !wpack || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if ((rank1_a == nullptr) != (rank1_b == nullptr)) return GCRNN_ERR_BAD_SHAPE;
  if ((gi == nullptr) != (gf == nullptr) || (gi && Xuser_inline)) return GCRNN_ERR_BAD_SHAPE;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B > (1 << 24) || entries <= 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  if (B * (NP * (F > G ? F : G) * 2) > 2147483647LL || T * F * N > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;   // 32-bit buffer offsets
  if (Huser && (N % 8 != 0 || (reinterpret_cast<uintptr_t>(Huser) & 15))) return GCRNN_ERR_BAD_SHAPE;
  if (Xuser_inline && (N % 8 != 0 || (reinterpret_cast<uintptr_t>(Xuser_inline) & 15) || T * G * N > 2147483647LL)) return GCRNN_ERR_BAD_SHAPE;
  const int64_t xstep = B * NP * G, hstep = B * NP * F;
  Seq32Args sa{};
  sa.x0 = (const uint16_t*)xs; sa.xstride = xstep;
  sa.hfirst = (const uint16_t*)h0;
  sa.out0 = (uint16_t*)hs; sa.ostride = hstep;
  sa.wpack = (const uint4*)wpack; sa.bias = bias;
  sa.a1 = (const uint16_t*)Huser; sa.a1stride = F * N; sa.a1_last_only = huser_last_only ? 1 : 0;
  sa.ubstride = (int)((huser_last_only ? 1 : T) * F * N);
  sa.tile_nodes = tile_nodes; sa.tile_off = tile_off; sa.ell_col4 = (const uint2*)ell_col4;
  sa.entries = (int)entries; sa.B = (int)B; sa.N = (int)N;
  sa.nsteps = (int)T;
  sa.gi0 = gi; sa.gf0 = gf; sa.gstride = B;
  sa.r1a = rank1_a; sa.r1b = rank1_b;
  {
    // de-synchronised starts pay when the launch is long enough and every CU has a sequence (GCRNN_SEQ32_STAGGER=cycles overrides, 0 = off)
    const char* sg = getenv("GCRNN_SEQ32_STAGGER");
    sa.stagger = sg ? atoi(sg) : 0;
  }
  const bool inline_pack = Xuser_inline != nullptr && T > 2;
  if (inline_pack) {      // (the kernel lays out steps 2 .. T-1, each finished one hop before the step that reads it ends: gcrnn_fused_seq32.h)
    sa.pk_src0 = (const uint16_t*)Xuser_inline; sa.pksrc_stride = G * N;
    sa.pk_dst0 = const_cast<uint16_t*>(sa.x0); sa.pkdst_stride = xstep;
    sa.pk_stride = (int)(T * G * N);
  }
  hipStream_t st = as_stream(stream);
  // split sequences: the plain un-gated forward of a batch that would leave half of the chip idle (and is not forced onto the persistent form)
  const bool split = !rank1_a && !seq32_wanted(B) && seq32_split_wanted(B, F);
  const bool pack_split = Xuser_inline != nullptr && T > 1;      // (one launch per step: launch t lays out x_{t+1}; the caller laid out x_0)
  if (split && pack_split && !inline_pack) {
    sa.pk_src0 = (const uint16_t*)Xuser_inline; sa.pksrc_stride = G * N;
    sa.pk_dst0 = const_cast<uint16_t*>(sa.x0); sa.pkdst_stride = xstep;
    sa.pk_stride = (int)(T * G * N);
  }
#define GCRNN_SEQ32_CASE(KK, HH, XX) if (K == KK && F == 32 * HH && G == 32 * XX) return seq32_launch<KK, HH, XX>(sa, split ? pack_split : inline_pack, st, split, T, N);
  GCRNN_SEQ32_CASE(5, 2, 2) GCRNN_SEQ32_CASE(4, 2, 2) GCRNN_SEQ32_CASE(3, 2, 2) GCRNN_SEQ32_CASE(2, 2, 2)
  GCRNN_SEQ32_CASE(5, 2, 1) GCRNN_SEQ32_CASE(4, 2, 1) GCRNN_SEQ32_CASE(3, 2, 1) GCRNN_SEQ32_CASE(2, 2, 1)
  GCRNN_SEQ32_CASE(5, 1, 1) GCRNN_SEQ32_CASE(4, 1, 1) GCRNN_SEQ32_CASE(3, 1, 1) GCRNN_SEQ32_CASE(2, 1, 1)
#undef GCRNN_SEQ32_CASE
  return GCRNN_ERR_UNSUPPORTED;
}


// The two time gates' pre-pass as ONE launch over all (t, b) items (reference Utils/graphML.py:2357-2374): item i's operand (x_t, h0 of its
// sequence) is loaded -- and with x_user laid out -- once for BOTH gates, whose sub-cells run as one cell of 2 F outputs on the wide kernel
// (wpack = gcrnn_fused_pack_weights_wide(Fout = 2 F) of [GFL_in ; GFL_forget] stacked over the output features, bias2 [2 F], gw2 [2][N][F] fp32
// the two read-outs' weights node-major). parts [T*B][2 * F/32 * 8] fp32: partial dot products <tanh(pre), read-out weights> per (chunk,
// wave), chunks 0 .. F/32-1 the input gate's -- the caller adds them in a fixed order. cs_in / cs_f (both or neither): the sub-cells' states
// [T][B][NPad][F] bf16 (their BPTT). x_user (or NULL): the user-layout X [B][T][G][N]; the caller has laid out the first
// gcrnn_fused_gate_pair_wide_supported(..., 1) time steps of xs, the items lay out the rest. h0_zero_flag as in gcrnn_fused_gate_prepass_bf16.
extern "C" int gcrnn_fused_gate_pair_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                                    double uniform_w, int img16, int with_pack) {
  if (uniform_w == 0.0 || !img16 || N <= 0 || N > NP || B <= 0 || T <= 0 || entries <= 0 || entries % 4) return 0;
  if (with_pack && (N % 8 || T * G * N > 2147483647LL)) return 0;
  if (B * T * (NP * (F > G ? F : G) * 2) > 2147483647LL) return 0;      // 32-bit buffer offsets over all items
  if (!seq32_wanted(B * T)) return 0;
  if (!seq32_lds_for(F, G, K, entries, with_pack != 0, (img16 & 2) != 0)) return 0;      // (img16 bit 1: rank-1-weighted graph)
  if (!with_pack) return 1;
  const int64_t first = B * T < gcrnn_persistent_grid() ? B * T : gcrnn_persistent_grid();      // the items of the first round of workgroups
  return (int)((first + B - 1) / B);
}

static int gate_pair_prepass_impl(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias2,
                                  const float* gw2, float* parts, const void* tapf, float* taps_out, int64_t ntaps, void* cs_in, void* cs_f,
                                  const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T,
                                  int64_t N, int64_t F, int64_t G, int64_t K, const int32_t* h0_zero_flag,
                                  const float* rank1_a, const float* rank1_b, void* stream) {
  if (!xs || !h0 || !wpack || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (tapf ? (!taps_out || ntaps <= 0 || ntaps > 16 || (reinterpret_cast<uintptr_t>(tapf) & 15)) : (!gw2 || !parts)) return GCRNN_ERR_NULL_POINTER;
  if ((rank1_a == nullptr) != (rank1_b == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if ((cs_in == nullptr) != (cs_f == nullptr)) return GCRNN_ERR_NULL_POINTER;
  const int64_t items = B * T;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || items > (1 << 24) || entries <= 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  if (items * (NP * (F > G ? F : G) * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (x_user && (N % 8 != 0 || (reinterpret_cast<uintptr_t>(x_user) & 15) || T * G * N > 2147483647LL)) return GCRNN_ERR_BAD_SHAPE;
  Seq32Args sa{};
  sa.x0 = (const uint16_t*)xs;
  sa.hfirst = (const uint16_t*)h0; sa.hmod = (int)B;
  sa.out0 = (uint16_t*)cs_in; sa.out1 = (uint16_t*)cs_f;
  sa.wpack = (const uint4*)wpack; sa.bias = bias2;
  sa.tile_nodes = tile_nodes; sa.tile_off = tile_off; sa.ell_col4 = (const uint2*)ell_col4;
  sa.entries = (int)entries; sa.B = (int)items; sa.N = (int)N;
  sa.nsteps = 1;
  sa.flags = h0_zero_flag; sa.gw = gw2; sa.go = parts;
  sa.tapf = (const uint4*)tapf; sa.taps_out = taps_out; sa.ntaps = (int)ntaps;
  sa.r1a = rank1_a; sa.r1b = rank1_b;
  const bool inline_pack = x_user != nullptr;
  if (inline_pack) {
    sa.pk_src0 = (const uint16_t*)x_user; sa.pksrc_stride = G * N; sa.pk_stride = (int)(T * G * N);
    sa.pk_dst0 = (uint16_t*)xs;
  }
  hipStream_t st = as_stream(stream);
#define GCRNN_SEQ32_CASE(KK, HH, XX) if (K == KK && F == 32 * HH && G == 32 * XX) return seq32_launch_pair<KK, HH, XX>(sa, inline_pack, st);
  GCRNN_SEQ32_CASE(5, 2, 2) GCRNN_SEQ32_CASE(4, 2, 2) GCRNN_SEQ32_CASE(3, 2, 2) GCRNN_SEQ32_CASE(2, 2, 2)
  GCRNN_SEQ32_CASE(5, 2, 1) GCRNN_SEQ32_CASE(4, 2, 1) GCRNN_SEQ32_CASE(3, 2, 1) GCRNN_SEQ32_CASE(2, 2, 1)
  GCRNN_SEQ32_CASE(5, 1, 1) GCRNN_SEQ32_CASE(4, 1, 1) GCRNN_SEQ32_CASE(3, 1, 1) GCRNN_SEQ32_CASE(2, 1, 1)
#undef GCRNN_SEQ32_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

extern "C" int gcrnn_fused_gate_pair_prepass_wide_bf16(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias2,
                                                       const float* gw2, float* parts, void* cs_in, void* cs_f, const int32_t* tile_nodes,
                                                       const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T,
                                                       int64_t N, int64_t F, int64_t G, int64_t K, const int32_t* h0_zero_flag,
                                                       const float* rank1_a, const float* rank1_b, void* stream) {
  if (!gw2 || !parts) return GCRNN_ERR_NULL_POINTER;
  return gate_pair_prepass_impl(x_user, xs, h0, wpack, bias2, gw2, parts, nullptr, nullptr, 0, cs_in, cs_f, tile_nodes, tile_off, ell_col4, entries, B, T,
                                N, F, G, K, h0_zero_flag, rank1_a, rank1_b, stream);
}

// The same launch for the NODE gates (Utils/graphML.py:2379-2393): both gate cells of every (t, b), and instead of a read-out the first stage of
// their F -> 1 graph filters (GFL_node_*, :2303, 2318; taps first, :2387): per-tap dot products of the gate cell's state on the matrix cores.
// tapf [2 gates][F/32][3 planes][64 lanes] x 16 B: the taps' A fragments (three bf16 planes, p0 + p1 + p2 = w to 24 bits; lane 16 kg + tap holds
// w_p[tap][32 cg + 8 kg .. + 7], taps >= ntaps zero); taps_out [T*B][2][F/32][ntaps][N] fp32: the partial dots of each 32-feature chunk (the caller
// adds a gate's chunks in a fixed order, then runs the K - 1 one-channel hops and the sigmoid). cs_in / cs_f as above (training).
extern "C" int gcrnn_fused_gate_pair_prepass_taps_wide_bf16(const void* x_user, void* xs, const void* h0, const void* wpack, const float* bias2,
                                                            const void* tapf, float* taps_out, int64_t ntaps, void* cs_in, void* cs_f,
                                                            const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4,
                                                            int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K,
                                                            const int32_t* h0_zero_flag, const float* rank1_a, const float* rank1_b, void* stream) {
  if (!tapf || !taps_out) return GCRNN_ERR_NULL_POINTER;
  return gate_pair_prepass_impl(x_user, xs, h0, wpack, bias2, nullptr, nullptr, tapf, taps_out, ntaps, cs_in, cs_f, tile_nodes, tile_off, ell_col4, entries,
                                B, T, N, F, G, K, h0_zero_flag, rank1_a, rank1_b, stream);
}


// dpre = dH (1 - h^2) on bf16 arrays: the seed of the chain (t = T-1)
__global__ void seq32_seed_kernel(const uint16_t* __restrict__ dH, const uint16_t* __restrict__ h, uint16_t* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= n) return;
  const uint32_t g = *reinterpret_cast<const uint32_t*>(dH + i), hv = *reinterpret_cast<const uint32_t*>(h + i);
  const float g0 = bf2f((uint16_t)(g & 0xffffu)), g1 = bf2f((uint16_t)(g >> 16));
  const float h0 = bf2f((uint16_t)(hv & 0xffffu)), h1 = bf2f((uint16_t)(hv >> 16));
  *reinterpret_cast<uint32_t*>(out + i) = pack2bf(g0 * (1.f - h0 * h0), g1 * (1.f - h1 * h1));
}

// split sequences (65 <= B <= 128, F = 64): one launch per chain step, F/32 workgroups per sequence (as seq32_launch_split)
template <int K, int HS, int VAR>
static int seq32_launch_chain_split(const Seq32Args& sa0, size_t lds, int64_t T, int64_t F, int64_t N, bool fin, hipStream_t st) {
  if constexpr (HS > 1) {
    auto sk = fused_seq32_kernel<K, HS, 0, VAR, 2, false, false, true>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GCRNN_ERR_LAUNCH;
    const unsigned grid = (unsigned)((int64_t)sa0.B * HS < gcrnn_persistent_grid() ? (int64_t)sa0.B * HS : gcrnn_persistent_grid() / HS * HS);
    const int64_t hstep = (int64_t)sa0.B * NP * F;
    GCRNN_PRE_LAUNCH();
    for (int64_t i = 0; i + 1 < T; ++i) {      // chain step i: t = T-1-i
      Seq32Args s1 = sa0;
      s1.nsteps = 1; s1.final_raw = 0;
      s1.hfirst = (i == 0) ? sa0.hfirst : sa0.out0 + (i - 1) * sa0.ostride;
      s1.out0 = sa0.out0 + i * sa0.ostride;
      s1.dh0_ = sa0.dh0_ + i * sa0.dhstride; s1.hs0 = sa0.hs0 + i * sa0.hsstride;
      s1.gsc0 = sa0.gsc0 ? sa0.gsc0 + i * sa0.gscstride : nullptr;
      s1.gpart0 = sa0.gpart0 ? sa0.gpart0 + i * sa0.gpartstride : nullptr;
      const bool pk = sa0.pk_src0 && i + 2 < T;      // this launch lays out the upstream gradient of chain step i + 1
      s1.pk_src0 = pk ? sa0.pk_src0 + (i + 1) * sa0.pksrc_stride : nullptr;
      s1.pk_dst0 = pk ? sa0.pk_dst0 + (i + 1) * sa0.pkdst_stride : nullptr;
      sk<<<grid, STHREADS, lds, st>>>(s1);
    }
    if (fin) {      // d h0 / the forget gate's step 0: operand dpre_0
      Seq32Args s1 = sa0;
      s1.nsteps = 1; s1.final_raw = 1;
      s1.hfirst = (T >= 2) ? sa0.out0 + (T - 2) * sa0.ostride : sa0.hfirst;
      s1.gsc0 = sa0.gsc0 ? sa0.gsc0 + (T - 1) * sa0.gscstride : nullptr;
      s1.gpart0 = sa0.gpart0 ? sa0.gpart0 + (T - 1) * sa0.gpartstride : nullptr;
      s1.pk_src0 = nullptr; s1.pk_dst0 = nullptr;
      sk<<<grid, STHREADS, lds, st>>>(s1);
    }
    GCRNN_CHECK_LAUNCH();
    return GCRNN_OK;
  } else {
    return GCRNN_ERR_UNSUPPORTED;
  }
}

template <int K, int HS>
static int seq32_launch_chain(const Seq32Args& sa, bool inline_pack, hipStream_t st, bool split = false, int64_t T = 0, int64_t N = 0, bool fin = false) {
  const size_t lds = seq32_lds<K, HS, 0>(sa.entries, inline_pack, sa.r1a != nullptr);
  if (!lds) return GCRNN_ERR_UNSUPPORTED;
  if (sa.r1a) {      // rank-1-weighted graph (adjoint plan of its 0/1 pattern, factors swapped): the persistent chain only
    if (split) return GCRNN_ERR_UNSUPPORTED;
    return inline_pack ? seq32_launch_v<K, HS, 0, 1, 2, false, true>(sa, lds, st) : seq32_launch_v<K, HS, 0, 0, 2, false, true>(sa, lds, st);
  }
  if (split)
    return inline_pack ? seq32_launch_chain_split<K, HS, 1>(sa, lds, T, 32 * HS, N, fin, st) : seq32_launch_chain_split<K, HS, 0>(sa, lds, T, 32 * HS, N, fin, st);
  return inline_pack ? seq32_launch_v<K, HS, 0, 1, 2>(sa, lds, st) : seq32_launch_v<K, HS, 0, 0, 2>(sa, lds, st);
}

// 1 when gcrnn_fused_backward_data_wide_bf16 takes the problem (uniform-weight bf16-image plan of the ADJOINT graph, a batch that fills whole
// rounds of the chip, LDS room; inline_pack: with dHuser_inline; img16 bit 1: a rank-1-weighted graph on the plan of its pattern)
extern "C" int gcrnn_fused_backward_data_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, int64_t entries, double uniform_w,
                                                        int img16, int inline_pack) {
  if (uniform_w == 0.0 || !img16 || N <= 0 || N > NP || B <= 0 || T <= 0 || entries <= 0 || entries % 4) return 0;
  if (inline_pack && (N % 8 || T * F * N > 2147483647LL)) return 0;
  if (B * (NP * F * 2) > 2147483647LL) return 0;
  const bool r1 = (img16 & 2) != 0;      // bit 1: rank-1-weighted graph (no split variant)
  if (!seq32_wanted(B) && !(!r1 && seq32_split_wanted(B, F))) return 0;
  return seq32_lds_chain(F, K, entries, inline_pack != 0, r1) ? 1 : 0;
}

// BPTT data gradient of the fused cell as ONE launch of the wide sequence-resident kernel (the adjoint of Utils/graphML.py:2420-2423; contract
// of gcrnn_fused_backward_data_bf16): dpre[T-1] = dHs[T-1] (1 - hs[T-1]^2); for t = T-1 .. 1: dpre[t-1] = (gf[t] sum_k (S)^k (dpre[t] B_k) +
// dHs[t-1]) (1 - hs[t-1]^2); dh0 = gf[0] sum_k (S)^k (dpre[0] B_k) (optional). wpackT = gcrnn_fused_pack_weights_wide of the TRANSPOSED state taps
// (G = 0, uniform_w of the adjoint plan); tile_nodes / tile_off / ell_col4: the bf16-image plan of the ADJOINT graph. dgf_parts (or NULL; needs
// h0s): [T][B][F/32*8] fp32 partials of <h_{t-1}, adjoint chain of dpre[t]>. dHuser_inline (or NULL): dH [B][T][F][N] bf16 in the user layout
// -- the caller has laid out dHs[T-1] and dHs[T-2], the launch lays out the rest.
extern "C" int gcrnn_fused_backward_data_wide_bf16(const void* dHs, const void* hs, void* dpre, void* dh0, const void* wpackT,
                                                   const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                                   int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, const float* gf, const void* h0s,
                                                   float* dgf_parts, const void* dHuser_inline, const float* rank1_a,
                                                   const float* rank1_b, void* stream) {
  if (dgf_parts && !h0s) return GCRNN_ERR_NULL_POINTER;
  if ((rank1_a == nullptr) != (rank1_b == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (dHuser_inline && (N % 8 != 0 || (reinterpret_cast<uintptr_t>(dHuser_inline) & 15) || T * F * N > 2147483647LL)) return GCRNN_ERR_BAD_SHAPE;
  if (!dHs || !hs || !dpre || !wpackT || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B > (1 << 24) || entries <= 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  if (B * (NP * F * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  const int64_t hstep = B * NP * F;
  hipStream_t st = as_stream(stream);
  GCRNN_PRE_LAUNCH();
  seq32_seed_kernel<<<(unsigned)cdiv(hstep / 2, 256), 256, 0, st>>>((const uint16_t*)dHs + (T - 1) * hstep, (const uint16_t*)hs + (T - 1) * hstep,
                                                                 (uint16_t*)dpre + (T - 1) * hstep, hstep);
  GCRNN_CHECK_LAUNCH();
  const bool fin = dh0 != nullptr || dgf_parts != nullptr;
  if (T < 2 && !fin) return GCRNN_OK;
  const int nch = (int)(F / 32);
  const int64_t gstep = B * (nch * SWAVES);
  Seq32Args sa{};
  sa.hfirst = (const uint16_t*)dpre + (T - 1) * hstep;
  sa.out0 = (uint16_t*)dpre + (T - 2) * hstep; sa.ostride = -hstep;
  sa.dh0_ = (const uint16_t*)dHs + (T - 2) * hstep; sa.dhstride = -hstep;
  sa.hs0 = (const uint16_t*)hs + (T - 2) * hstep; sa.hsstride = -hstep;
  sa.gsc0 = gf ? gf + (T - 1) * B : nullptr; sa.gscstride = -B;
  sa.gpart0 = dgf_parts ? dgf_parts + (T - 1) * gstep : nullptr; sa.gpartstride = -gstep;
  sa.final_raw = fin ? 1 : 0; sa.final_out = (uint16_t*)dh0; sa.final_h = dgf_parts ? (const uint16_t*)h0s : nullptr;
  sa.wpack = (const uint4*)wpackT;
  sa.tile_nodes = tile_nodes; sa.tile_off = tile_off; sa.ell_col4 = (const uint2*)ell_col4;
  sa.entries = (int)entries; sa.B = (int)B; sa.N = (int)N;
  sa.r1a = rank1_a; sa.r1b = rank1_b;      // (rank-1-weighted graph: the factors of S^T -- a and b of the forward direction swapped)
  sa.nsteps = (int)(T - 1) + (fin ? 1 : 0);
  const bool inline_pack = dHuser_inline != nullptr && T > 2;
  if (inline_pack) {      // chain step i reads dHs[T-2-i]: step 0's is the caller's, step i lays out step i + 1's
    sa.pk_src0 = (const uint16_t*)dHuser_inline + (T - 2) * F * N; sa.pksrc_stride = -(F * N);
    sa.pk_dst0 = const_cast<uint16_t*>((const uint16_t*)dHs) + (T - 2) * hstep; sa.pkdst_stride = -hstep;
    sa.pk_stride = (int)(T * F * N);
  }
  const bool split = !rank1_a && !seq32_wanted(B) && seq32_split_wanted(B, F);
#define GCRNN_SEQ32_CASE(KK, HH) if (K == KK && F == 32 * HH) return seq32_launch_chain<KK, HH>(sa, inline_pack, st, split, T, N, fin);
  GCRNN_SEQ32_CASE(5, 2) GCRNN_SEQ32_CASE(4, 2) GCRNN_SEQ32_CASE(3, 2) GCRNN_SEQ32_CASE(2, 2)
  GCRNN_SEQ32_CASE(5, 1) GCRNN_SEQ32_CASE(4, 1) GCRNN_SEQ32_CASE(3, 1) GCRNN_SEQ32_CASE(2, 1)
#undef GCRNN_SEQ32_CASE
  return GCRNN_ERR_UNSUPPORTED;
}
