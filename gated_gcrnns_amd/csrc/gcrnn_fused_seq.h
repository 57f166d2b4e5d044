// Sequence-resident fused GCRNN recurrence for gfx950 (round 3): ONE workgroup owns a WHOLE sequence -- for every time step.
//
// The chunk-parallel step kernel (gcrnn_fused_step.h) gives every 16-feature output chunk of a sequence its own workgroup, and each
// of them pulls the complete operand [h_{t-1} | x_t] of the sequence (256 KB at F = G = 64) through its CU's L2 path: 4x redundant,
// 268 MB per launch at B = 256, the "operand transfer" third of that launch. It has to: its taps u_0..u_{K-2} live in
// registers in fp32 (128 VGPRs per lane), which leaves no room to keep the operand.
// Here the registers hold the OPERAND instead of the taps: 8 node tiles x (F+G)/32 k-steps x 16 bytes per lane = the same 128
// VGPRs, but they serve all F/16 chunks. The workgroup walks the chunks one after the other; for chunk c the tap a hop needs is
// evaluated from the resident operand right before that hop (u_k = W_k(c) [h|x]^T on the matrix cores, the SAME 32 MFMAs per tap
// and wave, only later), so per chunk only one tap (32 VGPRs) is live. The operand crosses L2 -> CU once per sequence and step:
// the compulsory 256 KB. Arithmetic and its order are those of the chunk-parallel kernel (tap chains over k-steps, hop sums on
// the matrix cores over the bf16 image, fp32 accumulators, bias / tanh / bf16 epilogue): results are bit-identical (tested).
//
// Uniform-weight graphs on the bf16 hop image only (UNI == 2 in the step kernel's terms: GCRNN_HOP_ASM_UNI16_STREAM and the
// plan arrays of graph.fused_plan(img16=True)); other graphs keep the chunk-parallel kernel.
// LDS: hop image / transposed output tile 33 KB | weight fragments of the current and the next chunk 2 x K*KS KB (the next
// chunk's arrive by LDS-DMA while the last hop runs) | column words 32 B x entries | inline-pack tile (G or F) x 256 x 2 B.
//
// Persistent over time: a sequence's steps depend on nothing but that sequence, so ONE launch runs all T steps (the BPTT chain: all
// T-1) -- the workgroup stores h_t, waits for its own stores (s_waitcnt vmcnt(0) + barrier: same CU, same L1) and reads it back as the
// next step's operand. Graph image and tile tables are staged once per launch instead of once per step, there are no launch
// boundaries inside a forward, and the hipGraph of the T-loop is a single kernel node.
//
// MODE 0: forward step   h_t = tanh(sum_k S^k([h|x] W_k) + 2b)            (reference Utils/graphML.py:2420-2423)
// MODE 2: BPTT data step dpre_{t-1} = (gsc * sum_k S^k(dpre_t W_k^T) + dH_{t-1}) (1 - h_{t-1}^2)   (autograd of the same lines)
// MODE 0 with GATED: the time-gated step  h_t = tanh(gi_t (A(S)x_t + b) + gf_t (B(S)h_{t-1} + b)),  gates = scalars per (t, sequence)
//         (graphML.py:2357-2374): h-chain, scale by gf/gi, x-chain, scale by gi -- one accumulator chain, as in the chunk-parallel kernel.
// MODE 1: time-gate pre-pass (graphML.py:2362-2366), all T*B items of one gate in one launch ("step" = none, the items are the
//         batch): c = tanh(A_g(S)x_t + B_g(S)h0 + 2 b_g) per item, gate logit partials sum_{n,f} c[n][f] w[n][f] per (chunk, wave)
//         (fixed-order sum by the caller); c stored bf16 sequence-major when asked (the gate's BPTT, the node gates' filters);
//         with flags[0] != 0 (h0 all zeros: every training loop of the reference, train_rnn.py:256) the state half of the
//         operand is neither loaded nor multiplied.
#pragma once
#include <utility>
#include <type_traits>

// In-kernel phase stamps (diagnostic builds only, -DGCRNN_SEQ_STAMPS; tools/seq_stamps.py): wave 0 of every workgroup records
// s_memtime at its phase boundaries into a buffer of its own that no other code reads (MI355X_MICROARCH.md, DVFS item 6).
#if defined(GCRNN_SEQ_STAMPS) && defined(GCRNN_SEQ_STAMPS_READER)      // (the K = 5 translation unit only)
static __device__ unsigned long long gcrnn_seq_stamps[256 * 64];      // one copy per translation unit; the K = 5 unit exports the reader
// (staged in the 512 spare bytes behind the transposed output tile -- an LDS address needs no scalar registers -- and copied out at the end)
#define GCRNN_STAMP(slot)                                                                                     \
  do {                                                                                                        \
    if (tid == 0 && stamp_on) *reinterpret_cast<volatile unsigned long long*>(smem + 33280 + 8 * (slot)) = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define GCRNN_STAMP_FLUSH()                                                                                   \
  do {                                                                                                        \
    __syncthreads();                                                                                          \
    if (tid < 64 && blockIdx.x < 256) gcrnn_seq_stamps[blockIdx.x * 64 + tid] = *reinterpret_cast<volatile unsigned long long*>(smem + 33280 + 8 * tid); \
  } while (0)
#if 1
extern "C" int gcrnn_debug_read_seq_stamps(void* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(gcrnn_seq_stamps), sizeof(unsigned long long) * 256 * 64) == hipSuccess ? 0 : 1;
}
#endif
#else
#define GCRNN_STAMP(slot) do {} while (0)
#define GCRNN_STAMP_FLUSH() do {} while (0)
#endif

struct SeqArgs {
  const uint16_t* x0; int64_t xstride;                 // MODE 0: x of step 0 [B][NP][G] bf16, elements between steps
  const uint16_t* hfirst; const uint16_t* hrest; int64_t hstride;   // operand h_{t-1} (MODE 2: dpre_t): step 0 = hfirst, step i > 0 = hrest + (i - 1) hstride
  uint16_t* out0; int64_t ostride;                     // output of step 0 [B][NP][F] (or null), elements between steps
  const uint4* wpack;                                  // [F/16][K][KS][64] x 16 B
  const float* bias;                                   // [F] or null
  const float* gf0; int64_t gfstride;                  // MODE 2: forget gates [B] of the step back-propagated, or null; GATED: forget gates of step 0
  const float* gi0;                                    // GATED: input gates [B] of step 0 (stride gfstride)
  const float* gw;                                     // MODE 1: the gate read-out's weights, node-major [N][F] fp32
  const int32_t* flags;                                // MODE 1 (or null): flags[0] != 0 = the state operand h0 is all zeros
  int hmod;                                            // MODE 1: item i reads the state operand of sequence i % hmod (h0 of its sequence); else 0
  const int32_t* tile_nodes; const int32_t* tile_off; const uint2* ell_col4;      // bf16-image plan
  float* go0; int64_t gostride;                        // MODE 2 (or null): [B][F/16 * 8] partials of <h_{t-1}, adjoint chain of dpre_t>
  const uint16_t* a0; int64_t a0stride;                // MODE 2: upstream gradient dH_{t-1} [B][NP][F] (or null)
  const uint16_t* a1; int64_t a1stride;                // MODE 2: state h_{t-1} (or null); MODE 0: user-layout output H[0][t] (or null)
  int a1_last_only;                                    // MODE 0: only the last step writes the user-layout output (at a1 itself)
  int ubstride;                                        // MODE 0: elements between consecutive sequences of the user-layout output
  int entries, B, N; float uni_w;
  const uint16_t* pk_src0; int64_t pksrc_stride;       // inline pack of the NEXT step's operand: user-layout block of sequence 0 for step 0 (or null)
  uint16_t* pk_dst0; int64_t pkdst_stride;             // ... the sequence-major array [B][NP][rows] it is laid out into
  int pk_stride;                                       // ... elements between consecutive sequences of the user-layout tensor
  const uint2* tapf; float* taps_out; int ntaps;       // MODE 1 (or null): fused F -> 1 tap dots of the gate cell's state (node gates, graphML.py:2387): A fragments
                                                       // [F/16][3 planes][64 lanes] x 8 B of the taps, output [items][ntaps][N] fp32
  int sacc_off, tapf_off;                              // ... LDS byte offsets of the per-item accumulators [ntaps][NP] fp32 and of the staged fragments
  const float* ng0; int64_t ngstride;                  // MODE 5: node gates of step 0 [2][B][N] fp32 (input gates, forget gates), elements between steps
  uint16_t* yh0; int64_t yhstride;                     // MODE 5 (or null): receives Yh_t = B(S)h_{t-1} + b [B][NP][F] bf16 (training keeps it)
  int nsteps;                                          // steps of this launch; every step but the last lays out the next one's operand
  int pk_all;                                          // ... != 0: the last step too (per-step launches: the host decides)
  int xprefetch;                                       // MODE 0, experiment: request the x half of the next step's operand during the last hop (needs pk_ahead >= 2 or a fully packed X)
  int pk_ahead;                                        // the inline pack of step i lays out the operand of step i + pk_ahead (1; forward: 2, so that x_{t+1} is complete -- and can be
                                                       // requested into the dead operand registers -- while step t still runs); 0 = 1
};

// f(integral_constant<1>), f(integral_constant<2>), ... : the hops of a chunk with compile-time indices
template <class Fn, int... J>
__device__ __forceinline__ void seq_static_for(Fn&& f, std::integer_sequence<int, J...>) {
  (f(std::integral_constant<int, J + 1>{}), ...);
}

template <int K, int HS, int XS, int MODE, bool GATED = false>
__global__ __launch_bounds__(STHREADS) void fused_seq_kernel(const SeqArgs a) {
  constexpr int KS = HS + XS;
  constexpr int F = 32 * HS, G = 32 * XS;
  constexpr int NCH = F / FC;
  constexpr int HT = STILES;
  constexpr int PKROWS = (MODE == 0 || MODE == 1) ? G : F;
  // LDS map: [0, 33792) hop image A [NP][16] bf16 (32 KB) / the transposed output tile (8 x 4128 B), bias at 33024, flags and stamps at 33280;
  // [33792, 66560) hop image B = the inline-pack tile (free during the hops that read it: see the hop loop); then 2 x WB weight fragments,
  // the column words, MODE 1's tap accumulators. The hops alternate between the two images: a wave writes its tiles of hop j into the
  // image hop j + 1 reads as soon as ITS stream is done, and one barrier per hop hands over (one image needs two: done reading, done writing)
  constexpr int IMGB = GCRNN_HOP_IMAGE_B_OFFSET;
  constexpr int IMG = IMGB + 32 * 1024;          // end of the two images = start of the weight fragments
  static_assert(IMGB >= 33 * 1024 && IMGB % 1024 == 0, "image B sits behind image A / the transposed tile (33024 B) / bias (256 B) / flags and stamps (512 B)");
  constexpr int WB = K * KS * 1024;              // one chunk's weight fragments
  static_assert(STILES == 8 && GCRNN_HOP_ASM, "generated hop stream: 8 tiles per wave");
  static_assert((MODE == 0 || MODE == 1) ? XS > 0 : ((MODE == 2 || MODE == 5) ? XS == 0 : MODE == 4), "forward / pre-pass take [h | x], the BPTT and the node-gated step their one operand, the filter-output pass either");
  static_assert(!GATED || MODE == 0, "gated steps are forward steps");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* state = reinterpret_cast<float*>(smem);
  const int entries = a.entries, B = a.B, N = a.N;
  const float uni_w = a.uni_w;
  char* xtile = smem + IMGB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  if ((int)blockIdx.x >= B) return;
  [[maybe_unused]] bool stamp_on = true;      // (diagnostic builds: which step's phases are recorded)
  GCRNN_STAMP(0);

  // once per launch: tile tables, and -- by LDS-DMA, all pieces in flight together -- the column image and chunk 0's weights
  int tbeg[STILES], tend[STILES];
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
    tbeg[i] = a.tile_off[wave * STILES + i];
    tend[i] = a.tile_off[wave * STILES + i + 1];
  }
  int woff[STILES];      // node << 16 | row16 << 5 | hswz << 4, xor this lane's (half, piece)
#pragma unroll
  for (int i = 0; i < STILES; ++i)
    woff[i] = a.tile_nodes[(wave * STILES + i) * 16 + r] ^ (((q >> 1) << 4) | ((q & 1) << 3));
  {
    const int cbytes = entries * 32;
    const char* csrc = reinterpret_cast<const char*>(a.ell_col4);
    for (int p = wave; p * 1024 < cbytes; p += SWAVES)
      if (p * 1024 + lane * 16 < cbytes)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(csrc + p * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + IMG + 2 * WB + p * 1024), 16, 0, 0);
    const char* wsrc = reinterpret_cast<const char*>(a.wpack);
    for (int p = wave; p < WB / 1024; p += SWAVES)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + p * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + IMG + p * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // the bias (it enters through both filters, graphML.py:2420-2421: scaled by gi + gf = 2 at its use) in the 256 spare bytes behind the tile:
  // a chunk reads its four values from LDS instead of waiting for a global load at the top of every chunk
  if constexpr (MODE == 1) {
    if (a.tapf) {      // (wave-uniform) tap fragments into LDS, accumulators to zero
      for (int idx = tid; idx < NCH * 3 * 64; idx += STHREADS) reinterpret_cast<uint2*>(smem + a.tapf_off)[idx] = a.tapf[idx];
      for (int idx = tid; idx < a.ntaps * NP; idx += STHREADS) reinterpret_cast<float*>(smem + a.sacc_off)[idx] = 0.f;
    }
  }
  // zeros behind the column image: the summing stream's running column pointer is not clamped (GCRNN_HOP_COLUMN_PAD)
  if (tid < GCRNN_HOP_COLUMN_PAD / 4) reinterpret_cast<uint32_t*>(smem + IMG + 2 * WB + entries * 32)[tid] = 0u;
  float* lbias = reinterpret_cast<float*>(smem + 33024);
  if ((MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5) && tid < F) lbias[tid] = a.bias ? a.bias[tid] : 0.f;
  __syncthreads();

  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  if (lds0 != 0) __builtin_trap();        // the asm stream forms gather addresses from column words: the image must sit at LDS address 0
  const uint32_t lds_col = lds0 + IMG + 2 * WB;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
  // ---- the operand of a sequence and step: every B fragment of the wave, resident for all chunks ----------------------------------
  bf16x8 bfr[STILES][KS];
  // MODE 1: does this item lay out the operand of the workgroup's next item? A scalar integer on purpose: as a lane mask hipcc also parks
  // a per-lane copy of the (loop-invariant) condition in a vector register -- and spills that
  [[maybe_unused]] const int pk_item = __builtin_amdgcn_readfirstlane((MODE == 1 && a.pk_src0 && b + (int)gridDim.x < B) ? 1 : 0);
  [[maybe_unused]] bool xpre = false;        // (GCRNN_SEQ_X_PREFETCH) the x half of the NEXT step's operand has been requested (wave-uniform)
#pragma unroll 1
  for (int step = 0; step < a.nsteps; ++step) {
    stamp_on = (step == (a.nsteps > 1 ? a.nsteps - 2 : 0));      // a typical step: it also lays out the next step's operand
    // this step's arrays (wave-uniform pointer arithmetic; descriptors in SGPRs)
    const uint16_t* hprev = step == 0 ? a.hfirst : a.hrest + (int64_t)(step - 1) * a.hstride;
    const uint16_t* xt = (XS > 0) ? a.x0 + (int64_t)step * a.xstride : nullptr;
    uint16_t* hout = a.out0 ? a.out0 + (int64_t)step * a.ostride : nullptr;
    const uint16_t* aux0 = a.a0 ? a.a0 + (int64_t)step * a.a0stride : nullptr;
    const uint16_t* aux1 = a.a1 ? (a.a1_last_only ? (step == a.nsteps - 1 ? a.a1 : nullptr) : a.a1 + (int64_t)step * a.a1stride) : nullptr;
    const int pka = a.pk_ahead > 0 ? a.pk_ahead : 1;
    // MODE 1 (items, no recurrence): the pack lays out the operand of the NEXT item of this workgroup's loop, item b + gridDim.x =
    // (t', b') = (nb / hmod, nb % hmod) of the user-layout X[b'][t'] -- the caller laid out the first gridDim.x items
    const int nb = b + (int)gridDim.x;
    const uint16_t* pk_src;
    uint16_t* pk_dst;
    if constexpr (MODE == 1) {       // (integer mask instead of a boolean select: see pk_item)
      const uintptr_t m = (uintptr_t)0 - (uintptr_t)pk_item;
      pk_src = reinterpret_cast<const uint16_t*>(reinterpret_cast<uintptr_t>(a.pk_src0) & m);
      pk_dst = reinterpret_cast<uint16_t*>(reinterpret_cast<uintptr_t>(a.pk_dst0) & m);
    } else {
      const bool pk = a.pk_src0 && (a.pk_all || step + pka < a.nsteps);
      pk_src = pk ? a.pk_src0 + (int64_t)step * a.pksrc_stride : nullptr;
      pk_dst = pk ? a.pk_dst0 + (int64_t)step * a.pkdst_stride : nullptr;
    }
    const int pk_stride = a.pk_stride, ubstride = a.ubstride;
    const int64_t pk_soff = (MODE == 1) ? (int64_t)(nb % a.hmod) * pk_stride + (int64_t)(nb / a.hmod) * a.pksrc_stride : (int64_t)b * pk_stride;
    const int pk_db = (MODE == 1) ? nb : b;
    float* gate_out = a.go0 ? a.go0 + (int64_t)step * a.gostride : nullptr;
    // wave-uniform; MODE 4 with an input operand filters [0 | x_t]: the state half is all zeros by contract
    // (kept as a scalar integer: as a lane mask hipcc also parks a per-lane copy of it in a vector register -- and spills that)
    const int skip_hi = __builtin_amdgcn_readfirstlane(((MODE == 4 && XS > 0) || ((MODE == 1) && a.flags && a.flags[0] != 0)) ? 1 : 0);
    const bool skip_h = skip_hi != 0;
    // (skipped state operand: a zero-length descriptor -- its loads return zeros and cost nothing)
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(hprev), 0, skip_h ? 0 : (MODE == 1 ? a.hmod : B) * (NP * F * 2), 0x00020000);
    const int bh = (MODE == 1) ? b % a.hmod : b;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xt), 0, (XS > 0 && xt) ? B * (NP * G * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(hout, 0, hout ? B * (NP * F * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(aux0), 0, ((MODE == 2 || MODE == 5) && aux0) ? B * (NP * F * 2) : 0, 0x00020000);
    uint16_t* yhout = (MODE == 5 && a.yh0) ? a.yh0 + (int64_t)step * a.yhstride : nullptr;
    const __amdgpu_buffer_rsrc_t rsrc_yh = __builtin_amdgcn_make_buffer_rsrc(yhout, 0, yhout ? B * (NP * F * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(aux1), 0, (MODE == 2 && aux1) ? B * (NP * F * 2) : 0, 0x00020000);

    // (k-step major: the seed's first MFMAs need k-step 0 of every tile, which is then the first quarter of the requests to land;
    //  the x half may already be on its way: requested during the previous step's last hop, see below)
    int qo = q;                                        // (opaque per step, as above)
    asm volatile("" : "+v"(qo));
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#ifdef GCRNN_SEQ_X_PREFETCH
      if (s >= HS && xpre) continue;
#endif
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        int w = woff[i];
        asm volatile("" : "+v"(w));
        if (s < HS)
          bfr[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, (w >> 16) * (F * 2) + 16 * qo + 64 * s, bh * (NP * F * 2), 0));
        else
          bfr[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (w >> 16) * (G * 2) + 16 * qo + 64 * (s - HS), b * (NP * G * 2), 0));
      }
    }
    xpre = false;
    float gsc = 1.f;
    if (MODE == 2 && a.gf0) gsc = a.gf0[(int64_t)step * a.gfstride + b];
    float gin = 1.f, gfo = 1.f, gratio = 1.f;
    [[maybe_unused]] float epn[MODE == 5 ? STILES : 1][2];      // MODE 5: this lane's node gates (x scalar time gates) per tile, for all chunks of the step
    if constexpr (MODE == 5) {
      if (a.gi0) { gin = a.gi0[(int64_t)step * a.gfstride + b]; gfo = a.gf0[(int64_t)step * a.gfstride + b]; }
      const float* ng = a.ng0 + (int64_t)step * a.ngstride;
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        const int node = (wv >> 16) < N ? (wv >> 16) : N - 1;
        epn[i][0] = gin * ng[(int64_t)b * N + node];
        epn[i][1] = gfo * ng[(int64_t)(B + b) * N + node];
      }
    }
    if constexpr (GATED) {
      gin = a.gi0[(int64_t)step * a.gfstride + b];
      gfo = a.gf0[(int64_t)step * a.gfstride + b];
      gratio = gfo / fmaxf(gin, 1e-30f);
    }
    GCRNN_STAMP(1);

    f32x4 u[STILES];
    // u_tap of chunk c for the wave's 8 tiles: one weight fragment feeds 8 independent MFMA chains
    auto taps_to = [&](int tap, int c, f32x4 (&u)[STILES]) {
      int ln = lane;                                   // (opaque per call: the fragment address is re-derived, not kept -- or spilled -- across the hops)
      asm volatile("" : "+v"(ln));
      const uint32_t wofs = (uint32_t)(IMG + (c & 1) * WB) + (uint32_t)ln * 16u;       // 32-bit LDS offset of this lane's fragment piece
#pragma unroll
      for (int i = 0; i < STILES; ++i) u[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (GATED || MODE == 1 || (MODE == 4 && XS > 0)) {
        // (GATED) gi (x W_x) + gf (h W_h) on ONE accumulator chain: h-chain, scale by gf / gi, continue with x, scale by gi (gi = sigmoid(.) > 0;
        // the wave-uniform guard covers an underflowed gate); (MODE 1) the state half is skipped when h0 is all zeros
        if (!skip_h) {
#pragma unroll
          for (int s = 0; s < HS; ++s) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)((tap * KS + s) * 1024)));
#pragma unroll
            for (int i = 0; i < STILES; ++i) u[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[i][s], u[i], 0, 0, 0);
          }
        }
        const bool xpart = !GATED || gin > 1e-30f;
        if constexpr (GATED) {
#pragma unroll
          for (int i = 0; i < STILES; ++i) u[i] *= (xpart ? gratio : gfo);
        }
        if (xpart) {
#pragma unroll
          for (int s = HS; s < KS; ++s) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)((tap * KS + s) * 1024)));
#pragma unroll
            for (int i = 0; i < STILES; ++i) u[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[i][s], u[i], 0, 0, 0);
          }
          if constexpr (GATED) {
#pragma unroll
            for (int i = 0; i < STILES; ++i) u[i] *= gin;
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)((tap * KS + s) * 1024)));
#pragma unroll
          for (int i = 0; i < STILES; ++i) u[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[i][s], u[i], 0, 0, 0);
        }
      }
    };
    auto taps = [&](int tap, int c) { taps_to(tap, c, u); };
    // seed of chunk c: tap K-1 goes straight into the hop image (every wave has left the image: the barrier before)
    // hop j (1 .. K-1) reads image B when K - 1 - j is odd: the LAST hop always reads image A, because the inline-pack tile (= image B)
    // is filled by LDS-DMA during it; the seed goes into the image hop 1 reads
    constexpr bool SEED_B = ((K - 2) & 1) != 0;
    auto seed = [&](int c) {
      taps(K - 1, c);
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        state_put<true>(reinterpret_cast<float*>(smem + (SEED_B ? IMGB : 0)), wv, u[i]);
      }
    };
    seed(0);

#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk) {
      // opaque per chunk: the index arithmetic of the prefetches, the LDS-DMA pieces and the epilogue's row stores is re-derived
      // from it inside the loop -- hoisted out of the loops it would have to be spilled (the operand owns 128 registers)
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int r = tl & 15, q = (tl >> 4) & 3, lane = tl & 63;      // (shadow the kernel-scope lane coordinates: per-chunk copies are cheaper than their spills)
      lds_barrier();      // the seed is in the image
      GCRNN_STAMP(2 + chunk * 14);

      // cold lines: the user-layout rows the LDS-DMA of the last hop will fetch; MODE 2: the epilogue's operands
      uint32_t prefetched_pk = 0;
      {
        constexpr int NPC = NP / NCH;
        const int prow = tl >> 3, pj = tl & 7;
#ifdef GCRNN_SEQ_PK_PREFETCH      // A/B (tools/seq_stamps.py, GCRNN_STAMP_FLAGS): touching the cold user-layout rows from here makes hop 1 take 64 instead of 42 k-cycles/100 and the launch 4.4 % longer (profiles/r03_seq_stamps_*.txt) -- off
        if (pk_src && prow < PKROWS && pj < 5 && chunk * NPC < N) {
          int node = chunk * NPC + (pj < 4 ? pj * 64 : NPC - 2);
          node = node < N - 2 ? node : N - 2;
          prefetched_pk = *reinterpret_cast<const uint32_t*>(pk_src + pk_soff + (int64_t)prow * N + node);
        }
#endif
      }
      uint32_t prefetched_epi = 0, prefetched_epi2 = 0;
      if constexpr (MODE == 2) {
        constexpr int ELINES = NP * F * 2 / 128 / NCH;
        const int idx = tl < ELINES ? tl : tl - ELINES;
        // two branches, each with a wave-uniform descriptor (a per-lane choice of the descriptor becomes a waterfall loop) and its OWN
        // destination register (two in-flight loads into one register make hipcc wait for the first: a cold HBM latency per chunk)
#ifdef GCRNN_SEQ_EPI_PREFETCH      // A/B: with the operand loads spread over the hops (below) a touch of their lines from here only costs (taps of hop 1: 35 instead of 5 k-cycles/100, launch + 4.2 %) -- off
        if (tl < ELINES) prefetched_epi = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a1, (chunk * ELINES + idx) * 128, b * (NP * F * 2), 0);
        else if (tl < 2 * ELINES) prefetched_epi2 = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a0, (chunk * ELINES + idx) * 128, b * (NP * F * 2), 0);
#endif
      }

      u32x2 eph[MODE == 2 ? STILES : 1], epg[(MODE == 2 || MODE == 5) ? STILES : 1];
      // ---- Horner hops on the bf16 image; the tap a hop adds is evaluated from the resident operand right before it -------------
      auto hop = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;      // (a compile-time hop index: which image it reads is part of the stream's text)
        GCRNN_STAMP(2 + chunk * 14 + 2 * j - 1);
        if constexpr (MODE == 5) {
          // Yx_t = A(S)x_t + b of the all-items pass, this lane's (node, 4 features) per tile: spread over the hops like the chain's operands
          constexpr int HL = (K > 2) ? K - 2 : 1;
          constexpr int PER = (STILES + HL - 1) / HL;
          if (j <= HL) {
#pragma unroll
            for (int i = (j - 1) * PER; i < j * PER && i < STILES; ++i) {
              int wv = woff[i];
              asm volatile("" : "+v"(wv));
              epg[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, (wv >> 16) * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
            }
          }
        }
        if constexpr (MODE == 2) {
          // the epilogue's operands h_{t-1}, dH_{t-1} of this lane's (node, 4 features): 8-byte gathers of 32-byte row pieces, whose
          // issue alone costs ~4.5 k cycles per chunk when all 16 are requested at once (measured at the last hop). Spread over the
          // hops -- STILES / (K - 1) tiles before each -- the memory pipeline absorbs them while the wave streams LDS.
          constexpr int HL = (K > 2) ? K - 2 : 1;                     // hops that carry loads: all but the last (it issues the LDS-DMA pieces)
          constexpr int PER = (STILES + HL - 1) / HL;
          if (j <= HL) {
#pragma unroll
            for (int i = (j - 1) * PER; i < j * PER && i < STILES; ++i) {
              int wv = woff[i];
              asm volatile("" : "+v"(wv));
              const int eoff = (wv >> 16) * (F * 2) + (chunk * FC + q * 4) * 2;
              eph[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_a1, eoff, b * (NP * F * 2), 0);
              epg[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, eoff, b * (NP * F * 2), 0);
            }
          }
        }
        if (j == (K > 2 ? K - 2 : K - 1)) {
          // the next chunk's weight fragments (after the last chunk: those of chunk 0, for the next step): LDS-DMA into the other buffer
          // (free since this chunk's seed: the previous chunk was its last reader), one hop before the inline-pack pieces
          {
            const int nc = (chunk + 1) % NCH;
            const char* wsrc = reinterpret_cast<const char*>(a.wpack) + (size_t)nc * WB;
            char* wdst = smem + IMG + (nc & 1) * WB;
#pragma unroll
            for (int i = 0; i < (WB / 1024 + SWAVES - 1) / SWAVES; ++i) {
              const int piece = i * SWAVES + wave;
              if (piece < WB / 1024)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(wdst + piece * 1024), 16, 0, 0);
            }
          }
        }
#ifdef GCRNN_SEQ_X_PREFETCH      // compile-time experiment (tools/ab_build.sh "-DGCRNN_SEQ_X_PREFETCH" with GCRNN_SEQ_PK_AHEAD=2)
        if constexpr (MODE == 0 && XS > 0) {
          // Cross-step operand prefetch (EXPERIMENT, off: measured slower, see the launcher): after the last tap of the last chunk the x half of the operand registers is dead, and x_{t+1}
          // does not depend on this step -- it is complete in its sequence-major array (packed by the caller, or laid out by the inline
          // pack TWO steps ahead) -- so it is requested now and lands while the last hop and the epilogue run; the step boundary then
          // waits for h_t alone. (The h half IS this step's output.)
          if (a.xprefetch && j == K - 1 && chunk == NCH - 1 && step + 1 < a.nsteps && (!a.pk_src0 || pka >= 2)) {
            const __amdgpu_buffer_rsrc_t rsrc_xn = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.x0 + (int64_t)(step + 1) * a.xstride), 0, B * (NP * G * 2), 0x00020000);
#pragma unroll
            for (int s2 = HS; s2 < KS; ++s2)
#pragma unroll
              for (int i = 0; i < STILES; ++i) {
                int w = woff[i];
                asm volatile("" : "+v"(w));
                bfr[i][s2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_xn, (w >> 16) * (G * 2) + 16 * q + 64 * (s2 - HS), b * (NP * G * 2), 0));
              }
            xpre = true;
          }
        }
#endif
        if (j == K - 1) {
          if (pk_src) {
            constexpr int NPC = NP / NCH, PPR = NPC / 8, PIECES = PKROWS * PPR;
            static_assert(PIECES % STHREADS == 0, "whole pieces per thread");
            const uint16_t* xsrc = pk_src + pk_soff + chunk * NPC;
#pragma unroll
            for (int i = 0; i < PIECES / STHREADS; ++i) {
              const int id = i * STHREADS + tl;
              const int row = id / PPR, cs = id - row * PPR;
              const int col = (cs - (row >> 3)) & (PPR - 1);
              if (chunk * NPC + col * 8 < N)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + (int64_t)row * N + col * 8),
                                                 (__attribute__((address_space(3))) void*)(xtile + (i * STHREADS + wave * 64) * 16), 16, 0, 0);
            }
          }
        }
        // the stream only sums the gathered rows of every tile (its exits then cost nothing); the tap this hop adds is evaluated from the
        // resident operand after it, and acc = tap + w * sum is the same fused multiply-add the accumulating stream applies per tile
        constexpr bool READ_B = ((K - 1 - j) & 1) != 0;          // (j is a compile-time constant: the loop is unrolled)
        {
          f32x4 dsum[STILES];
          GCRNN_HOP_ASM_UNI16_SUMS_STREAM_IMG(dsum, READ_B);
          taps(K - 1 - j, chunk);
          const f32x2 w2 = f32x2{uni_w, uni_w};
#pragma unroll
          for (int i = 0; i < STILES; ++i) {       // (two packed FMAs per tile, as the accumulating stream's exits)
            const f32x2 lo = __builtin_elementwise_fma(w2, f32x2{dsum[i][0], dsum[i][1]}, f32x2{u[i][0], u[i][1]});
            const f32x2 hi = __builtin_elementwise_fma(w2, f32x2{dsum[i][2], dsum[i][3]}, f32x2{u[i][2], u[i][3]});
            u[i] = f32x4{lo[0], lo[1], hi[0], hi[1]};
          }
        }
        GCRNN_STAMP(2 + chunk * 14 + 2 * j);
        if (j < K - 1) {
          // write-back into the OTHER image (nobody reads it during this hop: its last readers passed the previous barrier), then the
          // hop's one barrier: every wave's rows are in, every wave has left the image just read
#pragma unroll
          for (int i = 0; i < STILES; ++i) {
            int wv = woff[i];
            asm volatile("" : "+v"(wv));
            state_put<true>(reinterpret_cast<float*>(smem + (READ_B ? 0 : IMGB)), wv, u[i]);
          }
          lds_barrier();
        }
      };
      seq_static_for(hop, std::make_integer_sequence<int, K - 1>{});
      // the LDS-DMA pieces (next weights, inline-pack tile) have had the last hop to land; wait before the epilogue's barriers
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      GCRNN_STAMP(2 + chunk * 14 + 9);

      if constexpr (MODE == 2) {
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int node = wv >> 16;
          const int eoff = node * (F * 2) + (chunk * FC + q * 4) * 2;
          const f32x4 raw = u[i];
          f32x4 o = raw * gsc;
          float hv0 = 0.f, hv1 = 0.f, hv2 = 0.f, hv3 = 0.f;
          if (aux1) {
            const u32x2 h2 = (K > 1) ? eph[i] : __builtin_amdgcn_raw_buffer_load_b64(rsrc_a1, eoff, b * (NP * F * 2), 0);
            hv0 = bf2f((uint16_t)(h2[0] & 0xffffu)); hv1 = bf2f((uint16_t)(h2[0] >> 16));
            hv2 = bf2f((uint16_t)(h2[1] & 0xffffu)); hv3 = bf2f((uint16_t)(h2[1] >> 16));
          }
          if (gate_out) part = __builtin_fmaf(raw[3], hv3, __builtin_fmaf(raw[2], hv2, __builtin_fmaf(raw[1], hv1, __builtin_fmaf(raw[0], hv0, part))));      // (explicit chain: with -ffp-contract=fast the association of a*b + c*d + .. is the compiler's choice, per instantiation)
          if (aux0) {
            const u32x2 g2 = (K > 1) ? epg[i] : __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, eoff, b * (NP * F * 2), 0);
            const float g0 = bf2f((uint16_t)(g2[0] & 0xffffu)), g1 = bf2f((uint16_t)(g2[0] >> 16));
            const float g2f = bf2f((uint16_t)(g2[1] & 0xffffu)), g3 = bf2f((uint16_t)(g2[1] >> 16));
            o[0] = (o[0] + g0) * (1.f - hv0 * hv0);
            o[1] = (o[1] + g1) * (1.f - hv1 * hv1);
            o[2] = (o[2] + g2f) * (1.f - hv2 * hv2);
            o[3] = (o[3] + g3) * (1.f - hv3 * hv3);
          }
          uint2 pkd;
          if (node < N) {
            pkd.x = pack2bf(o[0], o[1]);
            pkd.y = pack2bf(o[2], o[3]);
          } else {
            pkd.x = 0u; pkd.y = 0u;
          }
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pkd.x, pkd.y}, rsrc_o, eoff, b * (NP * F * 2), 0);
        }
        if (gate_out) {
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
          if (lane == 0) gate_out[(int64_t)b * (NCH * SWAVES) + chunk * SWAVES + wave] = part;
        }
      } else if constexpr (MODE == 4) {
        // filter-output pass (graphML.py:2420, one filter of an item): A(S) operand + b, bf16, sequence-major, no non-linearity
        const int ql = (tl & 63) >> 4;      // (from the per-chunk opaque id: hoisted out of the loops these addresses are spilled)
        float bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = lbias[chunk * FC + ql * 4 + c];
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int node = wv >> 16;
          uint2 pkd{0u, 0u};
          if (node < N) {
            const f32x4 acc = u[i];
            pkd.x = pack2bf(acc[0] + bv[0], acc[1] + bv[1]);
            pkd.y = pack2bf(acc[2] + bv[2], acc[3] + bv[3]);
          }
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pkd.x, pkd.y}, rsrc_o, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
        }
      } else if constexpr (MODE == 1) {
        // gate pre-pass: partial dot product of tanh(pre) with the gate's read-out weights over this chunk, one partial per wave (the
        // caller adds them in a fixed order); with an output array the gate cell's state c = tanh(pre) is also stored (bf16)
        const int ql = (tl & 63) >> 4;      // (from the per-chunk opaque id: hoisted out of the loops these addresses are spilled)
        float bs2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bs2[c] = 2.f * lbias[chunk * FC + ql * 4 + c];
        if (a.tapf) {
          // Node gates (inference): the F -> 1 GraphFilter of the gate cell's state starts with its per-tap dot products s_k[n] = <c[n, :], w_k>
          // (graphML.py:2387, taps first). They are taken here, from the bf16-rounded c this lane has just packed (the values the stored
          // states would hold), on the matrix cores: D[tap][slot] = sum_f W[tap][f] c[f][slot] with the taps as A rows -- a lane's packed
          // pair of dwords IS its B operand of v_mfma_f32_16x16x16_bf16 (k = 4 q + e = the lane's four features), the fp32 taps enter as
          // three bf16 planes (their sum is the fp32 value to 24 bits, every product with a bf16 c is exact in fp32). Lane (r, q') then
          // holds taps 4 q' .. 4 q' + 3 of ITS OWN node and adds them to the item's accumulators in LDS; the item's last chunk is followed
          // by a coalesced copy-out. Neither the state array nor a second pass over it is needed.
          typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
          const char* tf = smem + a.tapf_off + chunk * (3 * 512) + (tl & 63) * 8;
          const s16x4 wa0 = *reinterpret_cast<const s16x4*>(tf), wa1 = *reinterpret_cast<const s16x4*>(tf + 512), wa2 = *reinterpret_cast<const s16x4*>(tf + 1024);
          float* sacc = reinterpret_cast<float*>(smem + a.sacc_off);
#pragma unroll
          for (int i = 0; i < STILES; ++i) {
            int wv = woff[i];
            asm volatile("" : "+v"(wv));
            const int node = wv >> 16;
            uint2 pkd{0u, 0u};
            if (node < N) {
              const f32x4 acc = u[i];
              const float o0 = fast_tanh(acc[0] + bs2[0]), o1 = fast_tanh(acc[1] + bs2[1]);
              const float o2 = fast_tanh(acc[2] + bs2[2]), o3 = fast_tanh(acc[3] + bs2[3]);
              pkd.x = pack2bf(o0, o1);
              pkd.y = pack2bf(o2, o3);
            }
            if (hout) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pkd.x, pkd.y}, rsrc_o, node * (F * 2) + (chunk * FC + ql * 4) * 2, b * (NP * F * 2), 0);
            const s16x4 cb = __builtin_bit_cast(s16x4, pkd);
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa0, cb, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa1, cb, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa2, cb, d, 0, 0, 0);
            u[i] = d;                                  // (the tile's accumulator is dead: it keeps the tap dots until the adds below)
          }
          // add to the item's accumulators: plain read-modify-write -- a (tap, node) pair belongs to ONE lane of one wave (LDS float
          // atomics measured 12 % of the pre-pass); two tiles per round so that eight reads are in flight
#pragma unroll
          for (int i0 = 0; i0 < STILES; i0 += 2) {
            float old[2][4];
            int nd[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
              int wv = woff[i0 + t2];
              asm volatile("" : "+v"(wv));
              nd[t2] = wv >> 16;
#pragma unroll
              for (int c = 0; c < 4; ++c) old[t2][c] = (nd[t2] < N && ql * 4 + c < a.ntaps) ? sacc[(ql * 4 + c) * NP + nd[t2]] : 0.f;
            }
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (nd[t2] < N && ql * 4 + c < a.ntaps) sacc[(ql * 4 + c) * NP + nd[t2]] = old[t2][c] + u[i0 + t2][c];
          }
        } else {
        float part = 0.f;
        // the read-out weights [N][F] fp32 (shared by every item: L2-resident; held across an asm block they would spill): requested
        // two tiles at a time (more in flight spills in this instantiation), right after the last hop's stream -- its register window is free here. Inside the per-tile `node < N`
        // regions they were eight dependent L2 round trips per chunk.
        constexpr int WBATCH = 2;
        float4 w8[WBATCH];
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          if (i % WBATCH == 0) {
#pragma unroll
            for (int i2 = 0; i2 < WBATCH; ++i2) {
              int wv2 = woff[i + i2];
              asm volatile("" : "+v"(wv2));
              const int node2 = (wv2 >> 16) < N ? (wv2 >> 16) : N - 1;
              w8[i2] = *reinterpret_cast<const float4*>(a.gw + (int64_t)node2 * F + chunk * FC + ql * 4);
            }
          }
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int node = wv >> 16;
          uint2 pkd{0u, 0u};
          if (node < N) {
            const float4 w4 = w8[i % WBATCH];
            const f32x4 acc = u[i];
            const float o0 = fast_tanh(acc[0] + bs2[0]), o1 = fast_tanh(acc[1] + bs2[1]);
            const float o2 = fast_tanh(acc[2] + bs2[2]), o3 = fast_tanh(acc[3] + bs2[3]);
            part = __builtin_fmaf(o3, w4.w, __builtin_fmaf(o2, w4.z, __builtin_fmaf(o1, w4.y, __builtin_fmaf(o0, w4.x, part))));      // (explicit chain, as the chain's partials)
            pkd.x = pack2bf(o0, o1);
            pkd.y = pack2bf(o2, o3);
          }
          if (hout) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pkd.x, pkd.y}, rsrc_o, node * (F * 2) + (chunk * FC + ql * 4) * 2, b * (NP * F * 2), 0);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if (lane == 0) gate_out[(int64_t)b * (NCH * SWAVES) + chunk * SWAVES + wave] = part;
        }
      } else {
        float bsum[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bsum[c] = (MODE == 5 ? 1.f : (gin + gfo)) * lbias[chunk * FC + q * 4 + c];      // the one bias is added by both filters (MODE 5: Yx carries its own)
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int node = wv >> 16;
          const f32x4 acc = u[i];
          uint2 pkd;
          if constexpr (MODE == 5) {
            // node-gated step (graphML.py:2420-2423): h_t = tanh(ni (A(S)x_t + b) + nf (B(S)h_{t-1} + b)), gates per node (and sequence)
            const int eoff = node * (F * 2) + (chunk * FC + q * 4) * 2;
            if (node < N) {
              const u32x2 y2 = (K > 1) ? epg[i] : __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, eoff, b * (NP * F * 2), 0);
              const float ni = epn[i][0], nf = epn[i][1];
              const float yh0 = acc[0] + bsum[0], yh1 = acc[1] + bsum[1], yh2 = acc[2] + bsum[2], yh3 = acc[3] + bsum[3];
              if (yhout) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2bf(yh0, yh1), pack2bf(yh2, yh3)},
                                                               rsrc_yh, eoff, b * (NP * F * 2), 0);
              const float o0 = fast_tanh(ni * bf2f((uint16_t)(y2[0] & 0xffffu)) + nf * yh0);
              const float o1 = fast_tanh(ni * bf2f((uint16_t)(y2[0] >> 16)) + nf * yh1);
              const float o2 = fast_tanh(ni * bf2f((uint16_t)(y2[1] & 0xffffu)) + nf * yh2);
              const float o3 = fast_tanh(ni * bf2f((uint16_t)(y2[1] >> 16)) + nf * yh3);
              pkd.x = pack2bf(o0, o1);
              pkd.y = pack2bf(o2, o3);
            } else {
              pkd.x = 0u; pkd.y = 0u;
              if (yhout) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, rsrc_yh, eoff, b * (NP * F * 2), 0);
            }
          } else if (node < N) {
            const float o0 = fast_tanh(acc[0] + bsum[0]), o1 = fast_tanh(acc[1] + bsum[1]);
            const float o2 = fast_tanh(acc[2] + bsum[2]), o3 = fast_tanh(acc[3] + bsum[3]);
            pkd.x = pack2bf(o0, o1);
            pkd.y = pack2bf(o2, o3);
          } else {
            pkd.x = 0u; pkd.y = 0u;
          }
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pkd.x, pkd.y}, rsrc_o, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
          u[i] = f32x4{__uint_as_float(pkd.x), __uint_as_float(pkd.y), 0.f, 0.f};
        }
        GCRNN_STAMP(2 + chunk * 14 + 10);
        if (aux1) {
          // user layout H[b][t][f][:] (node-contiguous rows). The tile nodes are degree-ranked, i.e. scattered, so the chunk goes through a
          // transposed LDS tile -- of 32-bit words [feature pair][node] = {h[n][2p], h[n][2p+1]}: a lane's two packed registers ARE two such
          // words, 16 ds_write_b32 per lane instead of the chunk-parallel kernel's 32 two-byte scatters (measured there: 14 LDS cycles per
          // ds_write_b16, 11 % of a chunk). The row stores split the pairs again (v_perm_b32): 8 nodes x 2 features per thread and trip.
          constexpr int RS2 = 4 * NP + 32;               // 8 words more per row: the rows of a lane pair (q, q + 1) sit 16 banks apart
          char* tst = reinterpret_cast<char*>(state);
          lds_barrier();
#pragma unroll
          for (int i = 0; i < STILES; ++i) {
            int wv = woff[i];
            asm volatile("" : "+v"(wv));
            const int node = wv >> 16;
            char* ra = tst + (2 * q) * RS2 + node * 4;
            *reinterpret_cast<uint32_t*>(ra) = __float_as_uint(u[i][0]);
            *reinterpret_cast<uint32_t*>(ra + RS2) = __float_as_uint(u[i][1]);
          }
          lds_barrier();
          GCRNN_STAMP(2 + chunk * 14 + 11);
          const int segs = N >> 3;
          uint16_t* ub = const_cast<uint16_t*>(aux1) + (int64_t)b * ubstride + (int64_t)(chunk * FC) * N;
          const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(ub, 0, FC * N * 2, 0x00020000);
          for (int idx = tl; idx < (FC / 2) * segs; idx += STHREADS) {
            const int fp = idx / segs, sg = idx - fp * segs;
            typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;
            const u32x4_t w0 = *reinterpret_cast<const u32x4_t*>(tst + fp * RS2 + sg * 32);
            const u32x4_t w1 = *reinterpret_cast<const u32x4_t*>(tst + fp * RS2 + sg * 32 + 16);
            const u32x4_t ev = {__builtin_amdgcn_perm(w0[1], w0[0], 0x05040100u), __builtin_amdgcn_perm(w0[3], w0[2], 0x05040100u),
                                __builtin_amdgcn_perm(w1[1], w1[0], 0x05040100u), __builtin_amdgcn_perm(w1[3], w1[2], 0x05040100u)};
            const u32x4_t od = {__builtin_amdgcn_perm(w0[1], w0[0], 0x07060302u), __builtin_amdgcn_perm(w0[3], w0[2], 0x07060302u),
                                __builtin_amdgcn_perm(w1[1], w1[0], 0x07060302u), __builtin_amdgcn_perm(w1[3], w1[2], 0x07060302u)};
            __builtin_amdgcn_raw_buffer_store_b128(ev, rsrc_u, ((2 * fp) * N + sg * 8) * 2, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(od, rsrc_u, ((2 * fp + 1) * N + sg * 8) * 2, 0, 0);
          }
        }
      }
      GCRNN_STAMP(2 + chunk * 14 + 12);
      // second half of the inline pack (as the chunk-parallel kernel): [rows][NPC] tile -> whole sequence-major rows. gfx950 store-data
      // hazard: every piece sits in its own register tuple, which stays reserved until the stores have retired -- here
      // that wait comes AFTER the next chunk's seed, whose MFMAs cover the drain of this chunk's stores.
      constexpr int NPCp = NP / NCH, PCS = PKROWS / 8, RI = PCS * NPCp / STHREADS;
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;
      u32x4_t vv[RI];
#pragma unroll
      for (int i = 0; i < RI; ++i) vv[i] = u32x4_t{0u, 0u, 0u, 0u};
      if (pk_src) {
        if (!(MODE == 0 && aux1)) lds_barrier();
        const __amdgpu_buffer_rsrc_t rsrc_xn = __builtin_amdgcn_make_buffer_rsrc(pk_dst, 0, B * (NP * PKROWS * 2), 0x00020000);
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          const int id = i * STHREADS + tl;
          const int nl = id / PCS, pc = id - nl * PCS;
          const char* src = xtile + (pc * 8) * (NPCp * 2) + ((nl + 8 * pc) & (NPCp - 1)) * 2;
          uint32_t w4[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const uint32_t lo = *reinterpret_cast<const uint16_t*>(src + (2 * jj) * (NPCp * 2));
            const uint32_t hi = *reinterpret_cast<const uint16_t*>(src + (2 * jj + 1) * (NPCp * 2));
            w4[jj] = lo | (hi << 16);
          }
          const bool ok = chunk * NPCp + nl < N;
          vv[i] = u32x4_t{ok ? w4[0] : 0u, ok ? w4[1] : 0u, ok ? w4[2] : 0u, ok ? w4[3] : 0u};
        }
#pragma unroll
        for (int i = 0; i < RI; ++i) asm volatile("" : "+v"(vv[i]));          // every piece in its own tuple before the first store
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          const int id = i * STHREADS + tl;
          const int nl = id / PCS, pc = id - nl * PCS;
          __builtin_amdgcn_raw_buffer_store_b128(vv[i], rsrc_xn, (chunk * NPCp + nl) * (PKROWS * 2) + pc * 16 + pk_db * (NP * PKROWS * 2), 0, 0);
        }
      }
      lds_barrier();     // the last hop's reads of the image (and the epilogue's of its tiles) are done: the image may be seeded again
      GCRNN_STAMP(2 + chunk * 14 + 13);
      if (chunk + 1 < NCH) seed(chunk + 1);
      // every store of this chunk has retired before its registers are reused (and, after the last chunk, before this workgroup
      // reads its own h_t / laid-out rows back as the next step's operand)
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(vv[0]), "v"(vv[RI - 1]) : "memory");
#pragma unroll
      for (int i = 1; i + 1 < RI; ++i) asm volatile("" ::"v"(vv[i]));
      asm volatile("" ::"v"(prefetched_epi), "v"(prefetched_epi2), "v"(prefetched_pk));
    }  // chunks
#pragma unroll
    for (int i = 0; i < STILES; ++i)
#pragma unroll
      for (int s = 0; s < KS; ++s) asm volatile("" ::"v"(bfr[i][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();       // every wave's stores of this step have retired (the waits above): the next step may read them
    if constexpr (MODE == 1) {
      if (a.tapf) {
        // the item's tap dots: accumulators (every wave's adds are in: the barrier above) -> [ntaps][N] fp32, coalesced; zeroed for the
        // workgroup's next item, whose first add comes many barriers later
        float* sacc = reinterpret_cast<float*>(smem + a.sacc_off);
        float* so = a.taps_out + (int64_t)b * a.ntaps * N;
        for (int tap = 0; tap < a.ntaps; ++tap)
          for (int n = tid; n < N; n += STHREADS) {
            so[tap * N + n] = sacc[tap * NP + n];
            sacc[tap * NP + n] = 0.f;
          }
      }
    }
  }  // steps
  }  // sequences
  GCRNN_STAMP_FLUSH();
}

// LDS bytes of the sequence-resident kernel, or 0 when the problem does not fit
// (ONE formula for the dispatch and for every query of gcrnn_fused.hip -- ADVICE r3: three hand copies had to stay equal to the launch's)
constexpr int GCRNN_SEQ_MAX_GRID = 256;      // workgroups of a launch of the sequence-resident kernel: one per CU; all-items passes walk their items with this stride
static inline size_t fused_seq_lds_bytes(int64_t K, int64_t ks, int64_t entries, size_t extra = 0) {
  const size_t need = (size_t)GCRNN_HOP_IMAGE_B_OFFSET + 32 * 1024 + 2 * (size_t)K * (size_t)ks * 1024 + (size_t)entries * 32 + GCRNN_HOP_COLUMN_PAD + extra;
  return need <= 160 * 1024 ? need : 0;
}
template <int K, int HS, int XS>
static size_t fused_seq_lds(int64_t entries, bool /*inline_pack: its tile is the second hop image*/, int /*pkrows*/, size_t extra = 0) {
  return fused_seq_lds_bytes(K, HS + XS, entries, extra);
}

// Which launches take the sequence-resident kernel: it has ONE workgroup per sequence, so it wins where whole rounds of 256
// sequences fill the chip. Cost model (measured at B = 256, K = 5: a sequence costs the sequence-resident kernel 0.875 x the time of
// its F/16 chunk items in the chunk-parallel kernel, which deals 256 chunk items per round): rounds_seq x 0.875 NCH < rounds_chunk.
// GCRNN_SEQ_MIN_B=n overrides the model (tests), GCRNN_SEQ_KERNEL=0 keeps the chunk-parallel kernel (A/B).
// GCRNN_SEQ_PERSIST=0: one launch per time step instead of one per forward / chain (A/B)
static inline bool fused_seq_persistent() {
  const char* e = getenv("GCRNN_SEQ_PERSIST");
  return !(e && e[0] == '0');
}
static inline bool fused_seq_wanted(int64_t B, int nch) {
  const char* off = getenv("GCRNN_SEQ_KERNEL");      // read per call (once per forward / chain, not per step): tests switch it in-process
  if (off && off[0] == '0') return false;
  const char* mb = getenv("GCRNN_SEQ_MIN_B");
  if (mb) return B >= (atoi(mb) < 1 ? 1 : atoi(mb));
  const double rounds_seq = (double)((B + 255) / 256), rounds_chunk = (double)((B * nch + 255) / 256);
  return rounds_seq * 0.875 * nch < rounds_chunk;
}
