// Sequence-resident fused GCRNN recurrence, 32-feature chunks (round 4): the flagship forward of uniform-weight graphs.
//
// gcrnn_fused_seq.h keeps a sequence's operand [h_{t-1} | x_t] in registers and walks its F/16 output chunks; its phase stamps said
// (DESIGN 4.0): a hop stream costs ~1,150 fixed cycles + 73 per trip for ONE wave whatever the other waves do, the tap MFMAs and
// the acc = tap + w * sum FMAs follow strictly behind it, every chunk ends in a barrier-separated epilogue, and every step waits
// ~100 units for the operand it has just stored to come back from L2. This kernel changes what a hop processes:
//  * 32 output features per chunk as TWO 16-feature image planes (plane 1 = plane 0 + GCRNN_HOP_WIDE_PLANE bytes) read by ONE stream:
//    one column read and two address XORs per trip serve four 16-byte gathers and two v_smfmac (GCRNN_HOP_ASM_WIDE32_TEXT,
//    tools/gen_hop_asm.py gen_wide32) -- half the streams, barriers, seeds and epilogues per step, the per-trip overhead amortised
//    over twice the bytes. Plan arrays are those of the 16-feature bf16 image (graph.fused_plan_img16) unchanged.
//  * the graph weight w is folded into the tap weights (W_k <- w^k W_k, gcrnn_fused_pack_weights_wide): Horner then reads
//    t_j = u~_j + P0 t_{j+1} on the 0/1 pattern P0, so a hop's sums need no multiply and the tap MFMAs accumulate straight ONTO them
//    (D = A B + D): no separate tap tuple, no packed FMAs -- the registers that frees hold the second half's accumulators.
//  * output features are assigned to MFMA rows so that lane (r, q) ends a chunk holding features 32 c + 8 q .. + 7 of its node
//    (half h, row 4 q + e <-> feature 32 c + 8 q + 4 h + e; a permutation of the weight rows only): ONE 16-byte state store per lane
//    and tile -- and those four packed registers ARE the lane's B fragment of k-step c for the next time step. The last chunk's state
//    is handed over in registers; the earlier chunks' (stored long before) and x_{t+1} are requested at the start of the last
//    chunk's epilogue, so the step boundary waits for nothing that was not already on its way.
//  * LDS: planes / transposed user-layout tile 66 KB | ONE chunk's weight fragments K*KS*2 KB (the next chunk's arrive by LDS-DMA
//    during the epilogue) | column words | a 128-node inline-pack tile (8 rounds per step, one per hop at K = 5).
// Arithmetic differs from the 16-feature kernels only in rounding (w^k W_k rounded to bf16 instead of W_k, sums before taps), so this
// kernel is pinned to the fp64 oracle directly (tests/test_fused.py), not bit-compared with them.
//
// MODE 0: forward step h_t = tanh(sum_k S^k([h|x] W_k) + 2b)   (reference Utils/graphML.py:2420-2423), un-gated.
#pragma once

// In-kernel phase stamps (diagnostic builds only, -DGCRNN_SEQ_STAMPS; tools/seq32_stamps.py): lane 0 of wave 0 records s_memtime at its
// phase boundaries of ONE step into LDS (FLAG_OFF, no scalar registers held), copied out at the end.
#if defined(GCRNN_SEQ_STAMPS)
static __device__ unsigned long long gcrnn_seq32_stamps[256 * 96];
// (explicit LDS instructions: a generic-pointer volatile access of the dynamic LDS array makes hipcc emit an illegal compare in some instantiations)
static __device__ __forceinline__ void gcrnn_stamp32_put(uint32_t lds_addr) {
  const unsigned long long tv = __builtin_amdgcn_s_memtime();
  asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr), "v"(tv) : "memory");
}
static __device__ __forceinline__ unsigned long long gcrnn_stamp32_get(uint32_t lds_addr) {
  unsigned long long tv;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tv) : "v"(lds_addr) : "memory");
  return tv;
}
#define GCRNN_STAMP32(slot)                                                                                   \
  do {                                                                                                        \
    if (wave == 0 && stamp_on && lane_now() == 0) gcrnn_stamp32_put((uint32_t)reinterpret_cast<uintptr_t>(smem) + M::FLAG_OFF + 8 * (slot)); \
  } while (0)
#define GCRNN_STAMP32_WAVE(slot0)                                                                             \
  do {                                                                                                        \
    if (stamp_on && lane_now() == 0) gcrnn_stamp32_put((uint32_t)reinterpret_cast<uintptr_t>(smem) + M::FLAG_OFF + 8 * ((slot0) + wave)); \
  } while (0)
#define GCRNN_STAMP32_FLUSH()                                                                                 \
  do {                                                                                                        \
    __syncthreads();                                                                                          \
    if (threadIdx.x < 96 && blockIdx.x < 256) gcrnn_seq32_stamps[blockIdx.x * 96 + threadIdx.x] = gcrnn_stamp32_get((uint32_t)reinterpret_cast<uintptr_t>(smem) + M::FLAG_OFF + 8 * threadIdx.x); \
  } while (0)
#ifndef GCRNN_SEQ32_STAMP_READER_NAME
#define GCRNN_SEQ32_STAMP_READER_NAME gcrnn_debug_read_seq32_stamps      // (one reader per translation unit that includes this header: gcrnn_fused_seq32p.hip names its own)
#endif
extern "C" int GCRNN_SEQ32_STAMP_READER_NAME(void* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(gcrnn_seq32_stamps), sizeof(unsigned long long) * 256 * 96) == hipSuccess ? 0 : 1;
}
#else
#define GCRNN_STAMP32_WAVE(slot0) do {} while (0)
#define GCRNN_STAMP32(slot) do {} while (0)
#define GCRNN_STAMP32_FLUSH() do {} while (0)
#endif

#ifndef GCRNN_SEQ32_R1_STREAM_FIRST
#define GCRNN_SEQ32_R1_STREAM_FIRST 0      // 1 (with R1): every wave streams first, as round 4's first rank-1 variant did (A/B)
#endif
#ifndef GCRNN_SEQ32_OPERAND_AT
#define GCRNN_SEQ32_OPERAND_AT 1      // where the last chunk requests the next step's operand: 0 = at the start of its epilogue, 1 = behind its state stores, 2 = behind its row stores (A/B)
#endif
// cache policy experiments (aux bits of the buffer / LDS-DMA instructions: 2 = nt): traffic this launch never reads again should not
// displace from L2 what it reads back one step later (x_{t+1} laid out by the pack, h_t's first chunk)
#ifndef GCRNN_SEQ32_NT_ROWS
#define GCRNN_SEQ32_NT_ROWS 0      // user-layout row stores
#endif
#ifndef GCRNN_SEQ32_NT_DMA
#define GCRNN_SEQ32_NT_DMA 0       // LDS-DMA reads of the user-layout input
#endif
#ifndef GCRNN_SEQ32_NT_XLOAD
#define GCRNN_SEQ32_NT_XLOAD 0     // requests of the next operand (read once)
#endif
#ifndef GCRNN_SEQ32_NT_STATE
#define GCRNN_SEQ32_NT_STATE 0     // state stores of the LAST chunk (handed over in registers: not read back by this launch)
#endif
#ifndef GCRNN_SEQ32_NO_ITEM_PREFETCH
#define GCRNN_SEQ32_NO_ITEM_PREFETCH 0      // 1: the gate pre-pass loads every item's operand at the item's start (A/B)
#endif
#ifndef GCRNN_SEQ32_GATED_TAPS_FIRST
#define GCRNN_SEQ32_GATED_TAPS_FIRST 1      // 1 (default): in the time-gated recurrence EVERY wave evaluates the tap before its stream (exact, no reciprocal).
                                            // 0: waves 0..3 stream first and start the gated chain from S / gf -- built, but hipcc then spills 4-8 operand registers
                                            // in the K = 4, 5 instantiations (the extra scaling pass keeps a reciprocal live across the MFMA chains): not the default
#endif
#ifndef GCRNN_SEQ32_YOUNG_PRIO
#define GCRNN_SEQ32_YOUNG_PRIO 0
#endif
#ifndef GCRNN_SEQ32_END_WAIT
#define GCRNN_SEQ32_END_WAIT 0     // 1: every chunk ends with s_waitcnt vmcnt(0) (as the first versions did; A/B)
#endif
#ifndef GCRNN_SEQ32_SAME_ORDER
#define GCRNN_SEQ32_SAME_ORDER 0      // 1: every wave streams first, then evaluates the tap (A/B: tools/ab_build.sh "-DGCRNN_SEQ32_SAME_ORDER=1")
#endif

struct Seq32Args {
  const uint16_t* x0; int64_t xstride;                 // x of step 0 [B][NP][G] bf16 sequence-major, elements between steps
  const uint16_t* hfirst;                              // h_{-1} = h0 [B][NP][F]
  uint16_t* out0; int64_t ostride;                     // h_t of step 0 [B][NP][F], elements between steps
  const uint4* wpack;                                  // [F/32][K][2][KS][64] x 16 B (gcrnn_fused_pack_weights_wide)
  const float* bias;                                   // [F] or null
  const uint16_t* a1; int64_t a1stride;                // user-layout output H[0][t] (or null)
  int a1_last_only;                                    // only the last step writes the user-layout output (at a1 itself)
  int ubstride;                                        // elements between consecutive sequences of the user-layout output
  const int32_t* tile_nodes; const int32_t* tile_off; const uint2* ell_col4;      // bf16-image plan (graph.fused_plan_img16)
  int entries, B, N;
  const uint16_t* pk_src0; int64_t pksrc_stride;       // inline pack: user-layout block X[0][0] (or null), elements between steps
  uint16_t* pk_dst0; int64_t pkdst_stride;             // ... the sequence-major array of step 0, elements between steps
  int pk_stride;                                       // ... elements between consecutive sequences of the user-layout tensor
  int nsteps;
  // GATED (time-gated recurrence, graphML.py:2357-2374, 2420-2421): scalar gates of step 0 [B] fp32, elements between steps
  const float* gi0; const float* gf0; int64_t gstride;
  // MODE 1 (gate-PAIR pre-pass: both time gates' sub-cells of every (t, b) item in one pass, graphML.py:2362-2374):
  int hmod;                                            // item i reads the state operand of sequence i % hmod (its h0)
  const int32_t* flags;                                // (or null) flags[0] != 0: h0 is all zeros -- the state half is neither loaded nor multiplied
  const float* gw;                                     // read-out weights of the two gates, [2][N][F] fp32 (node-major)
  const uint4* tapf; float* taps_out; int ntaps;       // MODE 1 (or null; then gw is unused): NODE gates (graphML.py:2379-2393) -- the gate cells' F -> 1 filters, taps first
                                                       // (:2387): A fragments [2 gates][F/32][3 planes][64 lanes] x 16 B of their taps (p0 + p1 + p2 = w to 24 bits; lane
                                                       // 16 kg + tap: w_p[tap][32 cg + 8 kg .. + 7]), output [items][2][F/32][ntaps][N] fp32 partial dots per chunk
  float* go;                                           // [items][2 * F/32 * 8] partial dot products per (chunk, wave); chunks 0 .. F/32-1 = input gate
  uint16_t* out1;                                      // (or null, with out0) the forget gate cell's states [items][NP][F]; out0 = the input gate cell's
  // MODE 2 (BPTT data chain, the adjoint of graphML.py:2420-2423): step i walks t = T-1-i; operand dpre_t = hfirst (step 0) / the previous
  // step's output, output dpre_{t-1} = out0 + i ostride; epilogue operands of step i (negative strides walk backwards in time):
  const uint16_t* dh0_; int64_t dhstride;              // upstream gradient dH_{t-1} [B][NP][F] sequence-major
  const uint16_t* hs0; int64_t hsstride;               // state h_{t-1}
  const float* gsc0; int64_t gscstride;                // (or null) forget gates gf_t [B] of the time-gated cell
  float* gpart0; int64_t gpartstride;                  // (or null) [B][F/32 * 8] partials of <h_{t-1}, adjoint chain of dpre_t> (the forget gate's gradient)
  int final_raw;                                       // != 0: the launch's LAST step stores the raw state gradient (d h0: no upstream term, no tanh') into
  uint16_t* final_out; const uint16_t* final_h;        // final_out (or null), with final_h (h0, or null) as the state of its partials
  // MODE 3 (filter-output pass, graphML.py:2402-2403: A(S) x_t + b for every (t, b) item; operand [0 | x_t], the state half never loaded): out0
  // [items][NP][F] bf16 receives acc + b (no activation). MODE 4 (node-gated recurrence, graphML.py:2379-2407, 2420-2423): state-only operand,
  // h_t = tanh(gi ni_t . Yx_t + gf nf_t . (B(S) h_{t-1} + b)) -- Yx_t = dh0_ + step dhstride (MODE 3's output), the per-node gates:
  const float* ng0; int64_t ngstride; int64_t nghalf;  // ng0 + step ngstride: ni [B][N] fp32, nf = + nghalf; optional scalar time gates gi0 / gf0 (+ step gstride)
  const float* r1a; const float* r1b;                  // R1 (rank-1-weighted graph S[m][n] = a[m] b[n], plan of its 0/1 pattern): the factors [NP] fp32, zero for padding rows
  int stagger;                                         // > 0: workgroup i starts ((i / 8) % 8) * stagger shader cycles late (de-synchronises the CUs' memory phases for the whole launch)
};

// this lane's id, re-derived where it is needed (two VALU instructions; volatile: neither hoisted nor kept): anything derived from the
// thread id that lives across a hop's asm block costs a register the operand, the accumulators and the stream's window do not leave
__device__ __forceinline__ int lane_now() {
  int ln;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
  return ln;
}

// (the stream ADDS the gathered rows to D: the caller zeroes D, or has already put the hop's tap there)
#define GCRNN_HOP_ASM_WIDE32_STREAM(D)                                                             \
  do {                                                                                             \
    const int gwbeg = tbeg[0] >> 2, gwend = tend[STILES - 1] >> 2;                                 \
    if (gwbeg < gwend) {                                                                           \
      const int ls_ = lane_now(), r = ls_ & 15, q = ls_ >> 4;                                      \
      const uint32_t colb = lds_col + r * 8 + (q >> 1) * 4;       /* this lane's own column dword of a slot's pair */ \
      const uint32_t qh_ = (uint32_t)(q & 1) << 4;                                                 \
      asm volatile(GCRNN_HOP_ASM_WIDE32_TEXT                                                       \
                   : "+v"(D[0][0]), "+v"(D[0][1]), "+v"(D[1][0]), "+v"(D[1][1]), "+v"(D[2][0]), "+v"(D[2][1]), "+v"(D[3][0]), "+v"(D[3][1]), \
                     "+v"(D[4][0]), "+v"(D[4][1]), "+v"(D[5][0]), "+v"(D[5][1]), "+v"(D[6][0]), "+v"(D[6][1]), "+v"(D[7][0]), "+v"(D[7][1])  \
                   : "s"(GCRNN_SGPR(tend[0] >> 2)), "s"(GCRNN_SGPR(tend[1] >> 2)), "s"(GCRNN_SGPR(tend[2] >> 2)), "s"(GCRNN_SGPR(tend[3] >> 2)), \
                     "s"(GCRNN_SGPR(tend[4] >> 2)), "s"(GCRNN_SGPR(tend[5] >> 2)), "s"(GCRNN_SGPR(tend[6] >> 2)), "s"(GCRNN_SGPR(tend[7] >> 2)), \
                     "s"(GCRNN_SGPR(gwbeg)), "s"(GCRNN_SGPR(gwend - 1)), "v"(colb), "v"(qh_)               \
                   : GCRNN_HOP_ASM_WIDE32_CLOBBERS);                                               \
    }                                                                                              \
  } while (0)

template <int K, int HS, int XS>
struct Seq32Map {
  static constexpr int KS = HS + XS;
  static constexpr int PL = GCRNN_HOP_WIDE_PLANE;          // plane 1 of the hop image
  static constexpr int RS2 = 4 * NP + 16;                  // row of the transposed user-layout tile: [feature pair][node] words; q and q + 1 sit 16 banks apart
  static constexpr int BIAS_OFF = PL + 32 * 1024;          // behind plane 1 (the transposed tile, 16 rows, ends below it)
  static constexpr int FLAG_OFF = BIAS_OFF + 256;
  static constexpr int WOFF = 2 * PL;                      // one chunk's weight fragments
  static constexpr int WB = K * KS * 2048;
  static constexpr int COL_OFF = WOFF + WB;
  static constexpr int NPCK = 128;                         // nodes per inline-pack round
  static constexpr int PFS = 256;                          // prefetch scratch (one dword per lane)
  static_assert(16 * RS2 <= BIAS_OFF && FLAG_OFF + 768 <= WOFF, "LDS map");
  static size_t lds_bytes(int64_t entries, bool inline_pack, bool r1 = false) {
    const size_t need = (size_t)COL_OFF + (size_t)entries * 32 + GCRNN_HOP_COLUMN_PAD + NP * 4 + (r1 ? 2 * NP * 4 : 0) + (inline_pack ? (size_t)(32 * (XS > 0 ? XS : HS)) * NPCK * 2 : 0)
                        + PFS;      // (the last PFS bytes: where gcrnn_fused_seq32p.h's L2 prefetches land -- LDS-DMA needs a destination, nobody reads it)
    return need <= 160 * 1024 ? need : 0;
  }
};

// VAR: bit 0 = the launch lays out the input itself (inline pack), bit 1 = it writes the user-layout output. Compile-time so that every way
// the forward is issued -- as the module issues it (3), caller-packed X (2), sequence-major in and out (0) -- is a kernel symbol of its own in a
// trace (profiles/*kernel_stats.csv reproduce the bench line's roofline fraction), and the paths not taken cost neither code nor registers.
// MODE 0: the recurrence (GATED: with the scalar time gates gi_t, gf_t known before step 0 -- they read (x_t, h0), never h_{t-1}).
// MODE 1: the time gates' pre-pass for BOTH gates at once: an item (t, b) is one "sequence" of one step whose cell has 2 F outputs -- chunks
//         0 .. F/32-1 the input gate's sub-cell, the rest the forget gate's (weights and biases concatenated by the caller) -- so the operand
//         (x_t, h0) is loaded, and with VAR bit 0 laid out, ONCE per gate pair; epilogue: c = tanh(pre), partial <c, read-out weights>.
// MODE 2: the BPTT data chain dpre_{t-1} = (gf_t sum_k S^k (dpre_t B_k^T) + dH_{t-1}) (1 - h_{t-1}^2) on the adjoint graph with the transposed
//         state taps: a state-only operand (XS = 0) that is entirely this launch's own output -- the last chunk handed over in registers,
//         the first re-read --, the epilogue's operands dH_{t-1}, h_{t-1} requested at the chunk's start; VAR bit 0: lays out dH.
// R1: rank-1-weighted graph S[m][n] = a[m] b[n] (normalised adjacencies: graph.fused_plan_rank1) on the plan of its 0/1 pattern: a hop is
//     b[n] sum_{m in N(n)} (a[m] v[m]) -- every image write is scaled by a (seed and hop outputs), a hop's sums by b BEFORE its tap is added
//     (so every wave streams first), the taps carry no w^k (uniform_w = 1).
// SPLIT: batches that leave half of the chip idle (65 <= B <= 128 at F = 64): a sequence's F/32 chunks run as F/32 WORKGROUPS, one launch per time
//     step (the launch boundary is the hand-over between them: no cross-workgroup wait inside a launch). Each loads the whole operand, runs its
//     chunk, stores its 32 features and lays out its share of the next step's input; nothing is kept in registers across steps.
template <int K, int HS, int XS, int VAR, int MODE = 0, bool GATED = false, bool R1 = false, bool SPLIT = false>
__global__ __launch_bounds__(STHREADS) void fused_seq32_kernel(const Seq32Args a) {
  static_assert(!R1 || !SPLIT, "rank-1 graphs: every mode of the persistent form (the BPTT chain takes the adjoint plan: the factors swap)");
  static_assert(!SPLIT || ((MODE == 0 || MODE == 2) && !R1 && HS > 1), "split sequences: the (un-gated or time-gated) forward and the BPTT chain, with more than one chunk");
  constexpr bool ITEMS = (MODE == 1 || MODE == 3);      // one-step items (t, b) instead of sequences
  constexpr bool SONLY = (MODE == 2 || MODE == 4);      // state-only operand that is this launch's own output (XS = 0)
  constexpr bool PKV = (VAR & 1) != 0, USERV = (VAR & 2) != 0 && (MODE == 0 || MODE == 4);
  static_assert(MODE == 0 || ((MODE >= 1 && MODE <= 4) && !GATED), "modes");
  static_assert(MODE < 3 || (!R1 && !SPLIT), "modes 3, 4: uniform-weight graphs, persistent form");
  using M = Seq32Map<K, HS, XS>;
  constexpr int KS = HS + XS;
  constexpr int F = 32 * HS, G = 32 * XS;
  constexpr int NCH = (MODE == 1) ? 2 * HS : HS;  // 32-feature output chunks
  constexpr int PL = M::PL, WOFF = M::WOFF, WB = M::WB, COL_OFF = M::COL_OFF, RS2 = M::RS2;
  constexpr int NPCK = M::NPCK, NRND = NP / NPCK, NH = NCH * (K - 1), RPH = (NRND + NH - 1) / NH;      // pack rounds per step / hops per step / rounds per hop
  constexpr int PKROWS = SONLY ? F : G;
  static_assert(STILES == 8 && GCRNN_HOP_ASM && K >= 2 && (SONLY ? XS == 0 : XS > 0), "generated hop stream: 8 tiles per wave; the chain's operand is the state alone");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int entries = a.entries, B = a.B, N = a.N;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  constexpr int NSPL = SPLIT ? HS : 1;
  const int wg_seq = (int)blockIdx.x / NSPL, chunk0 = SPLIT ? (int)blockIdx.x % NSPL : 0, gseq = (int)gridDim.x / NSPL;
  if (wg_seq >= B) return;

  // once per launch: tile tables, and -- by LDS-DMA, all pieces in flight together -- the column image and chunk 0's weights
  int tbeg[STILES], tend[STILES];
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
    tbeg[i] = a.tile_off[wave * STILES + i];
    tend[i] = a.tile_off[wave * STILES + i + 1];
  }
  // slot table node << 16 | row16 << 5 | hswz << 4 of every tile slot, in LDS: a lane reads its slot's word where it needs it (eight
  // live registers would not survive the hops: operand 128 + accumulators 64 + the stream's window)
  char* wtab = smem + COL_OFF + entries * 32 + GCRNN_HOP_COLUMN_PAD;
  for (int idx = tid; idx < NP; idx += STHREADS) reinterpret_cast<int32_t*>(wtab)[idx] = a.tile_nodes[idx];
  [[maybe_unused]] float* r1tab = reinterpret_cast<float*>(wtab + NP * 4);      // R1: a then b, in SLOT order (slot = (wave * STILES + tile) * 16 + row: one read, no index)
  if constexpr (R1) {
    for (int idx = tid; idx < NP; idx += STHREADS) {
      const int nd = a.tile_nodes[idx] >> 16;
      r1tab[idx] = nd < NP ? a.r1a[nd] : 0.f;
      r1tab[NP + idx] = nd < NP ? a.r1b[nd] : 0.f;
    }
  }
  // (one asm statement: eight reads in flight, one wait; volatile so that the words are re-read at every use, not kept)
  auto slot_words = [&](int ln, int (&w)[STILES]) {
    const uint32_t ad = (uint32_t)(COL_OFF + entries * 32 + GCRNN_HOP_COLUMN_PAD) + (uint32_t)((wave * STILES * 16 + (ln & 15)) * 4);
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:64\n\tds_read_b32 %2, %8 offset:128\n\tds_read_b32 %3, %8 offset:192\n\t"
                 "ds_read_b32 %4, %8 offset:256\n\tds_read_b32 %5, %8 offset:320\n\tds_read_b32 %6, %8 offset:384\n\tds_read_b32 %7, %8 offset:448\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(ad));
    const int lb = (((ln >> 4) >> 1) << 4) | (((ln >> 4) & 1) << 3);      // xor this lane's (half, piece)
#pragma unroll
    for (int i = 0; i < STILES; ++i) w[i] ^= lb;
  };
  // R1: the accumulators of this wave's tiles times a per-node factor -- kind 0: b, 1: 1 / b, 2: b (b = 0, a node without in-neighbours: 1 for
  // kinds 1 and 2). Four tiles at a time: the eight slot words at once cost registers some instantiations do not have.
  [[maybe_unused]] auto r1_scale = [&](f32x4 (&ac)[STILES][2], int kind) __attribute__((always_inline)) {
    if constexpr (R1) {
      const uint32_t ad = (uint32_t)(COL_OFF + entries * 32 + GCRNN_HOP_COLUMN_PAD + NP * 4 + NP * 4) + (uint32_t)((wave * STILES * 16 + (lane_now() & 15)) * 4);      // b, slot order
      constexpr int NB = (GATED || ITEMS || K == 4) ? 1 : 4;      // factors read at a time (the instantiations at the register limit take them one by one: a few hundred cycles per hop)
#pragma unroll
      for (int h4 = 0; h4 < STILES; h4 += NB) {
        float b4[NB];
        if constexpr (NB == 4)
          asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:64\n\tds_read_b32 %2, %4 offset:128\n\tds_read_b32 %3, %4 offset:192\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(b4[0]), "=&v"(b4[1]), "=&v"(b4[2]), "=&v"(b4[3])
                       : "v"(ad + (uint32_t)(h4 * 64)));
        else
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b4[0]) : "v"(ad + (uint32_t)(h4 * 64)));
#pragma unroll
        for (int e = 0; e < NB; ++e) {
          const float bv = b4[e];
          const float f = kind == 0 ? bv : (bv == 0.f ? 1.f : (kind == 1 ? __builtin_amdgcn_rcpf(bv) : bv));
          ac[h4 + e][0] *= f; ac[h4 + e][1] *= f;
        }
      }
    }
  };

  {
    const int cbytes = entries * 32;
    const char* csrc = reinterpret_cast<const char*>(a.ell_col4);
    for (int p = wave; p * 1024 < cbytes; p += SWAVES)
      if (p * 1024 + lane * 16 < cbytes)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(csrc + p * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + COL_OFF + p * 1024), 16, 0, 0);
    const char* wsrc = reinterpret_cast<const char*>(a.wpack) + (size_t)chunk0 * WB;      // (SPLIT: this workgroup's chunk)
    for (int p = wave; p < WB / 1024; p += SWAVES)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + p * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + WOFF + p * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // zeros behind the column image: the stream's running column pointer is not clamped (GCRNN_HOP_COLUMN_PAD)
  if (tid < GCRNN_HOP_COLUMN_PAD / 4) reinterpret_cast<uint32_t*>(smem + COL_OFF + entries * 32)[tid] = 0u;
  float* lbias = reinterpret_cast<float*>(smem + M::BIAS_OFF);
  if (tid < NCH * 32) lbias[tid] = a.bias ? a.bias[tid] : 0.f;
  __syncthreads();

  if (a.stagger > 0) {
    // All 256 workgroups run the same phases at the same time, so the chip's HBM sees every CU's operand requests / state stores / row
    // stores as bursts (11 B per clock and CU when all ask at once: profiles/r04_seq32_stamps_*.txt). A one-time offset per group of
    // workgroups persists over all T steps of this launch (same code, same speed) and spreads those bursts over the step.
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long d = (unsigned long long)((blockIdx.x >> 3) & 7) * (unsigned long long)a.stagger;
    while (__builtin_amdgcn_s_memtime() - t0 < d) __builtin_amdgcn_s_sleep(16);
  }
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  if (lds0 != 0) __builtin_trap();        // the asm stream forms gather addresses from column words: the image must sit at LDS address 0
  const uint32_t lds_col = lds0 + COL_OFF;
  char* xtile = wtab + NP * 4 + (R1 ? 2 * NP * 4 : 0);

  // wave-uniform, as a scalar integer (a lane mask would also be parked in a vector register)
  const int skip_hi = __builtin_amdgcn_readfirstlane((MODE == 3 || (MODE == 1 && a.flags && a.flags[0] != 0)) ? 1 : 0);      // (MODE 3: the operand is [0 | x_t])
  const bool skip_h = skip_hi != 0;
  // ---- the operand of a sequence and step: every B fragment of the wave, resident for all chunks ----------------------------------
  bf16x8 bfr[STILES][KS];
  // the operand of item / sequence `bb` (its first step): at the top of the loop, or -- MODE 1 -- requested behind the previous item's last state
  // stores, when the operand registers have just died (the items of the pre-pass are one step each: without it every item starts with a
  // wait for its whole operand)
  auto load_first_operand = [&](int b) {
    // (MODE 1 with an all-zero h0: a zero-length descriptor -- the loads return zeros and cost nothing)
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.hfirst), 0, skip_h ? 0 : (ITEMS ? a.hmod : B) * (NP * F * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.x0), 0, XS > 0 ? B * (NP * G * 2) : 0, 0x00020000);
    const int bh = ITEMS ? __builtin_amdgcn_readfirstlane(b % a.hmod) : b;
    const int ln0 = lane_now(), qo = ln0 >> 4;
    int sw[STILES];
    slot_words(ln0, sw);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        const int w = sw[i];
        if (s < HS)
          bfr[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, (w >> 16) * (F * 2) + 16 * qo + 64 * s, bh * (NP * F * 2), 0));
        else
          bfr[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (w >> 16) * (G * 2) + 16 * qo + 64 * (s - HS), b * (NP * G * 2), 0));
      }
    }
  };
  bool have_operand = false;      // (wave-uniform)
  for (int b = wg_seq; b < B; b += gseq) {
  if (!have_operand) load_first_operand(b);
  have_operand = false;
#pragma unroll 1
  for (int step = 0; step < a.nsteps; ++step) {
    const bool fin = (MODE == 2) && a.final_raw && step == a.nsteps - 1;      // (chain: the d h0 step)
    uint16_t* hout = fin ? a.final_out : (a.out0 ? a.out0 + (int64_t)step * a.ostride : nullptr);
    [[maybe_unused]] const uint16_t* ep_dh = (SONLY && !fin && a.dh0_) ? a.dh0_ + (int64_t)step * a.dhstride : nullptr;      // (MODE 4: Yx_t)
    [[maybe_unused]] const float* ep_ng = (MODE == 4) ? a.ng0 + (int64_t)step * a.ngstride : nullptr;
    [[maybe_unused]] const uint16_t* ep_h = (MODE == 2) ? (fin ? a.final_h : (a.hs0 ? a.hs0 + (int64_t)step * a.hsstride : nullptr)) : nullptr;
    [[maybe_unused]] float* gpart = (MODE == 2 && a.gpart0) ? a.gpart0 + (int64_t)step * a.gpartstride : nullptr;
    [[maybe_unused]] float gsc = 1.f;
    if constexpr (MODE == 2) {
      if (a.gsc0) gsc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.gsc0[(int64_t)step * a.gscstride + b])));
    }
    float gin = 1.f, gfo = 1.f, gratio = 1.f;
    if constexpr (MODE == 4) {      // scalar time gates on top of the node gates (applied in the epilogue)
      if (a.gi0) {
        gin = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.gi0[(int64_t)step * a.gstride + b])));
        gfo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.gf0[(int64_t)step * a.gstride + b])));
      }
    }
    if constexpr (GATED) {      // (wave-uniform: kept in scalar registers -- three vector registers live across the hops are three too many)
      gin = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.gi0[(int64_t)step * a.gstride + b])));
      gfo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.gf0[(int64_t)step * a.gstride + b])));
      gratio = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, gfo / fmaxf(gin, 1e-30f))));
    }
    const uint16_t* aux1 = (USERV && a.a1) ? (a.a1_last_only ? (step == a.nsteps - 1 ? a.a1 : nullptr) : a.a1 + (int64_t)step * a.a1stride) : nullptr;
    // Inline pack: the 128-node rounds of a step's input are laid out ONE HOP ahead of "during the step before": hop i of step t takes
    // round i + 1 of x_{t+1}, the step's last hop round 0 of x_{t+2}. x_{t+1} is then complete (stored, waited for, behind a barrier) when
    // step t's last epilogue requests it -- and recent: an input laid out a whole step earlier has left the caches by the time it is read
    // (the requests of all CUs then run at HBM latency: ~190 instead of ~70 units per step, profiles/r04_seq32_stamps_*). The caller lays
    // out steps 0 and 1.
    const bool pk_any = PKV && a.pk_src0 != nullptr;
    const int ubstride = a.ubstride;
    // (MODE 1: the pack lays out the operand of the NEXT item of this workgroup's loop, item nb = (t', b') = (nb / hmod, nb % hmod))
    const int nb = b + (int)gridDim.x;
    const int nbq = ITEMS ? __builtin_amdgcn_readfirstlane(nb / a.hmod) : 0, nbr = ITEMS ? __builtin_amdgcn_readfirstlane(nb - nbq * a.hmod) : 0;      // (scalar registers)
    const int64_t pk_soff = ITEMS ? (int64_t)nbr * a.pk_stride + (int64_t)nbq * a.pksrc_stride : (int64_t)b * a.pk_stride;
    const int pk_db = ITEMS ? nb : b;
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(hout, 0, hout ? B * (NP * F * 2) : 0, 0x00020000);
    [[maybe_unused]] const bool stamp_on = (step == (a.nsteps > 2 ? a.nsteps - 3 : 0)) && b == wg_seq;      // a typical step (diagnostic builds)
    GCRNN_STAMP32(0);
    const bool more = step + 1 < a.nsteps;      // the next step's operand is requested at the start of this step's last epilogue
    const __amdgpu_buffer_rsrc_t rsrc_xn = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(XS > 0 ? a.x0 + (int64_t)(step + 1) * a.xstride : nullptr), 0, (more && XS > 0) ? B * (NP * G * 2) : 0, 0x00020000);

    f32x4 acc[STILES][2];
    // acc[i][h] += W_tap(c, half h) [h|x]^T for the wave's 8 tiles: one weight fragment feeds 8 independent MFMA chains
    auto taps = [&](int tap, [[maybe_unused]] bool on_sums = false) {
#ifdef GCRNN_SEQ32_EXPERIMENT_NO_HOP_TAPS      // timing experiment, WRONG results: what would a hop cost if its tap were free (hidden inside the stream)?
      if (tap != K - 1) return;
#endif
      const int ln = lane_now();                       // (the fragment address is re-derived per call, not kept -- or spilled -- across the hops)
      const uint32_t wofs = (uint32_t)WOFF + (uint32_t)ln * 16u;
      if constexpr (GATED) {
        // gi (x W_x) + gf (h W_h) on ONE accumulator chain per half: h-chain, scale by gf / gi, continue with x, scale by gi (gi = sigmoid(.) > 0;
        // the wave-uniform guard covers an underflowed gate). Waves 4..7 (tap first): the accumulators are ZERO on entry and the stream then
        // adds the hop's sums to the finished tap.
        // on_sums (the waves that stream first): the accumulators hold the hop's sums S; S + gf hW + gi xW comes out of the same chain started
        // from S / gf (three more roundings at 1e-7 relative; an underflowed forget gate skips the h-chain instead)
        const bool xpart = gin > 1e-30f;
        const bool hpart = !on_sums || gfo > 1e-30f;
        if (on_sums) {
          const float pre = hpart ? 1.f / gfo : (xpart ? 1.f / gin : 1.f);
#pragma unroll
          for (int i = 0; i < STILES; ++i) { acc[i][0] *= pre; acc[i][1] *= pre; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (hpart) {
#pragma unroll
            for (int s = 0; s < HS; ++s) {
              const bf16x8 afr = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)(((tap * 2 + h) * KS + s) * 1024)));
#pragma unroll
              for (int i = 0; i < STILES; ++i) acc[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[i][s], acc[i][h], 0, 0, 0);
            }
          }
          if (hpart) {
#pragma unroll
            for (int i = 0; i < STILES; ++i) acc[i][h] *= (xpart ? gratio : gfo);
          }
          if (xpart) {
#pragma unroll
            for (int s = HS; s < KS; ++s) {
              const bf16x8 afr = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)(((tap * 2 + h) * KS + s) * 1024)));
#pragma unroll
              for (int i = 0; i < STILES; ++i) acc[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[i][s], acc[i][h], 0, 0, 0);
            }
            if constexpr (GATED) {
#pragma unroll
              for (int i = 0; i < STILES; ++i) acc[i][h] *= gin;
            }
          }
        }
      } else if constexpr (ITEMS) {
        // (gate pre-pass / filter-output pass; with an all-zero h0 the state half of the operand is neither loaded nor multiplied)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            if (s < HS && skip_h) continue;
            const bf16x8 afr = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)(((tap * 2 + h) * KS + s) * 1024)));
#pragma unroll
            for (int i = 0; i < STILES; ++i) acc[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[i][s], acc[i][h], 0, 0, 0);
          }
      } else {
      // (the next fragment is requested before the current one's eight MFMAs)
      uint4 af[2];
      af[0] = *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)((tap * 2 * KS) * 1024));
#pragma unroll
      for (int hs = 0; hs < 2 * KS; ++hs) {
        const int h = hs / KS, s = hs - h * KS;
        if (hs + 1 < 2 * KS) af[(hs + 1) & 1] = *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)((tap * 2 * KS + hs + 1) * 1024));
        const bf16x8 afr = __builtin_bit_cast(bf16x8, af[hs & 1]);
#pragma unroll
        for (int i = 0; i < STILES; ++i) acc[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[i][s], acc[i][h], 0, 0, 0);
      }
      }
    };
    // R1, tprime: the accumulators hold t / b (a wave that taps first carries the hop in t' = t / b -- see the hop): the image's a (.) t is (a b) (.) t'
    auto put = [&](auto tpc) __attribute__((always_inline)) {
      constexpr bool tprime = decltype(tpc)::value;
      int sw[STILES];
      slot_words(lane_now(), sw);
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        const int wv = sw[i];
        if constexpr (R1) {      // the image holds a (.) v
          const int slot = (wave * STILES + i) * 16 + (lane_now() & 15);
          float av = r1tab[slot];
          if constexpr (tprime) { const float bv = r1tab[NP + slot]; av *= (bv == 0.f ? 1.f : bv); }
          state_put<true>(reinterpret_cast<float*>(smem), wv, acc[i][0] * av);
          state_put<true>(reinterpret_cast<float*>(smem + PL), wv, acc[i][1] * av);
        } else {
          state_put<true>(reinterpret_cast<float*>(smem), wv, acc[i][0]);
          state_put<true>(reinterpret_cast<float*>(smem + PL), wv, acc[i][1]);
        }
      }
    };
    // seed of a chunk: tap K-1 goes straight into the hop image (every wave has left the image: the barrier before)
    auto seed = [&]() {
#pragma unroll
      for (int i = 0; i < STILES; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      taps(K - 1);
      put(std::false_type{});
    };
    seed();

#pragma unroll 1
    for (int chunk = chunk0; chunk < (SPLIT ? chunk0 + 1 : NCH); ++chunk) {
      // opaque per chunk: index arithmetic is re-derived from it inside the loop -- hoisted out it would have to be spilled
      // (nothing derived from the lane id lives across the hops: every use below re-derives it, lane_now())
      const bool last = SPLIT || chunk == NCH - 1;
      lds_barrier();      // the seed is in the image
      GCRNN_STAMP32(1 + chunk * 24);

      // virtual round v of this step (NRND per step, shifted by one hop) -> (target step, round); false: nothing to lay out
      auto pack_target = [&](int v, int& tgt, int& rnd) -> bool {
        if constexpr (ITEMS) {      // the workgroup's next item, round v (its operand is read when that item starts)
          rnd = v; tgt = 0;
          return pk_any && v < NRND && nb < B;
        }
        if constexpr (SPLIT) {      // one step per launch: this workgroup's rounds (those of its chunk's hops) of the NEXT step's input, the host has set the pointers
          rnd = v; tgt = 0;
          return pk_any && v < NRND;
        }
        if constexpr (MODE == 2) {      // step i lays out the upstream gradient step i + 1's epilogue reads (the caller laid out step 0's)
          rnd = v; tgt = step + 1;
          return pk_any && v < NRND && tgt < a.nsteps - (a.final_raw ? 1 : 0);
        }
        v += RPH;
        const int wrap = v >= NRND ? 1 : 0;
        rnd = v - wrap * NRND;
        tgt = step + 1 + wrap;
        return pk_any && tgt >= 2 && tgt < a.nsteps;
      };
      // inline pack, round rnd of step tgt: x_tgt[:, rnd * 128 .. + 127] (user layout, rows = features) by LDS-DMA into the tile ...
      auto pack_issue = [&](int v) {
        int tgt, rnd;
        if (!pack_target(v, tgt, rnd)) return;
        const uint16_t* pk_src = a.pk_src0 + ((ITEMS || SPLIT) ? 0 : (int64_t)tgt * a.pksrc_stride);
        constexpr int PPR = NPCK / 8, PIECES = PKROWS * PPR;
        static_assert(PIECES % STHREADS == 0, "whole pieces per thread");
        const uint16_t* xsrc = pk_src + pk_soff + rnd * NPCK;
        const int tl = wave * 64 + lane_now();
#pragma unroll
        for (int i = 0; i < PIECES / STHREADS; ++i) {
          const int id = i * STHREADS + tl;
          const int row = id / PPR, cs = id - row * PPR;
          const int col = (cs - (row >> 3)) & (PPR - 1);
          if (rnd * NPCK + col * 8 < N)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + (int64_t)row * N + col * 8),
                                             (__attribute__((address_space(3))) void*)(xtile + (i * STHREADS + wave * 64) * 16), 16, 0, GCRNN_SEQ32_NT_DMA);
        }
      };
      // ... and out of it transposed: whole sequence-major rows [node][features]
      auto pack_drain = [&](int v) {
        int tgt, rnd;
        if (!pack_target(v, tgt, rnd)) return;
        uint16_t* pk_dst = a.pk_dst0 + ((ITEMS || SPLIT) ? 0 : (int64_t)tgt * a.pkdst_stride);
        constexpr int PCS = PKROWS / 8, RI = PCS * NPCK / STHREADS;
        static_assert(PCS * NPCK % STHREADS == 0, "whole row pieces per thread");
        typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;
        const __amdgpu_buffer_rsrc_t rsrc_pk = __builtin_amdgcn_make_buffer_rsrc(pk_dst, 0, B * (NP * PKROWS * 2), 0x00020000);
        const int tl = wave * 64 + lane_now();
        u32x4_t vv[RI];
        // the tile's 2-byte columns of this thread's row piece(s): ALL reads in flight, one wait (hipcc waits after every pair: eight LDS round
        // trips per piece, ~8 of a hop's ~19 units of write-back; d16 loads cannot merge the halves -- with SRAM-ECC they rewrite the whole register)
        static_assert(RI == 1 || RI == 2, "one or two row pieces per thread");
        uint32_t h16[RI][8];
        uint32_t sa[RI];
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          const int id = i * STHREADS + tl;
          const int nl = id / PCS, pc = id - nl * PCS;
          sa[i] = (uint32_t)(xtile - smem) + (uint32_t)((pc * 8) * (NPCK * 2) + ((nl + 8 * pc) & (NPCK - 1)) * 2);
        }
        static_assert(NPCK * 2 == 256, "row pitch of the pack tile in the asm offsets");
        if constexpr (RI == 2) {
          asm volatile("ds_read_u16 %0, %16\n\tds_read_u16 %1, %16 offset:256\n\tds_read_u16 %2, %16 offset:512\n\tds_read_u16 %3, %16 offset:768\n\t"
                       "ds_read_u16 %4, %16 offset:1024\n\tds_read_u16 %5, %16 offset:1280\n\tds_read_u16 %6, %16 offset:1536\n\tds_read_u16 %7, %16 offset:1792\n\t"
                       "ds_read_u16 %8, %17\n\tds_read_u16 %9, %17 offset:256\n\tds_read_u16 %10, %17 offset:512\n\tds_read_u16 %11, %17 offset:768\n\t"
                       "ds_read_u16 %12, %17 offset:1024\n\tds_read_u16 %13, %17 offset:1280\n\tds_read_u16 %14, %17 offset:1536\n\tds_read_u16 %15, %17 offset:1792\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(h16[0][0]), "=&v"(h16[0][1]), "=&v"(h16[0][2]), "=&v"(h16[0][3]), "=&v"(h16[0][4]), "=&v"(h16[0][5]), "=&v"(h16[0][6]), "=&v"(h16[0][7]),
                         "=&v"(h16[RI - 1][0]), "=&v"(h16[RI - 1][1]), "=&v"(h16[RI - 1][2]), "=&v"(h16[RI - 1][3]), "=&v"(h16[RI - 1][4]), "=&v"(h16[RI - 1][5]), "=&v"(h16[RI - 1][6]), "=&v"(h16[RI - 1][7])
                       : "v"(sa[0]), "v"(sa[RI - 1]));
        } else {
          asm volatile("ds_read_u16 %0, %8\n\tds_read_u16 %1, %8 offset:256\n\tds_read_u16 %2, %8 offset:512\n\tds_read_u16 %3, %8 offset:768\n\t"
                       "ds_read_u16 %4, %8 offset:1024\n\tds_read_u16 %5, %8 offset:1280\n\tds_read_u16 %6, %8 offset:1536\n\tds_read_u16 %7, %8 offset:1792\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(h16[0][0]), "=&v"(h16[0][1]), "=&v"(h16[0][2]), "=&v"(h16[0][3]), "=&v"(h16[0][4]), "=&v"(h16[0][5]), "=&v"(h16[0][6]), "=&v"(h16[0][7])
                       : "v"(sa[0]));
        }
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          const int id = i * STHREADS + tl;
          const int nl = id / PCS;
          const bool ok = rnd * NPCK + nl < N;
          uint32_t w4[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) w4[jj] = h16[i][2 * jj] | (h16[i][2 * jj + 1] << 16);
          vv[i] = u32x4_t{ok ? w4[0] : 0u, ok ? w4[1] : 0u, ok ? w4[2] : 0u, ok ? w4[3] : 0u};
        }
#pragma unroll
        for (int i = 0; i < RI; ++i) asm volatile("" : "+v"(vv[i]));          // every piece in its own tuple before the first store (gfx950 store-data hazard)
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          const int id = i * STHREADS + tl;
          const int nl = id / PCS, pc = id - nl * PCS;
          __builtin_amdgcn_raw_buffer_store_b128(vv[i], rsrc_pk, (rnd * NPCK + nl) * (PKROWS * 2) + pc * 16 + pk_db * (NP * PKROWS * 2), 0, 0);
        }
      };

      // Weight fragments of chunk wc, tap `tap` -> their place in LDS (2 KS pieces of 1 KB). One tap at a time, into a place whose last
      // reader is a barrier behind: during hop j the NEXT chunk's tap K - j (read by this chunk's seed / hop j - 1), during hop 1 also this
      // chunk's own tap 0 (its place still held the previous chunk's, read at that chunk's last hop). Every piece is covered by the hop's
      // vmcnt(0) + barrier, and no LDS-DMA is pending during an epilogue -- hipcc orders each LDS access of a wave behind its pending
      // LDS-DMA pieces with a vmcnt(0), which in the last epilogue would also wait for the next operand's requests.
      auto r0_of = [&](int j) { return (chunk * (K - 1) + (j - 1)) * RPH; };
      auto weights_issue = [&](int wc, int tap) {
        const char* wsrc = reinterpret_cast<const char*>(a.wpack) + (size_t)wc * WB + (size_t)tap * (2 * KS * 1024);
        const int lane = lane_now();
#pragma unroll
        for (int i = 0; i < (2 * KS + SWAVES - 1) / SWAVES; ++i) {
          const int piece = i * SWAVES + wave;
          if (piece < 2 * KS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + WOFF + tap * (2 * KS * 1024) + piece * 1024), 16, 0, 0);
        }
      };
      // MODE 2: the epilogue's operands h_{t-1}, dH_{t-1} of this lane's (node, 8 features) per tile. Requested in the write-back phases of hops
      // 1 .. EPH (behind the hop's vmcnt(0) + barrier), a share of the tiles each: they land under the NEXT hop's stream and are covered by its
      // vmcnt(0). Requested in front of hop 1 (round 4's first form; still so for K < 4) the waves that tap first met hipcc's vmcnt(0) in front
      // of the tap's weight reads a few hundred cycles later and sat out the whole HBM latency: hop 1 took ~200 units instead of ~75
      // (profiles/r04_seq32_stamps_chain.txt).
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4e_t;
      [[maybe_unused]] u32x4e_t eph[MODE == 2 ? STILES : 1], epg[SONLY ? STILES : 1];
      [[maybe_unused]] float epn[MODE == 4 ? STILES : 1][2];      // MODE 4: this lane's node's input / forget gate per tile
#ifndef GCRNN_SEQ32_EP_FIRST
#define GCRNN_SEQ32_EP_FIRST 1      // the requests in front of the hop's write-back (0: behind it; same-box A/B: profiles/r04_seq32_chain_requests_ab.txt)
#endif
#ifndef GCRNN_SEQ32_EP_EARLY
#define GCRNN_SEQ32_EP_EARLY 0      // 1: all of them in front of hop 1 (A/B)
#endif
      constexpr int EPH = (SONLY && K >= 4 && !GCRNN_SEQ32_EP_EARLY) ? K - 2 : 0;
      auto ep_request = [&](auto i0c, auto i1c) __attribute__((always_inline)) {
        if constexpr (MODE == 4) {
          constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
          const int lq = lane_now();
          int swp[STILES];
          slot_words(lq, swp);
          const __amdgpu_buffer_rsrc_t rsrc_eg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ep_dh), 0, ep_dh ? B * (NP * F * 2) : 0, 0x00020000);
          const __amdgpu_buffer_rsrc_t rsrc_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep_ng), 0, (int)(a.nghalf * 8), 0x00020000);      // ni | nf, 2 x nghalf floats; rows >= N: out of range -> 0
#pragma unroll
          for (int i = I0; i < I1; ++i) {
            const int node = swp[i] >> 16;
            epg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_eg, node * (F * 2) + (chunk * 32 + (lq >> 4) * 8) * 2, b * (NP * F * 2), 0);
            const int go = node < N ? (b * N + node) * 4 : 0x7ffffff0;
            epn[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_n, go, 0, 0));
            epn[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_n, go, (int)(a.nghalf * 4), 0));
          }
        }
        if constexpr (MODE == 2) {
          constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
          const int lq = lane_now();
          int swp[STILES];
          slot_words(lq, swp);
          const __amdgpu_buffer_rsrc_t rsrc_eh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ep_h), 0, ep_h ? B * (NP * F * 2) : 0, 0x00020000);
          const __amdgpu_buffer_rsrc_t rsrc_eg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ep_dh), 0, ep_dh ? B * (NP * F * 2) : 0, 0x00020000);
#pragma unroll
          for (int i = I0; i < I1; ++i) {
            const int eoff = (swp[i] >> 16) * (F * 2) + (chunk * 32 + (lq >> 4) * 8) * 2;
            eph[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_eh, eoff, b * (NP * F * 2), 0);
            epg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_eg, eoff, b * (NP * F * 2), 0);
          }
        }
      };
      if constexpr (SONLY && EPH == 0) ep_request(std::integral_constant<int, 0>{}, std::integral_constant<int, STILES>{});
      // ---- Horner hops on the two-plane bf16 image; the tap a hop adds accumulates onto its sums --------------------------------
      auto hop = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        auto dma_issue = [&]() {      // this wave's LDS-DMA pieces of the hop, right in front of its stream
          if (NCH > 1 && !SPLIT) {
            weights_issue((chunk + 1) % NCH, K - j);
            if (j == 1 && K > 2) weights_issue(chunk, 0);
          }
          if (r0_of(j) < NRND) pack_issue(r0_of(j));
        };
        const int r0 = r0_of(j);      // this hop's first pack round
        // The two waves of a SIMD take the hop's two phases in opposite order: issue arbitration favours the older wave, so with all
        // eight streaming at once waves 4..7 finish their streams ~30 % after waves 0..3 and only then start their tap MFMAs while the
        // others wait at the barrier (profiles/r04_seq32_stamps_native_same_order.txt). The tap of a hop does not depend on its sums
        // (D = A B + D accumulates either way), so waves 4..7 evaluate it FIRST: their MFMAs run beside the streams of waves 0..3, which
        // then have the LDS to four waves, and the other way round afterwards.
#pragma unroll
        for (int i = 0; i < STILES; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // (the pack's LDS-DMA pieces go out right before the wave's stream: hipcc orders every later LDS read of the wave behind them with a
        //  vmcnt(0) -- behind the stream that wait is free, in front of the tap's weight reads it would expose the pieces' whole latency)
        // R1 (S = a b^T on its pattern): a hop is t_j = u_j + b (.) sum (a (.) t_{j+1}). A wave that streams first scales its sums by b and adds the
        // tap (it carries t; image write: a (.) t). A wave that taps first cannot scale the sums alone once the tap is in the accumulator -- it
        // carries t' = t / b instead: t'_j = u_j / b + sum, image write (a b) (.) t', and b (.) t' after the chunk's last hop (b = 0: no
        // in-neighbours, the sum is empty: factor 1). All in fp32 registers; round 4's first R1 made every wave stream first (-13 %).
        const bool stream_first = (!GATED || !GCRNN_SEQ32_GATED_TAPS_FIRST) && (wave < SWAVES / 2 || GCRNN_SEQ32_SAME_ORDER || (R1 && GCRNN_SEQ32_R1_STREAM_FIRST));
        if (stream_first) {
          dma_issue();
          GCRNN_HOP_ASM_WIDE32_STREAM(acc);
          r1_scale(acc, 0);      // R1: sums of a (.) v over the in-neighbours, times b[n]: then the tap
          GCRNN_STAMP32(1 + chunk * 24 + 4 * (j - 1) + 1);
          if (j == 2 && chunk == 0) GCRNN_STAMP32_WAVE(56);
          taps(K - 1 - j, true);
        } else {
          taps(K - 1 - j);
          r1_scale(acc, 1);      // R1: the tap in units of b (t' = t / b)
          if (j == 2 && chunk == 0) GCRNN_STAMP32_WAVE(56);
          dma_issue();
#if GCRNN_SEQ32_YOUNG_PRIO      // experiment: the younger wave of each SIMD streams at raised issue priority (it starts its stream a tap later)
          __builtin_amdgcn_s_setprio(GCRNN_SEQ32_YOUNG_PRIO);
#endif
          GCRNN_HOP_ASM_WIDE32_STREAM(acc);
#if GCRNN_SEQ32_YOUNG_PRIO
          __builtin_amdgcn_s_setprio(0);
#endif
        }
        GCRNN_STAMP32(1 + chunk * 24 + 4 * (j - 1) + 2);
        if (j == 2 && chunk == 0) GCRNN_STAMP32_WAVE(64);
        if (pk_any || (NCH > 1 && !SPLIT)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA pieces have landed (they had the stream)
        lds_barrier();      // every wave has left the image (and the weights, after the last hop); every piece of the pack tile is in
        GCRNN_STAMP32(1 + chunk * 24 + 4 * (j - 1) + 3);
#if GCRNN_SEQ32_EP_FIRST      // (A/B: the requests in front of the write-back instead of behind it)
        if constexpr (SONLY && EPH > 0 && j <= EPH)
          ep_request(std::integral_constant<int, (j - 1) * STILES / (EPH > 0 ? EPH : 1)>{}, std::integral_constant<int, j * STILES / (EPH > 0 ? EPH : 1)>{});
#endif
        if (j < K - 1) {
          if (R1 && !stream_first) put(std::true_type{});
          else put(std::false_type{});
        }
        if constexpr (R1 && j == K - 1) {
          if (!stream_first) r1_scale(acc, 2);      // the chunk's result: t_0 = b (.) t'_0
        }
        if (K == 2 && NCH > 1 && !SPLIT) weights_issue((chunk + 1) % NCH, 0);      // (K = 2: tap 0's fragments are free only now, and needed at the next chunk's only hop)
        if (r0 < NRND) pack_drain(r0);
#pragma unroll
        for (int e = 1; e < RPH; ++e) {       // (fewer hops than rounds: the extra rounds are not hidden behind a stream)
          if (pk_any && r0 + e < NRND) {
            lds_barrier();
            pack_issue(r0 + e);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            pack_drain(r0 + e);
          }
        }
#if !GCRNN_SEQ32_EP_FIRST
        if constexpr (SONLY && EPH > 0 && j <= EPH)
          ep_request(std::integral_constant<int, (j - 1) * STILES / (EPH > 0 ? EPH : 1)>{}, std::integral_constant<int, j * STILES / (EPH > 0 ? EPH : 1)>{});
#endif
        if (j < K - 1) lds_barrier();      // the image is complete (and the pack tile read)
        GCRNN_STAMP32(1 + chunk * 24 + 4 * (j - 1) + 4);
      };
      seq_static_for(hop, std::make_integer_sequence<int, K - 1>{});

      // ---- epilogue: + 2 b, tanh, bf16; lane (r, q) holds features 32 c + 8 q .. + 7 of its node: ONE 16-byte store per tile ------------
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;
      const int lane = lane_now(), q = lane >> 4, tl = wave * 64 + lane;
      auto request_next_operand = [&]() {
        if (ITEMS || !(last && more)) return;
        // the next step's operand: x_{t+1} (laid out two steps ahead, or by the caller) and the state features of the earlier chunks
        // (stored -- and waited for -- at their chunk's end); the last chunk's come from this epilogue's registers below
        const int qo = lane_now() >> 4;
        int sw[STILES];
        slot_words(lane, sw);
        const __amdgpu_buffer_rsrc_t rsrc_hn = __builtin_amdgcn_make_buffer_rsrc(hout, 0, B * (NP * F * 2), 0x00020000);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (s == HS - 1) continue;
#pragma unroll
          for (int i = 0; i < STILES; ++i) {
            const int w = sw[i];
#ifdef GCRNN_SEQ32_EXPERIMENT_SKIP      // timing experiment, WRONG results: 1 = no state requests, 2 = no input requests (which half of the operand costs the wait?)
            if ((GCRNN_SEQ32_EXPERIMENT_SKIP == 1) == (s < HS)) continue;
#endif
            if (s < HS)
              bfr[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_hn, (w >> 16) * (F * 2) + 16 * qo + 64 * s, b * (NP * F * 2), GCRNN_SEQ32_NT_XLOAD));
            else
              bfr[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_xn, (w >> 16) * (G * 2) + 16 * qo + 64 * (s - HS), b * (NP * G * 2), GCRNN_SEQ32_NT_XLOAD));
          }
        }
      };
      if (GCRNN_SEQ32_OPERAND_AT == 0) request_next_operand();
      if (ITEMS) GCRNN_STAMP32(1 + chunk * 24 + 19);      // (diagnostic builds: start of the epilogue proper)
      float bs[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[h][e] = (MODE >= 3 ? 1.f : (gin + gfo)) * lbias[chunk * 32 + q * 8 + h * 4 + e];      // (modes 3, 4: ONE filter's bias)      // the one bias is added by both filters (graphML.py:2420-2421): 2 b, or (gi + gf) b
      u32x4_t pkd[STILES];
      int swe[STILES];
      slot_words(lane, swe);
      if constexpr (MODE == 1) {
        // gate pre-pass: c = tanh(pre) of this gate's sub-cell (chunk / HS = the gate), partial dot product with its read-out weights
        // [N][F] fp32 (graphML.py:2364-2366: vec over (f, n)), one partial per (chunk, wave) -- the caller adds them in a fixed order; with
        // output arrays the sub-cell's state is stored (bf16) for the gate's BPTT. The weights are shared by every item (L2-resident); two
        // tiles' worth are requested at a time.
        const int gate = chunk / HS, cg = chunk - gate * HS;
        if (a.tapf) {
          // node gates: s_k[n] += sum_f c[n][f] w_k[f] over this chunk's 32 features on the matrix cores -- the lane's packed state (node r, features
          // 8 q .. + 7) IS the B fragment (k = feature, n = node), the taps' three bf16 planes the A fragments (m = tap): rows 4 q' + e of D are
          // taps, columns nodes; the chunk's partial goes to memory (the two chunks of a gate are added by the caller, fixed order)
          typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4t;
          const uint4* tfp = a.tapf + ((size_t)(gate * HS + cg) * 3) * 64 + lane;
          bf16x8 ta[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) ta[pl] = __builtin_bit_cast(bf16x8, tfp[pl * 64]);
          const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(gate ? a.out1 : a.out0, 0, (gate ? a.out1 : a.out0) ? B * (NP * F * 2) : 0, 0x00020000);
          float* so = a.taps_out + (((int64_t)b * 2 + gate) * HS + cg) * (int64_t)a.ntaps * N;
#pragma unroll
          for (int i = 0; i < STILES; ++i) {
            const int node = swe[i] >> 16;
            u32x4t p{0u, 0u, 0u, 0u};
            if (node < N) {
              const f32x4 a0 = acc[i][0], a1 = acc[i][1];
              p[0] = pack2bf(fast_tanh(a0[0] + bs[0][0]), fast_tanh(a0[1] + bs[0][1]));
              p[1] = pack2bf(fast_tanh(a0[2] + bs[0][2]), fast_tanh(a0[3] + bs[0][3]));
              p[2] = pack2bf(fast_tanh(a1[0] + bs[1][0]), fast_tanh(a1[1] + bs[1][1]));
              p[3] = pack2bf(fast_tanh(a1[2] + bs[1][2]), fast_tanh(a1[3] + bs[1][3]));
            }
            if (gate ? a.out1 : a.out0) __builtin_amdgcn_raw_buffer_store_b128(p, rsrc_c, node * (F * 2) + (cg * 32 + q * 8) * 2, b * (NP * F * 2), 0);
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta[pl], __builtin_bit_cast(bf16x8, p), d, 0, 0, 0);
            // D rows 4 q + e = taps, column r = this lane's node of the tile
            if (node < N) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (q * 4 + e < a.ntaps) so[(int64_t)(q * 4 + e) * N + node] = d[e];
            }
          }
        } else {
        const float* gwp = a.gw + (int64_t)gate * N * F + cg * 32 + q * 8;
        uint16_t* cso = gate ? a.out1 : a.out0;
        const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(cso, 0, cso ? B * (NP * F * 2) : 0, 0x00020000);
        float part = 0.f;
        // (the read-out weights of the NEXT two tiles are requested before this pair is evaluated: one L2 round trip per chunk is exposed instead
        //  of four; -0.6 % on the time-gated forward, profiles/r04_seq32_gate_readout_ab.txt. The same weights as bf16 -- half the bytes -- ran
        //  SLOWER: this epilogue, 95 units per chunk against the recurrence's 35 in r04_seq32_stamps_gate_pair.txt, is not bound by their bytes)
#ifndef GCRNN_SEQ32_GATE_W_PIPELINE
#define GCRNN_SEQ32_GATE_W_PIPELINE 1
#endif
        float4 w8s[2][2][2];
        auto w8_load = [&](int set, int i0l) __attribute__((always_inline)) {
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int nd = (swe[i0l + t2] >> 16) < N ? (swe[i0l + t2] >> 16) : N - 1;
            w8s[set][t2][0] = *reinterpret_cast<const float4*>(gwp + (int64_t)nd * F);
            w8s[set][t2][1] = *reinterpret_cast<const float4*>(gwp + (int64_t)nd * F + 4);
          }
        };
        if (GCRNN_SEQ32_GATE_W_PIPELINE) w8_load(0, 0);
#pragma unroll
        for (int i0 = 0; i0 < STILES; i0 += 2) {
          const int wset = GCRNN_SEQ32_GATE_W_PIPELINE ? (i0 >> 1) & 1 : 0;
          if (GCRNN_SEQ32_GATE_W_PIPELINE) { if (i0 + 2 < STILES) w8_load(wset ^ 1, i0 + 2); }
          else w8_load(0, i0);
          const auto& w8 = w8s[wset];
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int i = i0 + t2;
            const int node = swe[i] >> 16;
            u32x4_t p{0u, 0u, 0u, 0u};
            if (node < N) {
              const f32x4 a0 = acc[i][0], a1 = acc[i][1];
              const float o0 = fast_tanh(a0[0] + bs[0][0]), o1 = fast_tanh(a0[1] + bs[0][1]), o2 = fast_tanh(a0[2] + bs[0][2]), o3 = fast_tanh(a0[3] + bs[0][3]);
              const float o4 = fast_tanh(a1[0] + bs[1][0]), o5 = fast_tanh(a1[1] + bs[1][1]), o6 = fast_tanh(a1[2] + bs[1][2]), o7 = fast_tanh(a1[3] + bs[1][3]);
              // (explicit chain: with -ffp-contract=fast the association of a*b + c*d + .. would be the compiler's choice per instantiation)
              part = __builtin_fmaf(o3, w8[t2][0].w, __builtin_fmaf(o2, w8[t2][0].z, __builtin_fmaf(o1, w8[t2][0].y, __builtin_fmaf(o0, w8[t2][0].x, part))));
              part = __builtin_fmaf(o7, w8[t2][1].w, __builtin_fmaf(o6, w8[t2][1].z, __builtin_fmaf(o5, w8[t2][1].y, __builtin_fmaf(o4, w8[t2][1].x, part))));
              p[0] = pack2bf(o0, o1); p[1] = pack2bf(o2, o3); p[2] = pack2bf(o4, o5); p[3] = pack2bf(o6, o7);
            }
            if (cso) __builtin_amdgcn_raw_buffer_store_b128(p, rsrc_c, node * (F * 2) + (cg * 32 + q * 8) * 2, b * (NP * F * 2), 0);
          }
        }
        GCRNN_STAMP32(1 + chunk * 24 + 18);      // (diagnostic builds: end of the tile loop)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if (lane == 0) a.go[(int64_t)b * (NCH * SWAVES) + chunk * SWAVES + wave] = part;
        }
      } else if constexpr (MODE == 3) {
        // filter-output pass: A(S) x_t + b as it is (bf16), [items][NP][F]
        const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(a.out0, 0, B * (NP * F * 2), 0x00020000);
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          const int node = swe[i] >> 16;
          u32x4_t p{0u, 0u, 0u, 0u};
          if (node < N) {
            const f32x4 a0 = acc[i][0], a1 = acc[i][1];
            p[0] = pack2bf(a0[0] + bs[0][0], a0[1] + bs[0][1]); p[1] = pack2bf(a0[2] + bs[0][2], a0[3] + bs[0][3]);
            p[2] = pack2bf(a1[0] + bs[1][0], a1[1] + bs[1][1]); p[3] = pack2bf(a1[2] + bs[1][2], a1[3] + bs[1][3]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(p, rsrc_c, node * (F * 2) + (chunk * 32 + q * 8) * 2, b * (NP * F * 2), 0);
        }
      } else if constexpr (MODE == 4) {
        // node-gated step: the x part comes from the all-items pass, both parts are scaled per node (and per sequence)
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          const int node = swe[i] >> 16;
          u32x4_t p{0u, 0u, 0u, 0u};
          if (node < N) {
            const float ni = gin * epn[i][0], nf = gfo * epn[i][1];
            float yv[8], rw[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              yv[2 * e] = bf2f((uint16_t)(epg[i][e] & 0xffffu)); yv[2 * e + 1] = bf2f((uint16_t)(epg[i][e] >> 16));
              rw[e] = acc[i][0][e] + bs[0][e]; rw[4 + e] = acc[i][1][e] + bs[1][e];
            }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fast_tanh(ni * yv[e] + nf * rw[e]);
            p[0] = pack2bf(o[0], o[1]); p[1] = pack2bf(o[2], o[3]); p[2] = pack2bf(o[4], o[5]); p[3] = pack2bf(o[6], o[7]);
          }
          pkd[i] = p;
          __builtin_amdgcn_raw_buffer_store_b128(p, rsrc_o, node * (F * 2) + (chunk * 32 + q * 8) * 2, b * (NP * F * 2), 0);
        }
      } else if constexpr (MODE == 2) {
        // BPTT data step: the hops applied sum_k (S)^k (dpre_t B_k^T) = d h_{t-1} (recurrent part, scaled by the forget gate of the step it came
        // through); add the upstream gradient and go through tanh': dpre_{t-1} = (gsc acc + dH_{t-1}) (1 - h_{t-1}^2); without dH the raw state
        // gradient is stored (d h0). gpart: <h_{t-1}, acc> = <B(S) h_{t-1}, dpre_t> by the adjoint identity (the forget gate's gradient).
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          const int node = swe[i] >> 16;
          float hv[8], gv[8], rw[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hv[2 * e] = bf2f((uint16_t)(eph[i][e] & 0xffffu)); hv[2 * e + 1] = bf2f((uint16_t)(eph[i][e] >> 16));
            gv[2 * e] = bf2f((uint16_t)(epg[i][e] & 0xffffu)); gv[2 * e + 1] = bf2f((uint16_t)(epg[i][e] >> 16));
            rw[e] = acc[i][0][e]; rw[4 + e] = acc[i][1][e];
          }
          if (gpart) {
#pragma unroll
            for (int e = 0; e < 8; ++e) part = __builtin_fmaf(rw[e], hv[e], part);      // (explicit chain; rows >= N of h are zero)
          }
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[e] = rw[e] * gsc;
            if (ep_dh) o[e] = (o[e] + gv[e]) * (1.f - hv[e] * hv[e]);
          }
          u32x4_t p{0u, 0u, 0u, 0u};
          if (node < N) { p[0] = pack2bf(o[0], o[1]); p[1] = pack2bf(o[2], o[3]); p[2] = pack2bf(o[4], o[5]); p[3] = pack2bf(o[6], o[7]); }
          pkd[i] = p;
          __builtin_amdgcn_raw_buffer_store_b128(p, rsrc_o, node * (F * 2) + (chunk * 32 + q * 8) * 2, b * (NP * F * 2), 0);      // (dropped when there is no output array)
        }
        if (gpart) {
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
          if (lane == 0) gpart[(int64_t)b * (NCH * SWAVES) + chunk * SWAVES + wave] = part;
        }
      } else {
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        const int node = swe[i] >> 16;
        u32x4_t p{0u, 0u, 0u, 0u};
        if (node < N) {
          const f32x4 a0 = acc[i][0], a1 = acc[i][1];
          p[0] = pack2bf(fast_tanh(a0[0] + bs[0][0]), fast_tanh(a0[1] + bs[0][1]));
          p[1] = pack2bf(fast_tanh(a0[2] + bs[0][2]), fast_tanh(a0[3] + bs[0][3]));
          p[2] = pack2bf(fast_tanh(a1[0] + bs[1][0]), fast_tanh(a1[1] + bs[1][1]));
          p[3] = pack2bf(fast_tanh(a1[2] + bs[1][2]), fast_tanh(a1[3] + bs[1][3]));
        }
        pkd[i] = p;
        if (GCRNN_SEQ32_NT_STATE && last) __builtin_amdgcn_raw_buffer_store_b128(p, rsrc_o, node * (F * 2) + (chunk * 32 + q * 8) * 2, b * (NP * F * 2), GCRNN_SEQ32_NT_STATE);
        else __builtin_amdgcn_raw_buffer_store_b128(p, rsrc_o, node * (F * 2) + (chunk * 32 + q * 8) * 2, b * (NP * F * 2), 0);
        if (GCRNN_SEQ32_OPERAND_AT == 3 && last && more) {
          // the next operand's fragments of THIS tile (its registers died with the last tap), a tile at a time between the epilogue's own
          // work: a wave that issues all 24 requests at once sits in their issue for most of the fetch (the memory pipeline takes a request
          // when a slot frees) and does nothing else meanwhile
          const __amdgpu_buffer_rsrc_t rsrc_hn = __builtin_amdgcn_make_buffer_rsrc(hout, 0, B * (NP * F * 2), 0x00020000);
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            if (s == HS - 1) continue;
            if (s < HS)
              bfr[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_hn, node * (F * 2) + 16 * q + 64 * s, b * (NP * F * 2), 0));
            else
              bfr[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_xn, node * (G * 2) + 16 * q + 64 * (s - HS), b * (NP * G * 2), 0));
          }
        }
      }
      }
      GCRNN_STAMP32(1 + chunk * 24 + 17);
      if (!ITEMS && last) {
        // h_t's last 32 features ARE the lanes' B fragments of k-step HS-1: handed to the next step in registers
#pragma unroll
        for (int i = 0; i < STILES; ++i) bfr[i][HS - 1] = __builtin_bit_cast(bf16x8, pkd[i]);
      }
      if (GCRNN_SEQ32_OPERAND_AT == 1) request_next_operand();
      if constexpr (ITEMS) {
        // the workgroup's next item: its input was laid out by this item's pack rounds (stored, waited for and behind barriers since the
        // first half of the hops) or by the caller; its state operand is h0
        if (last && b + (int)gridDim.x < B && !GCRNN_SEQ32_NO_ITEM_PREFETCH) {
          load_first_operand(b + (int)gridDim.x);
          have_operand = true;
        }
      }
      if (aux1) {
        // user layout H[b][t][f][:] (node-contiguous rows) through a transposed LDS tile of 32-bit words [feature pair][node]: a lane's
        // four packed registers are four such words (pairs 4 q .. 4 q + 3). Nobody reads the image any more (the last hop's barrier).
        char* tst = smem;
        int swt[STILES];
        slot_words(lane, swt);
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          const int node = swt[i] >> 16;
          char* ra = tst + (4 * q) * RS2 + node * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<uint32_t*>(ra + k * RS2) = pkd[i][k];
        }
        lds_barrier();
        GCRNN_STAMP32(1 + chunk * 24 + 18);
        const int segs = N >> 3;
        uint16_t* ub = const_cast<uint16_t*>(aux1) + (int64_t)b * ubstride + (int64_t)(chunk * 32) * N;
        const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(ub, 0, 32 * N * 2, 0x00020000);
        for (int idx = tl; idx < 16 * segs; idx += STHREADS) {
          const int fp = idx / segs, sg = idx - fp * segs;
          const u32x4_t w0 = *reinterpret_cast<const u32x4_t*>(tst + fp * RS2 + sg * 32);
          const u32x4_t w1 = *reinterpret_cast<const u32x4_t*>(tst + fp * RS2 + sg * 32 + 16);
          const u32x4_t ev = {__builtin_amdgcn_perm(w0[1], w0[0], 0x05040100u), __builtin_amdgcn_perm(w0[3], w0[2], 0x05040100u),
                              __builtin_amdgcn_perm(w1[1], w1[0], 0x05040100u), __builtin_amdgcn_perm(w1[3], w1[2], 0x05040100u)};
          const u32x4_t od = {__builtin_amdgcn_perm(w0[1], w0[0], 0x07060302u), __builtin_amdgcn_perm(w0[3], w0[2], 0x07060302u),
                              __builtin_amdgcn_perm(w1[1], w1[0], 0x07060302u), __builtin_amdgcn_perm(w1[3], w1[2], 0x07060302u)};
          __builtin_amdgcn_raw_buffer_store_b128(ev, rsrc_u, ((2 * fp) * N + sg * 8) * 2, 0, GCRNN_SEQ32_NT_ROWS);
          __builtin_amdgcn_raw_buffer_store_b128(od, rsrc_u, ((2 * fp + 1) * N + sg * 8) * 2, 0, GCRNN_SEQ32_NT_ROWS);
        }
      }
      // this chunk's stores have retired (the next step reads some of them back; K = 2: the next chunk's tap 0 has landed), the tile
      // has been read: the image may be seeded again
      if (GCRNN_SEQ32_OPERAND_AT == 2) request_next_operand();
      GCRNN_STAMP32(1 + chunk * 24 + 19);
      // K = 2 only: the next chunk's tap 0 (LDS-DMA behind the last hop) has landed. Otherwise NO wait here: every LDS-DMA piece was covered by
      // its hop's wait, the stores the next step reads back (the earlier chunks' states, the pack's rows) have been behind a hop's vmcnt(0)
      // since, and the next operand's requests are waited for where the seed's MFMAs first use them (hipcc counts them: the state fragments
      // were requested first, so the seed starts on them while the input's are still landing).
      if (K == 2 && NCH > 1 && !SPLIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (GCRNN_SEQ32_END_WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      GCRNN_STAMP32(1 + chunk * 24 + 20);
      if (!SPLIT && chunk + 1 < NCH) seed();
    }  // chunks
  }  // steps
  }  // sequences
  GCRNN_STAMP32_FLUSH();
}
