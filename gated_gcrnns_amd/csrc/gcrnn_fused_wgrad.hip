// Weight-gradient kernel of the fused BPTT (shares the hop macros and the graph image of gcrnn_fused_step.h).
#include "gcrnn_fused_step.h"

// ------------------------------------------------------------------------------------------
// BPTT weight gradient of the un-gated cell:   dW_k[f'][j] = sum_{t,b,n} du_k[t,b][n][f'] * z[t,b][n][j],
//   du_0 = dpre_t,  du_k = S du_{k-1}  (adjoint hops, CSR(S)),   z = [h_{t-1} | x_t]   (adjoint of graphML.py:134-135).
// One workgroup = one (item (t,b), 16-feature chunk of dpre); wave w owns input-feature tile w of z. Every item is
// independent (no recurrence once dpre is known), so ONE launch covers all T*B items and each workgroup keeps its
// K accumulator tiles D_k [16 f' x 16 j] in registers across its items and stores them once, as ITS partial sum: the
// caller adds the partials of the workgroup slots in a fixed order (no atomics anywhere: two runs give the same bits).
//  - node index = the MFMA contraction dimension. B operand: 8 consecutive nodes of one input feature = one 16-byte
//    load from the USER layout (x[b][t][g][:], H[b][t-1][f][:] are node-contiguous), held in registers across the taps.
//    A operand: du_k transposed, a bf16 [16 f'][512 nodes] LDS image per half of the nodes, row stride 1056 B chosen so
//    that the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte bank slots (slot = 2 f' + kg mod 16).
//  - du_k is consumed by the GEMM of tap k and by the hop that produces du_{k+1}: no per-tap storage at all.
// LDS: state fp32 [1024][16] (64 KiB) | graph image 96 B x entries | transposed half image (16.5 KiB).
// ------------------------------------------------------------------------------------------
// In-kernel phase stamps (diagnostic builds only, -DGCRNN_WGRAD_STAMPS; tools/wgrad_stamps.py): thread 0 of every workgroup records s_memtime
// at the phase boundaries of its THIRD item (a global store each; the timing build is not the shipped one).
#if defined(GCRNN_WGRAD_STAMPS)
static __device__ unsigned long long gcrnn_wgrad_stamps[1024 * 32];
#define WG_STAMP(slot)                                                                                         \
  do {                                                                                                         \
    if (stamp_on && threadIdx.x == 0) {                                                                        \
      unsigned long long tv_;                                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tv_)::"memory");                              \
      gcrnn_wgrad_stamps[blockIdx.x * 32 + (slot)] = tv_;                                                      \
    }                                                                                                          \
  } while (0)
extern "C" int gcrnn_debug_read_wgrad_stamps(void* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(gcrnn_wgrad_stamps), sizeof(unsigned long long) * 1024 * 32) == hipSuccess ? 0 : 1;
}
#else
#define WG_STAMP(slot) do {} while (0)
#endif
template <class Fn, int... J>
__device__ __forceinline__ void wg_for(Fn&& f, std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
template <int N, class Fn>
__device__ __forceinline__ void wg_forn(Fn&& f) { wg_for(f, std::make_integer_sequence<int, N>{}); }
namespace { constexpr int TSTRIDE = 1056; constexpr int TBYTES = 16 * TSTRIDE; constexpr bool HT_IS_8 = (TILES == 8); }

// UNI (uniform-weight graphs, gcrnn_ell_fill_z): no weight image in LDS, so BOTH halves of the transposed du_k image fit; a tap
// is then {images of all nodes} barrier {MFMAs of the first half | re-fetched fragments of the second half in two batches, the
// second one in flight across the hop | adjoint hop on the uniform stream} barrier -- two barriers per tap instead of four.
// R1 (UNI == 2 only): rank-1-weighted graph S[m][n] = a[m] b[n] on the adjoint plan of its 0/1 pattern (graph.fused_plan_rank1(adjoint=True)):
// du_{k+1} = b (.) sum over the pattern of (a (.) du_k) -- the hop image holds a (.) du_k, the sums are scaled by b (r1a / r1b: the factors of
// this direction, [NP] fp32); the transposed image (the GEMM's operand) holds du_k itself.
// CPW (round 5; UNI == 2, z resident): 16-feature chunks of dpre per workgroup VISIT of an item. With 2, the item's operand z -- 256 KB through a
// CU's ~29 B per clock from L2: a quarter of a (item, chunk) -- is loaded once for two chunks; the accumulators of the chunk that is not being
// worked on are parked in LDS (K f32x4 per thread, one 16-byte access each way per item: the order of the two chunks alternates from item
// to item, so they change places once per item).
template <int K, int HS, int XS, int UNI, bool R1 = false, int CPW = 1>
__global__ __launch_bounds__(512, 2) void fused_wgrad_kernel(
    const uint16_t* __restrict__ dpre,       // [T][B][NP][F] bf16 sequence-major
    const uint16_t* __restrict__ Xuser,      // [B][T][G][N] bf16
    const uint16_t* __restrict__ Huser,      // [B][T][F][N] bf16 (forward output)
    const uint16_t* __restrict__ h0user,     // [B][F][N]   bf16
    float* __restrict__ dW,                  // [slots][F][K][F+G] fp32 partial sums, slot = this workgroup's item slot (plain stores)
    float* __restrict__ dbsum,               // [slots][F] fp32 partials of the bias gradient: sum_{t,b} (gi + gf) sum_n dpre, 2 sum dpre without gates (or null)
    const int32_t* __restrict__ tile_nodes, const int32_t* __restrict__ tile_off,
    const float4* __restrict__ ell_val4, const uint2* __restrict__ ell_col4,
    const float* __restrict__ gi,            // [T][B] input-filter gates of the time-gated cell, or null
    const float* __restrict__ gf,            // [T][B] state-filter gates, or null
    int h_is_h0,                             // the state operand of EVERY item is h0 (gate sub-cells, graphML.py:2362, 2370)
    const int32_t* __restrict__ hzero,       // with h_is_h0 (or null): hzero[0] != 0 = h0 is all zeros: the state-feature waves skip their loads and MFMAs
    int entries, int B, int Tn, int N, float uni_w,
    const float* __restrict__ r1a, const float* __restrict__ r1b) {
  constexpr int F = 32 * HS, G = 32 * XS, C = F + G, NCH = F / FC, JT = C / 16;
  constexpr int NWG = NCH / CPW;      // workgroups per item
  constexpr int STB = (CPW == 2) ? NP * FC * 2 : NP * FC * 4;      // the hop image (CPW == 2: bf16 rows only, the fp32 size was half unused)
  static_assert(JT <= WAVES, "one input-feature tile per wave");
  static_assert(!R1 || UNI == 2, "rank-1 graphs: the bf16-image variant");
  static_assert(CPW == 1 || (CPW == 2 && UNI == 2 && NCH % 2 == 0), "chunk pairs: the bf16-image variant, an even number of chunks");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* state = reinterpret_cast<float*>(smem);
  float4* lval4 = reinterpret_cast<float4*>(smem + STB);
  uint2* lcol4 = reinterpret_cast<uint2*>(lval4 + (UNI ? 0 : entries * 4));
  char* tbuf = reinterpret_cast<char*>(lcol4 + entries * 4) + (UNI == 2 ? GCRNN_HOP_COLUMN_PAD : 0);      // (UNI == 2: zeros behind the column image, the summing stream does not clamp its column pointer)
  float* lbias = reinterpret_cast<float*>(tbuf + (UNI ? 2 : 1) * TBYTES);      // [CPW][WAVES][16] bias-gradient partial sums, one row per wave (no LDS atomics)
  f32x4* park = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lbias) + CPW * WAVES * FC * 4 + WAVES * 256);      // CPW == 2: [K][512] the other chunk's accumulators (behind the prefetch scratch)

  const int L = blockIdx.x;
  const int grp = L / (8 * NWG), rem = L - grp * (8 * NWG);
  const int cw = rem >> 3, it0 = grp * 8 + (rem & 7);
  const int seq_slots = (gridDim.x / (8 * NWG)) * 8;
  const int items = B * Tn;
  if (it0 >= items) return;

  constexpr int HT = TILES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  {
    const int n = (entries >> 2) * 16;
    for (int i = tid; i < n; i += 512) { if (!UNI) lval4[i] = ell_val4[i]; lcol4[i] = ell_col4[i]; }
    if (UNI == 2 && tid < GCRNN_HOP_COLUMN_PAD / 4) reinterpret_cast<uint32_t*>(lcol4 + entries * 4)[tid] = 0u;
  }
  int tbeg[TILES], tend[TILES], woff[TILES];
#pragma unroll
  for (int i = 0; i < TILES; ++i) {
    tbeg[i] = tile_off[wave * TILES + i];
    tend[i] = tile_off[wave * TILES + i + 1];
    const int nd = tile_nodes[(wave * TILES + i) * 16 + r];
    woff[i] = nd ^ (UNI == 2 ? (((q >> 1) << 4) | ((q & 1) << 3)) : (q << 4));      // slot = node << 16 | row << 6 | swz << 4  (UNI == 2: bf16 image, row16 << 5 | hswz << 4)
  }
  const int qoff = q * 16;
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const uint32_t qx = (uint32_t)qoff;
  const uint32_t lds_val = lds0 + STB;
  const uint32_t lds_col = lds_val + (UNI ? 0 : entries * 64);

  f32x4 accD[K];
#pragma unroll
  for (int k = 0; k < K; ++k) accD[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < CPW * WAVES * FC) lbias[tid] = 0.f;       // row w: wave w's share of the bias gradient (its lanes r == 0 own quad q's 4 features)
  if constexpr (CPW == 2) {
#pragma unroll
    for (int k = 0; k < K; ++k) park[k * 512 + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int held = 0;               // CPW == 2: which chunk of the pair (0 / 1) has its accumulators in registers; the item starts with it
  float gpark = 1.f;          // ... and the gate unit of the parked ones (see gprev below)
  const bool has_tile = wave < JT;
  const bool is_x = wave >= F / 16;                 // wave-uniform: tiles 0..F/16-1 are h features, the rest x features
  const int jrow = (is_x ? wave * 16 - F : wave * 16) + r;      // this lane's row (feature) inside its source block
  // Time-gated cell: item (t, b) enters dW_A with weight gi_t[b] and dW_B with gf_t[b]. A wave owns features of ONE of the
  // two filters, so its accumulators are kept in units of the current item's gate: accD_true = gprev * accD. Re-basing
  // costs K*4 multiplies per item and no registers; items whose gate underflowed contribute nothing.
  const float* gw_ = is_x ? gi : gf;
  float gprev = 1.f;
  const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dpre), 0, Tn * B * (NP * F * 2) > 0 ? Tn * B * (NP * F * 2) : 0x7fffffff, 0x00020000);
  __syncthreads();

#ifndef GCRNN_WGRAD_PREFETCH
#define GCRNN_WGRAD_PREFETCH 1      // 0: off (A/B)
#endif
  // L2 prefetch of this workgroup slot's NEXT item (round 5): its x, h and dpre blocks, one dword per 128-byte line by LDS-DMA into a scratch row
  // (no register, nobody waits for it), issued behind tap 0's GEMM -- the item's own loads have landed by then and the taps issue no global
  // traffic. The NWG workgroups of an item sit on one XCD (consecutive-by-8 workgroup ids) and share the lines between them. A CU pulls
  // ~10-14 B per clock from HBM / Infinity Cache and ~29 from L2 (MI355X_MICROARCH.md): the item-start loads, 290 KB, were 40 % of an item
  // (profiles/r05_wgrad_stamps_before.txt).
  const uint32_t pf_lds = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(lbias + CPW * WAVES * FC) + (uint32_t)wave * 256u);
  auto prefetch_next_item = [&](int itn) {
    if (!GCRNN_WGRAD_PREFETCH || itn >= items) return;
    const int tn = itn / B, bn = itn - tn * B;
    const uint16_t* hsrc = (tn > 0 && !h_is_h0) ? Huser + ((int64_t)bn * Tn + (tn - 1)) * F * N : h0user + (int64_t)bn * F * N;
    const uint16_t* srcs[3] = {Xuser + ((int64_t)bn * Tn + tn) * G * N, hsrc, dpre + (int64_t)(tn * B + bn) * NP * F};
    const uint32_t bytes[3] = {(uint32_t)(G * N * 2), (uint32_t)(F * N * 2), (uint32_t)(NP * F * 2)};
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
      const uint32_t lines = (bytes[blk] + 127u) / 128u, share = (lines + NWG - 1) / NWG;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(srcs[blk]), 0, (int)bytes[blk], 0x00020000);
      for (uint32_t l0 = 0; l0 < share; l0 += 512) {      // (wave-uniform trip count; lines past the block's end are dropped by the bounds check)
        const uint32_t line = (uint32_t)cw * share + l0 + (uint32_t)tid;
        const uint32_t voff = (l0 + (uint32_t)tid < share) ? line * 128u : 0xfffffff0u;
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dword %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(pf_lds) : "memory");
      }
    }
  };
#ifndef GCRNN_WGRAD_Z_RESIDENT
#define GCRNN_WGRAD_Z_RESIDENT 1      // UNI == 2: all 32 fragments of z stay in registers for the item (the summing stream's window is 28 registers, not 60)
#endif
#ifndef GCRNN_WGRAD_PIPELINE
#define GCRNN_WGRAD_PIPELINE 0        // 1: the NEXT item's operands are requested inside the current item's last tap (A/B; measured: no gain)
#endif
  constexpr bool ZRES = (UNI == 2) && GCRNN_WGRAD_Z_RESIDENT;
  // PIPE (round 5 experiment, off): z resident, the NEXT item's operands requested inside the current item's last tap -- du_0 of its first chunk
  // behind that tap's images (du_k's registers are dead: no hop follows), then one fragment of z behind each MFMA that has just read the old one.
  // Measured (profiles/r05_wgrad_stamps_pipeline.txt): the item start shrinks from 180 to 60 units and the last GEMM grows by the same 120 -- a
  // CU keeps ~8 KB of requests in flight, so 290 KB take their ~10 k cycles wherever they are issued, and the only place the registers allow
  // is one GEMM long. 862 vs 857 units per item pair.
  constexpr bool PIPE = ZRES && GCRNN_WGRAD_PIPELINE;
  bf16x8 bfr[ZRES ? 32 : 16];
  u32x2 d2n[TILES];                                          // PIPE: du_0 of the upcoming item's first chunk, in flight across the item boundary
  const int vo = (jrow * N + 8 * q) * 2;
  // item -> this wave's z rows (a zero-length buffer when the wave has nothing to add: the loads cost nothing) and its gate units
  struct ItemSrc { __amdgpu_buffer_rsrc_t rz, rd; float gcur, gbias; bool live; int soff_d; };
  auto item_src = [&](int itx) {
    ItemSrc o;
    const bool valid = itx < items;
    const int tx = valid ? itx / B : 0, bx = valid ? itx - tx * B : 0;
    o.gcur = 1.f;
    o.gbias = 2.f;                                         // the one bias enters both filters
    if (gw_) { o.gcur = gw_[tx * B + bx]; o.gbias = gi[tx * B + bx] + gf[tx * B + bx]; }
    o.live = has_tile && (!gw_ || o.gcur > 1e-12f) && !(h_is_h0 && !is_x && hzero && hzero[0] != 0);      // wave-uniform
    const uint16_t* zsrc;
    int zrows;
    if (is_x) { zsrc = Xuser + ((int64_t)bx * Tn + tx) * G * N; zrows = G; }
    else if (tx > 0 && !h_is_h0) { zsrc = Huser + ((int64_t)bx * Tn + (tx - 1)) * F * N; zrows = F; }
    else { zsrc = h0user + (int64_t)bx * F * N; zrows = F; }
    o.rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(zsrc), 0, (o.live && valid) ? zrows * N * 2 : 0, 0x00020000);
    o.soff_d = ((tx * B + bx) * NP) * (F * 2);
    o.rd = valid ? rsrc_d : __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dpre), 0, 0, 0x00020000);      // (no such item: a zero-length buffer)
    return o;
  };
  auto request_dpre = [&](u32x2* dst, const __amdgpu_buffer_rsrc_t& rd, int soff, int chunkx) {
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      dst[i] = __builtin_amdgcn_raw_buffer_load_b64(rd, (wv >> 16) * (F * 2) + (chunkx * FC + q * 4) * 2, soff, 0);
    }
  };
  if constexpr (PIPE) {
    const ItemSrc s0 = item_src(it0);
    request_dpre(d2n, s0.rd, s0.soff_d, cw * CPW);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2)
      bfr[s2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(s0.rz, vo + 64 * s2, 0, 0));
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int it = it0; it < items; it += seq_slots) {
    const int t = it / B, b = it - t * B;
    [[maybe_unused]] const bool stamp_item = (it == it0 + 2 * seq_slots) && blockIdx.x < 1024;
    { [[maybe_unused]] const bool stamp_on = stamp_item; WG_STAMP(0); }
    // ---- B operand: this wave's 16 input features x 1024 nodes, straight from the user layout --------------------
    // The fragments of nodes 0..511 stay in registers across the taps; those of nodes 512..1023 are re-fetched per tap
    // (L2-resident after the first tap) through registers that the hop pipeline has just released -- the kernel must
    // stay spill-free: a spilled destination of an in-flight asm ds_read would be saved before its data lands.
    const ItemSrc src = item_src(it);
    const float gcur = gw_ ? src.gcur : gprev, gbias = src.gbias;
    const bool live = src.live;
    const __amdgpu_buffer_rsrc_t rsrc_z = src.rz;
    // ---- du_0 = dpre chunk of this item: requested FIRST (round 5) -- memory returns in order, and tap 0's images and hop need du_0 only,
    // so the 32 fragments of z (8 x the bytes) have until tap 0's GEMM to land instead of standing in front of everything
    f32x4 cur[TILES];
    const int soff_d = src.soff_d;
    // the item's chunks (compile-time index: the two visits differ in what they request, and shared code would make the compiler's waits
    // for du_0 cover the z loads as well)
    wg_forn<CPW>([&](auto ccc) {
    constexpr int cc = decltype(ccc)::value;
    [[maybe_unused]] const bool stamp_on = stamp_item && cc == 0;
    if constexpr (cc == 1) {
      // the pair's other chunk: its accumulators (and their gate unit) change places with the parked ones
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const f32x4 o = park[k * 512 + tid];
        park[k * 512 + tid] = accD[k];
        accD[k] = o;
      }
      const float go = gpark;
      gpark = gprev;
      gprev = go;
      held ^= 1;
    }
    const int chunk = cw * CPW + (CPW == 2 ? held : 0);
    float* lbias_c = lbias + (CPW == 2 ? held : 0) * (WAVES * FC);
    if (live && gcur != gprev) {
      const float rb = gprev / gcur;
#pragma unroll
      for (int k = 0; k < K; ++k) accD[k] *= rb;
      gprev = gcur;
    }
    if constexpr (!ZRES) {      // (the variants that re-fetch half of z per tap sit at the register budget: their resident half first, as in rounds 1-4)
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2)
        bfr[s2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_z, vo + 64 * s2, 0, 0));
    }
    u32x2 d2r[TILES];
    if constexpr (PIPE && cc == 0) {
#pragma unroll
      for (int i = 0; i < TILES; ++i) d2r[i] = d2n[i];      // requested inside the previous item's last tap (or ahead of the first item)
    } else {
      request_dpre(d2r, src.rd, soff_d, chunk);
    }
    if constexpr (ZRES && !PIPE) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (cc == 0) {
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2)
          bfr[s2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_z, vo + 64 * s2, 0, 0));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < TILES; ++i)
      cur[i] = f32x4{bf2f((uint16_t)(d2r[i][0] & 0xffffu)), bf2f((uint16_t)(d2r[i][0] >> 16)),
                     bf2f((uint16_t)(d2r[i][1] & 0xffffu)), bf2f((uint16_t)(d2r[i][1] >> 16))};
    if (dbsum) {                                        // sum of dpre over this item's nodes (padded rows are zero)
      f32x4 bacc = cur[0];
#pragma unroll
      for (int i = 1; i < TILES; ++i) bacc += cur[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = bacc[c];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);     // over the 16 slots r of this quad
        if (r == 0) lbias_c[wave * FC + q * 4 + c] += v * gbias;                  // one owner lane per address, items in program order: deterministic
      }
    }
    WG_STAMP(1);
    if constexpr (UNI != 0) {
    // The GEMM's image of du_k (round 5): bf16 in NODE order, piece-major -- piece p (features 4 p .. 4 p + 3 of every node, 8 bytes per node) at
    // p * 8 NP + CP[p] + 8 node: a lane's four values are one 8-byte ds_write_b64 -- read back TRANSPOSED by the hardware: ds_read_b64_tr_b16
    // hands lane (f' = lane & 15, g = lane >> 4) the values of feature f' at four consecutive nodes, two such reads are the A fragment (8 nodes
    // of ONE feature) of v_mfma_f32_16x16x32_bf16. (Rounds 3-4: 32-bit words [feature pair][node], two ds_write_b32 per tile, two 16-byte reads
    // + 4 v_perm_b32 per fragment, half of every read discarded.) The pads CP = {0, 32, 128, 160} bytes put the 32 addresses of a 32-lane half of
    // a transposed read (two blocks of 4 nodes x 4 pieces, 8 nodes apart) on 32 different bank pairs; a write's 16 lanes (one piece of 16
    // arbitrary nodes) spread over 16 bank pairs by node -- with whole 32-byte rows per node they shared four (4-way by construction).
    constexpr int PSTRIDE = NP * 8;
    static_assert(4 * PSTRIDE + 160 + 64 <= 2 * TBYTES, "the node-order image fits the transposed-image allocation");
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const uint32_t tb0 = (uint32_t)reinterpret_cast<uintptr_t>(tbuf);
    auto piece_base = [](int pp) { return (uint32_t)(pp * PSTRIDE + ((pp >> 1) * 128 + (pp & 1) * 32)); };
    // lane 4 q' + p of group g: node 8 g + 4 i + q' of the fragment's 32, piece p; read i and fragment s are immediate offsets (32 i + 256 s)
    const lds_s16x4 tr_b = reinterpret_cast<lds_s16x4>(tb0 + piece_base(lane & 3) + (uint32_t)(8 * (8 * q + ((lane >> 2) & 3))));
    auto afrag = [&](int s) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(tr_b + 32 * s), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(tr_b + 32 * s + 4);      // (+ 256 s, + 32 bytes)
      return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const uint32_t nat_q = piece_base(q);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      // first batch of the second half's fragments: requested before the images are written, consumed after the first half
      bf16x8 bl0[ZRES ? 1 : 8];
      if constexpr (!ZRES) {
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2)
          bl0[s2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_z, vo + 64 * (16 + s2), 0, 0));
      }
      // S1: du_k -> LDS state rows (for the next hop) and the transposed word images of both node halves
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        if (k < K - 1) {
          if constexpr (R1) state_put<UNI == 2>(state, wv, cur[i] * r1a[wv >> 16]);
          else state_put<UNI == 2>(state, wv, cur[i]);
        }
        const int node = wv >> 16;
        typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int u32x2w;
        *reinterpret_cast<u32x2w*>(tbuf + (nat_q + 8 * node)) = u32x2w{pack2bf(cur[i][0], cur[i][1]), pack2bf(cur[i][2], cur[i][3])};
      }
      lds_barrier();      // LDS hand-off only: loads in flight stay in flight (gcrnn_fused_step.h)
      WG_STAMP(2 + 4 * k);
      if constexpr (ZRES) {
      // (round 5) the item's first tap runs its hop FIRST -- the hop needs the hop image only, the GEMM the item's z, still landing -- every other
      // tap its GEMM first: du_k's registers are dead then (the hop refills them), which is what lets eight A fragments be in flight
      auto hop_phase = [&]() {
  if (k < K - 1) {
          LGKM_WAIT(0);
#define GCRNN_WG_INIT(i) f32x4{0.f, 0.f, 0.f, 0.f}
#define GCRNN_WG_STORE(i, a) cur[i] = a
          if constexpr (UNI == 2) {
            // the summing stream (tile exits cost nothing, register window 28 instead of 42): du_{k+1} = w * (sum of the gathered rows)
            GCRNN_HOP_ASM_UNI16_SUMS_STREAM(cur);
            if constexpr (R1) {
#pragma unroll
              for (int i = 0; i < TILES; ++i) {
                int wv = woff[i];
                asm volatile("" : "+v"(wv));
                cur[i] *= r1b[wv >> 16];
              }
            } else {
#pragma unroll
              for (int i = 0; i < TILES; ++i) cur[i] *= uni_w;
            }
          }
          else GCRNN_HOP_ASM_UNI_STREAM(GCRNN_WG_INIT, GCRNN_WG_STORE);
#undef GCRNN_WG_INIT
#undef GCRNN_WG_STORE
        }
      };
      // the item's last GEMM also requests the NEXT item's operands (PIPE). It then runs whether the wave is live or not (a dead wave's z is zeros:
      // its MFMAs add nothing) -- one straight-line block in which every fragment register is read by its MFMA and re-requested right behind it.
      auto gemm_phase = [&]() {
        if constexpr (PIPE && cc == CPW - 1) {
          if (k == K - 1) {      // (a constant once the taps are unrolled)
            const ItemSrc nx = item_src(it + seq_slots);
            request_dpre(d2n, nx.rd, nx.soff_d, cw * CPW + (CPW == 2 ? held : 0));      // (the pair's order alternates: the next item starts with the chunk that is in registers now)
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s8 = 0; s8 < 32; s8 += 8) {
              bf16x8 a8[8];
#pragma unroll
              for (int p = 0; p < 8; ++p) a8[p] = afrag(s8 + p);
#pragma unroll
              for (int p = 0; p < 8; ++p) {
                if (p & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[p], bfr[(ZRES ? s8 + p : 0)], acc2, 0, 0, 0);
                else accD[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[p], bfr[(ZRES ? s8 + p : 0)], accD[k], 0, 0, 0);
                bfr[(ZRES ? s8 + p : 0)] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(nx.rz, vo + 64 * (s8 + p), 0, 0));
              }
            }
            accD[k] += acc2;
            return;
          }
        }
        if (live) {
          f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};      // two accumulation chains: 32 dependent MFMAs on one tile were the GEMM's whole duration
#pragma unroll
          for (int s8 = 0; s8 < 32; s8 += 8) {
            bf16x8 a8[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) a8[p] = afrag(s8 + p);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
              if (p & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[p], bfr[(ZRES ? s8 + p : 0)], acc2, 0, 0, 0);
              else accD[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[p], bfr[(ZRES ? s8 + p : 0)], accD[k], 0, 0, 0);
            }
          }
          accD[k] += acc2;
        }
      };
      const bool hop_first = (k == 0 && cc == 0);      // (a constant once the taps are unrolled)
      if (hop_first) hop_phase(); else gemm_phase();
      WG_STAMP(3 + 4 * k);
      if (hop_first) gemm_phase(); else hop_phase();
      if (k == 0 && cc == 0) {
        // every wave, live or not, has its z (the compiler's own waits sit inside `if (live)`: without this one it re-waits in every later
        // tap -- and those waits would then also cover the prefetches below, which it does not know about)
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
        prefetch_next_item(it + seq_slots);
      }
      WG_STAMP(4 + 4 * k);
      } else {
      // S2: D_k += du_k^T z over nodes 0..511 (register-resident fragments), then the first re-fetched batch (nodes 512..767)
      if (live) {
#pragma unroll
        for (int s4 = 0; s4 < 16; s4 += 4) {
          bf16x8 a4[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) a4[p] = afrag(s4 + p);
#pragma unroll
          for (int p = 0; p < 4; ++p) accD[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a4[p], bfr[s4 + p], accD[k], 0, 0, 0);
        }
#pragma unroll
        for (int s4 = 0; s4 < 8; s4 += 4) {
          bf16x8 a4[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) a4[p] = afrag(16 + s4 + p);
#pragma unroll
          for (int p = 0; p < 4; ++p) accD[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a4[p], ZRES ? bfr[(ZRES ? 16 : 0) + s4 + p] : bl0[ZRES ? 0 : s4 + p], accD[k], 0, 0, 0);
        }
      }
      WG_STAMP(3 + 4 * k);
      // second batch (nodes 768..1023): in flight across the hop
      bf16x8 bl1[ZRES ? 1 : 8];
      if constexpr (!ZRES) {
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2)
          bl1[s2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_z, vo + 64 * (24 + s2), 0, 0));
      }
      if (k < K - 1) {
        LGKM_WAIT(0);
#define GCRNN_WG_INIT(i) f32x4{0.f, 0.f, 0.f, 0.f}
#define GCRNN_WG_STORE(i, a) cur[i] = a
        if constexpr (UNI == 2) {
          // the summing stream (tile exits cost nothing, register window 28 instead of 42): du_{k+1} = w * (sum of the gathered rows)
          GCRNN_HOP_ASM_UNI16_SUMS_STREAM(cur);
          if constexpr (R1) {
#pragma unroll
            for (int i = 0; i < TILES; ++i) {
              int wv = woff[i];
              asm volatile("" : "+v"(wv));
              cur[i] *= r1b[wv >> 16];
            }
          } else {
#pragma unroll
            for (int i = 0; i < TILES; ++i) cur[i] *= uni_w;
          }
        }
        else GCRNN_HOP_ASM_UNI_STREAM(GCRNN_WG_INIT, GCRNN_WG_STORE);
#undef GCRNN_WG_INIT
#undef GCRNN_WG_STORE
      }
      WG_STAMP(4 + 4 * k);
      if (live) {
#pragma unroll
        for (int s4 = 0; s4 < 8; s4 += 4) {
          bf16x8 a4[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) a4[p] = afrag(24 + s4 + p);
#pragma unroll
          for (int p = 0; p < 4; ++p) accD[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a4[p], ZRES ? bfr[(ZRES ? 24 : 0) + s4 + p] : bl1[ZRES ? 0 : s4 + p], accD[k], 0, 0, 0);
        }
      }
      }
      lds_barrier();      // LDS hand-off only: loads in flight stay in flight (gcrnn_fused_step.h)
      WG_STAMP(5 + 4 * k);
    }
    } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      // S1: du_k -> LDS state rows (for the next hop) and the transposed bf16 image of nodes 0..511
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        if (k < K - 1) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(state) + (wv & 0xffff)) = cur[i];
        const int node = wv >> 16;
        if (node < 512) {
          const int ro = (q & 1) ? 2 : 0;                    // odd quads: rows in the order 2, 3, 0, 1 (see the uniform variant)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float v = (q & 1) ? cur[i][c ^ 2] : cur[i][c];
            *reinterpret_cast<uint16_t*>(tbuf + (q * 4 + (c ^ ro)) * TSTRIDE + node * 2) = f2bf(v);
          }
        }
      }
      lds_barrier();      // LDS hand-off only: loads in flight stay in flight (gcrnn_fused_step.h)
      // S2: D_k += du_k^T z over nodes 0..511
      if (live) {
#pragma unroll
        for (int s4 = 0; s4 < 16; s4 += 4) {            // 4 A fragments in flight per batch: LDS latency overlaps the MFMAs
          bf16x8 a4[4];
#pragma unroll
          for (int p = 0; p < 4; ++p)
            a4[p] = *reinterpret_cast<const bf16x8*>(tbuf + r * TSTRIDE + (32 * (s4 + p) + 8 * q) * 2);
#pragma unroll
          for (int p = 0; p < 4; ++p) accD[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a4[p], bfr[s4 + p], accD[k], 0, 0, 0);
        }
      }
      lds_barrier();      // LDS hand-off only: loads in flight stay in flight (gcrnn_fused_step.h)
      // S3: transposed image of nodes 512..1023
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        const int node = wv >> 16;
        if (node >= 512) {
          const int ro = (q & 1) ? 2 : 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float v = (q & 1) ? cur[i][c ^ 2] : cur[i][c];
            *reinterpret_cast<uint16_t*>(tbuf + (q * 4 + (c ^ ro)) * TSTRIDE + (node - 512) * 2) = f2bf(v);
          }
        }
      }
      lds_barrier();      // LDS hand-off only: loads in flight stay in flight (gcrnn_fused_step.h)
      // S4: du_{k+1} = S du_k (reads `state`; the transposed image of du_k stays valid);  S5: second half of the contraction
#ifdef GCRNN_WGRAD_ABLATE_HOP      // profiling builds (tools/wgrad_ablate.sh): results are wrong by construction
      if (false) {
#else
      if (k < K - 1) {
#endif
#ifdef GCRNN_WGRAD_PLAIN_HOP
#pragma unroll
        for (int i = 0; i < TILES; ++i) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          for (int gq = tbeg[i] >> 2; gq < (tend[i] >> 2); ++gq) {
            const uint2 c4 = lcol4[gq * 16 + r];
            const float4 v4 = lval4[gq * 16 + r];
            const char* sb = reinterpret_cast<const char*>(state);
            acc += v4.x * *reinterpret_cast<const f32x4*>(sb + ((c4.x & 0xffffu) ^ qx));
            acc += v4.y * *reinterpret_cast<const f32x4*>(sb + ((c4.x >> 16) ^ qx));
            acc += v4.z * *reinterpret_cast<const f32x4*>(sb + ((c4.y & 0xffffu) ^ qx));
            acc += v4.w * *reinterpret_cast<const f32x4*>(sb + ((c4.y >> 16) ^ qx));
          }
          cur[i] = acc;
        }
#else
        LGKM_WAIT(0);
#define GCRNN_WG_INIT(i) f32x4{0.f, 0.f, 0.f, 0.f}
#define GCRNN_WG_STORE(i, a) cur[i] = a
#if GCRNN_HOP_ASM
        GCRNN_HOP_ASM_STREAM(GCRNN_WG_INIT, GCRNN_WG_STORE);      // one asm block: no compiler copy can land between a read and its wait
#else
        GCRNN_HOP_TILED(GCRNN_WG_INIT, GCRNN_WG_STORE);
#endif
#undef GCRNN_WG_INIT
#undef GCRNN_WG_STORE
#endif
      }
      if (live) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          bf16x8 bl[8];
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2)
#ifdef GCRNN_WGRAD_ABLATE_REFETCH
            bl[s2] = bfr[8 * h2 + s2];
#else
            bl[s2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_z, vo + 64 * (16 + 8 * h2 + s2), 0, 0));
#endif
          bf16x8 al[8];
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2)
            al[s2] = *reinterpret_cast<const bf16x8*>(tbuf + r * TSTRIDE + (32 * (8 * h2 + s2) + 8 * q) * 2);
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) accD[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[s2], bl[s2], accD[k], 0, 0, 0);
        }
      }
      lds_barrier();      // LDS hand-off only: loads in flight stay in flight (gcrnn_fused_step.h)
    }
    }
    });      // chunks of the item
    { [[maybe_unused]] const bool stamp_on = stamp_item; WG_STAMP(25); }
  }
  // ---- flush: D_k[f' = 4q + c][j = 16 wave + r] -> this slot's partial dW[it0][chunk*16 + f'][k][j] (every element of the
  // slot's [F][K][C] block is written by exactly one lane of one workgroup) ----------------------------------------------
  if (has_tile) {
    float* dWs = dW + (int64_t)it0 * (F * K * C);
#pragma unroll
    for (int hc = 0; hc < CPW; ++hc) {
      const int chunk = cw * CPW + (CPW == 2 ? (held ^ hc) : 0);      // first the chunk in registers, then the parked one
      const float gs = hc ? gpark : gprev;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const f32x4 a = hc ? park[k * 512 + tid] : accD[k];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          dWs[((int64_t)(chunk * FC + q * 4 + c) * K + k) * C + wave * 16 + r] = a[c] * gs;
      }
    }
  }
  if (dbsum) {
    __syncthreads();
    if (tid < CPW * FC) {
      const int hc = tid / FC, f = tid - hc * FC;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += lbias[hc * (WAVES * FC) + w * FC + f];                // fixed order over the waves
      dbsum[(int64_t)it0 * F + (cw * CPW + hc) * FC + f] = v;
    }
  }
}

// Number of partial-sum slots gcrnn_fused_backward_weight_bf16 writes for B*T items and F state features.
extern "C" int64_t gcrnn_fused_wgrad_slots(int64_t items, int64_t F) {
  const int64_t NCH = F / FC;
  int64_t slots = cdiv(items, 8) * 8;
  const int64_t max_slots = (256 / NCH) / 8 * 8 > 0 ? (256 / NCH) / 8 * 8 : 8;
  return slots > max_slots ? max_slots : slots;
}

// LDS bytes of the bf16 weight-gradient kernel (cpw: 16-feature chunks per workgroup visit of an item)
static size_t wgrad_lds_bytes(bool uni, bool img16, int64_t entries, int K, int cpw) {
  if (!uni) return (size_t)NP * FC * 4 + (size_t)entries * 96 + TBYTES + WAVES * FC * 4 + WAVES * 256;      // (the last 256 B per wave: where the L2 prefetches land)
  if (cpw == 2) return (size_t)NP * FC * 2 + (size_t)entries * 32 + 2 * TBYTES + 2 * WAVES * FC * 4 + WAVES * 256 + GCRNN_HOP_COLUMN_PAD + (size_t)K * 512 * 16;
  return (size_t)NP * FC * 4 + (size_t)entries * 32 + 2 * TBYTES + WAVES * FC * 4 + WAVES * 256 + (img16 ? GCRNN_HOP_COLUMN_PAD : 0);
}
// chunks per visit the launch will use: pairs on the bf16-image plans (uniform-weight and rank-1 graphs) when F has an even number of chunks and the parked accumulators fit
static int wgrad_cpw(bool uni, bool img16, int64_t entries, int64_t F, int K) {
#ifdef GCRNN_WGRAD_NO_PAIRS
  return 1;
#endif
  return (uni && img16 && (F / FC) % 2 == 0 && wgrad_lds_bytes(true, true, entries, K, 2) <= 160 * 1024) ? 2 : 1;
}
// ... and the slots of THAT launch: with chunk pairs an item has half the workgroups, so twice the slots fill the chip. img16: the graph arrays
// address a bf16 hop image (bit 1 of h_is_h0 in gcrnn_fused_backward_weight_bf16).
extern "C" int64_t gcrnn_fused_wgrad_bf16_slots(int64_t items, int64_t F, int64_t K, int64_t entries, int img16) {
  const int64_t NWG = (F / FC) / wgrad_cpw(img16 != 0, img16 != 0, entries, F, (int)K);
  int64_t slots = cdiv(items, 8) * 8;
  const int64_t max_slots = (256 / NWG) / 8 * 8 > 0 ? (256 / NWG) / 8 * 8 : 8;
  return slots > max_slots ? max_slots : slots;
}

template <int K, int HS, int XS>
static int fused_wgrad_t(const void* dpre, const void* Xuser, const void* Huser, const void* h0user, float* dW, float* dbsum,
                         const FusedGraphArgs& ga, const float* gi, const float* gf, int h_is_h0, const int32_t* hzero, int64_t B,
                         int64_t T, int64_t N, hipStream_t st, const float* r1a = nullptr, const float* r1b = nullptr) {
  constexpr int F = 32 * HS;
#if GCRNN_HOP_ASM
  const bool uni = ga.uniform_w != 0.f && HT_IS_8;
#else
  const bool uni = false;
#endif
  const int cpw = wgrad_cpw(uni, ga.img16 != 0, ga.entries, F, K);
  const size_t lds = wgrad_lds_bytes(uni, ga.img16 != 0, ga.entries, K, cpw);
  if (lds > 160 * 1024 || !ga.ell_val4 || !ga.ell_col4) return GCRNN_ERR_UNSUPPORTED;
  if (ga.img16 && !uni) return GCRNN_ERR_UNSUPPORTED;
  if (r1a && !(uni && ga.img16)) return GCRNN_ERR_UNSUPPORTED;
  auto kern = uni ? (ga.img16 ? (r1a ? fused_wgrad_kernel<K, HS, XS, 2, true> : fused_wgrad_kernel<K, HS, XS, 2>) : fused_wgrad_kernel<K, HS, XS, 1>)
                  : fused_wgrad_kernel<K, HS, XS, 0>;
  if (cpw == 2) kern = r1a ? fused_wgrad_kernel<K, HS, XS, 2, true, 2> : fused_wgrad_kernel<K, HS, XS, 2, false, 2>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  const int NWG = (F / FC) / cpw;
  const int64_t slots = gcrnn_fused_wgrad_bf16_slots(B * T, F, K, ga.entries, ga.img16);
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)(slots * NWG), 512, lds, st>>>((const uint16_t*)dpre, (const uint16_t*)Xuser, (const uint16_t*)Huser,
                                                   (const uint16_t*)h0user, dW, dbsum, ga.tile_nodes, ga.tile_off,
                                                   (const float4*)ga.ell_val4, (const uint2*)ga.ell_col4, gi, gf, h_is_h0, hzero,
                                                   (int)ga.entries, (int)B, (int)T, (int)N, ga.uniform_w, r1a, r1b);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_fused_backward_weight_bf16(const void* dpre, const void* Xuser, const void* Huser, const void* h0user,
                                                float* dW, float* dbsum, const int32_t* tile_nodes, const int32_t* tile_off,
                                                const void* ell_val4, const void* ell_col4, int64_t entries, int64_t B,
                                                int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, const float* gi,
                                                const float* gf, int h_is_h0, const int32_t* h0_zero_flag, double uniform_w,
                                                const float* rank1_a, const float* rank1_b, void* stream) {
  const int img16 = (h_is_h0 >> 1) & 1;      // bit 1: the graph arrays address a bf16 hop image (fused_plan_img16(adjoint=True))
  h_is_h0 &= 1;
  if ((gi == nullptr) != (gf == nullptr) || (rank1_a == nullptr) != (rank1_b == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (rank1_a && !img16) return GCRNN_ERR_UNSUPPORTED;      // rank-1-weighted graphs: the bf16-image plan of the pattern only
  if (!dpre || !Xuser || (!Huser && !h_is_h0) || !h0user || !dW || !tile_nodes || !tile_off || !ell_val4 || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || N % 8 || entries < 0 || entries % 4 || B * T > (1 << 24)) return GCRNN_ERR_BAD_SHAPE;
  if (T * B * (NP * F * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;      // 32-bit buffer offsets into dpre
  const FusedGraphArgs ga{tile_nodes, tile_off, nullptr, nullptr, ell_val4, ell_col4, entries, (float)uniform_w, img16};
  hipStream_t st = as_stream(stream);
#define GCRNN_WG_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) return fused_wgrad_t<KK, HH, XX>(dpre, Xuser, Huser, h0user, dW, dbsum, ga, gi, gf, h_is_h0, h_is_h0 ? h0_zero_flag : nullptr, B, T, N, st, rank1_a, rank1_b);
  GCRNN_WG_CASE(5, 2, 2)
  GCRNN_WG_CASE(4, 2, 2)
  GCRNN_WG_CASE(3, 2, 2)
  GCRNN_WG_CASE(2, 2, 2)
  GCRNN_WG_CASE(5, 1, 1)
  GCRNN_WG_CASE(4, 1, 1)
  GCRNN_WG_CASE(3, 1, 1)
  GCRNN_WG_CASE(2, 1, 1)
  GCRNN_WG_CASE(5, 2, 1)
  GCRNN_WG_CASE(4, 2, 1)
  GCRNN_WG_CASE(3, 2, 1)
  GCRNN_WG_CASE(2, 2, 1)
#undef GCRNN_WG_CASE
  return GCRNN_ERR_UNSUPPORTED;
}
