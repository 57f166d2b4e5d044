// The flagship forward with a HAND-ALLOCATED hop (round 5): the un-gated recurrence of gcrnn_fused_seq32.h on uniform-weight graphs with
// every hop -- gather stream, the hop's tap MFMAs, and in a step's last hop the NEXT step's operand requests -- as ONE generated asm block
// that owns the register file (tools/gen_hop_asm.py gen_wide32_taps).
//
// Round 4's stamps of the wide kernel (profiles/r04_seq32_stamps_*.txt, ~1,100 units of 100 cycles per step as issued): a wave ran its stream
// THEN its tap (or the other way round), so the LDS array was half idle at both ends of every hop -- a timing experiment without the hops'
// taps ran 12-13 % faster -- and the step boundary spent 170-190 units requesting the next operand, which could not start before the last tap
// had read the old one. Both need registers named inside the asm, which hipcc's operand lists (30 operands) do not give. Here the operand
// and the accumulators are 32-register TUPLES pinned by constraint ("{v[0:31]}" ..): six operands instead of 48, and every register number
// is known to the generator:
//   v[0:127]   operand [h_{t-1} | x_t]: k-step s, tile i = v[32 s + 4 i .. + 3]         (B fragments of the tap MFMAs)
//   v[128:191] accumulators: half h, tile i = v[128 + 32 h + 4 i .. + 3]               (D of the stream's v_smfmac AND of the taps)
//   v[192:253] the block's own window (gather sets, sparse A operand, weight fragments, tile node ids, addresses)
// * Taps inside the stream: one weight fragment (8 MFMAs, one per tile) at each of the stream's 8 tile exits, issued while the next tile's
//   first gathers are in flight; every wave runs the same program (no stream-first / tap-first halves any more).
// * Operand requests dribbled through the last hop: fragments are ordered k-step-major, input k-steps first; behind the exit that issues a
//   k-step's second fragment its 32 operand registers are dead, and the 8 requests of the NEXT step's k-step follow, four per exit, straight
//   into the operand registers. They fly under the rest of the stream and the epilogue; the wait sits in front of the user-layout row
//   stores (counted: only the state stores are younger) or the seed.
// * LDS-DMA through asm (p32_dma16): hipcc orders every LDS access of a wave behind ITS OWN pending LDS-DMA builtins with vmcnt(0), which in
//   the last epilogue would also wait for the operand requests; DMA it does not know about is covered by this file's own counted waits.
// * The inline pack runs TWO hops ahead (round 4: one): x_{t+1} is complete -- stored, waited for, behind a barrier -- before the step's last
//   hop starts requesting it.
// Same arithmetic as gcrnn_fused_seq32.h up to the order in which a hop's taps and sums meet in the fp32 accumulator; pinned to the fp64
// oracle directly (tests/test_wide.py). Reference: Utils/graphML.py:2351-2427 (un-gated: gi = gf = 1).
#pragma once

typedef unsigned int p32_u32x32 __attribute__((ext_vector_type(32)));
typedef float p32_f32x32 __attribute__((ext_vector_type(32)));
typedef unsigned int p32_u32x4 __attribute__((ext_vector_type(4)));

template <int I, class V>
__device__ __forceinline__ auto p32_get4(const V& v) { return __builtin_shufflevector(v, v, 4 * I, 4 * I + 1, 4 * I + 2, 4 * I + 3); }
template <int I, class V, class T4>
__device__ __forceinline__ void p32_set4(V& v, const T4& t) { v[4 * I] = t[0]; v[4 * I + 1] = t[1]; v[4 * I + 2] = t[2]; v[4 * I + 3] = t[3]; }
template <class Fn, int... J>
__device__ __forceinline__ void p32_for(Fn&& f, std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
template <int N, class Fn>
__device__ __forceinline__ void p32_forn(Fn&& f) { p32_for(f, std::make_integer_sequence<int, N>{}); }

// 16 bytes per lane global -> LDS (lane l lands at lds + 16 l), issued where hipcc does not see it (see the header)
// (a wave-uniform 64-bit base in scalar registers + a 32-bit per-lane byte offset: no 64-bit vector arithmetic, nothing derived from a
//  pointer lives in vector registers across a hop)
__device__ __forceinline__ void p32_dma16(const void* sbase, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
__device__ __forceinline__ p32_u32x4 p32_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  return p32_u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}

#define P32_TILE_ENDS                                                                                                                    \
  "s"(GCRNN_SGPR(tend[0] >> 2)), "s"(GCRNN_SGPR(tend[1] >> 2)), "s"(GCRNN_SGPR(tend[2] >> 2)), "s"(GCRNN_SGPR(tend[3] >> 2)),            \
      "s"(GCRNN_SGPR(tend[4] >> 2)), "s"(GCRNN_SGPR(tend[5] >> 2)), "s"(GCRNN_SGPR(tend[6] >> 2)), "s"(GCRNN_SGPR(tend[7] >> 2)),        \
      "s"(GCRNN_SGPR(tbeg[0] >> 2)), "s"(GCRNN_SGPR((tend[STILES - 1] >> 2) - 1))

// (the operand tuples of the k-steps a cell does not have are not operands: their registers stay the compiler's)
#define P32_OPS_IN_2 "{v[0:31]}"(op0), "{v[32:63]}"(op1)
#define P32_OPS_IN_3 P32_OPS_IN_2, "{v[64:95]}"(op2)
#define P32_OPS_IN_4 P32_OPS_IN_3, "{v[96:127]}"(op3)
#define P32_OPS_IO_2 "+{v[0:31]}"(op0), "+{v[32:63]}"(op1)
#define P32_OPS_IO_3 P32_OPS_IO_2, "+{v[64:95]}"(op2)
#define P32_OPS_IO_4 P32_OPS_IO_3, "+{v[96:127]}"(op3)

#define P32_HOP(TEXT, OPS_IN, WOFS)                                                                                                      \
  asm volatile(TEXT                                                                                                                      \
               : "+{v[128:159]}"(acc0), "+{v[160:191]}"(acc1)                                                                            \
               : P32_TILE_ENDS, "s"(lds_col), "s"(WOFS), OPS_IN                                                                          \
               : GCRNN_HOP_ASM_P32_CLOBBERS)

#define P32_HOP_LOADS(TEXT, OPS_IO, WOFS, RH, RX, SOH, SOX, SLOT)                                                                        \
  asm volatile(TEXT                                                                                                                      \
               : "+{v[128:159]}"(acc0), "+{v[160:191]}"(acc1), OPS_IO                                                                    \
               : P32_TILE_ENDS, "s"(lds_col), "s"(WOFS), "s"(RH), "s"(RX), "s"(SOH), "s"(SOX), "s"(SLOT)                                  \
               : GCRNN_HOP_ASM_P32_CLOBBERS)

#ifndef GCRNN_P32_MODE
#define GCRNN_P32_MODE 0         // 0: every wave runs the hop block with the tap MFMAs at the stream's tile exits, the next operand is requested inside the last hop
                                 // 1: round 4's split -- waves 0..3 stream then tap, waves 4..7 tap then stream (the two waves of a SIMD run complementary
                                 //    phases), the stream alone as the hand-allocated block (three sets of gathers in flight: no spill at this depth with the
                                 //    pinned tuples), the operand requested behind the last chunk's state stores
#endif
#ifndef GCRNN_P32_PRIO
#define GCRNN_P32_PRIO 0         // issue priority during the hop block: 1 = raised for waves 4..7 (the younger wave of each SIMD loses arbitration), 2 = for waves 0..3 (A/B)
#endif
#ifndef GCRNN_P32_PREFETCH
#define GCRNN_P32_PREFETCH 1     // (native layout only; 0 = off, 2 / 3: hops 1.., K-2 of the last chunk share the prefetch instead of hops 2..; profiles/r05_p32_ab.txt) x_{t+1} is pulled into L2 during the last chunk's middle hops (one dword per 128-byte line by LDS-DMA into a scratch
                                 // row: no register, nobody waits for it), so that the operand requests of the last hop are L2 hits instead of HBM misses
#endif
#ifndef GCRNN_P32_WAIT_AT
#define GCRNN_P32_WAIT_AT 0      // where the next operand's requests are waited for: 0 = in front of the user-layout row stores (or the seed), 1 = at the seed with vmcnt(0) (A/B)
#endif

// VAR: bit 0 = the launch lays out the input itself (inline pack), bit 1 = it writes the user-layout output (as gcrnn_fused_seq32.h)
template <int K, int HS, int XS, int VAR>
__global__ __launch_bounds__(STHREADS) void fused_seq32p_kernel(const Seq32Args a) {
  constexpr bool PKV = (VAR & 1) != 0, USERV = (VAR & 2) != 0;
  using M = Seq32Map<K, HS, XS>;
  constexpr int KS = HS + XS;
  constexpr int F = 32 * HS, G = 32 * XS;
  constexpr int NCH = HS;
  constexpr int PL = M::PL, WOFF = M::WOFF, WB = M::WB, COL_OFF = M::COL_OFF, RS2 = M::RS2;
  constexpr int NPCK = M::NPCK, NRND = NP / NPCK, NH = NCH * (K - 1), RPH = (NRND + NH - 1) / NH;
  constexpr int PKROWS = G;
  static_assert(STILES == 8 && K >= 2 && ((HS == 2 && XS == 2) || (HS == 2 && XS == 1) || (HS == 1 && XS == 1)), "generated hop: 8 tiles per wave; (HS, XS) in {(2,2), (2,1), (1,1)}");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int entries = a.entries, B = a.B, N = a.N;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= B) return;

  int tbeg[STILES], tend[STILES];
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
    tbeg[i] = a.tile_off[wave * STILES + i];
    tend[i] = a.tile_off[wave * STILES + i + 1];
  }
  const uint32_t slot_base = (uint32_t)(COL_OFF + entries * 32 + GCRNN_HOP_COLUMN_PAD);
  char* wtab = smem + slot_base;
  for (int idx = tid; idx < NP; idx += STHREADS) reinterpret_cast<int32_t*>(wtab)[idx] = a.tile_nodes[idx];
  // the slot words node << 16 | row16 << 5 | hswz << 4 of this lane's slot in the wave's 8 tiles: read where needed, never kept
  auto slot_words = [&](int ln, int (&w)[STILES]) {
    const uint32_t ad = slot_base + (uint32_t)((wave * STILES * 16 + (ln & 15)) * 4);
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:64\n\tds_read_b32 %2, %8 offset:128\n\tds_read_b32 %3, %8 offset:192\n\t"
                 "ds_read_b32 %4, %8 offset:256\n\tds_read_b32 %5, %8 offset:320\n\tds_read_b32 %6, %8 offset:384\n\tds_read_b32 %7, %8 offset:448\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(ad));
    const int lb = (((ln >> 4) >> 1) << 4) | (((ln >> 4) & 1) << 3);
#pragma unroll
    for (int i = 0; i < STILES; ++i) w[i] ^= lb;
  };
  {
    // once per launch (hipcc's own LDS-DMA: drained by the vmcnt(0) + __syncthreads below, nothing of it is pending inside the loops)
    const int lane = tid & 63;
    const int cbytes = entries * 32;
    const char* csrc = reinterpret_cast<const char*>(a.ell_col4);
    for (int p = wave; p * 1024 < cbytes; p += SWAVES)
      if (p * 1024 + lane * 16 < cbytes)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(csrc + p * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + COL_OFF + p * 1024), 16, 0, 0);
    const char* wsrc = reinterpret_cast<const char*>(a.wpack);
    for (int p = wave; p < WB / 1024; p += SWAVES)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + p * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + WOFF + p * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (tid < GCRNN_HOP_COLUMN_PAD / 4) reinterpret_cast<uint32_t*>(smem + COL_OFF + entries * 32)[tid] = 0u;
  float* lbias = reinterpret_cast<float*>(smem + M::BIAS_OFF);
  if (tid < NCH * 32) lbias[tid] = a.bias ? a.bias[tid] : 0.f;
  __syncthreads();

  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  if (lds0 != 0) __builtin_trap();        // gather addresses are formed from column words: the image must sit at LDS address 0
  const uint32_t lds_col = (uint32_t)COL_OFF;
  const uint32_t xtile_off = slot_base + NP * 4;
  char* xtile = smem + xtile_off;
  const uint32_t slot_wave = __builtin_amdgcn_readfirstlane(slot_base + (uint32_t)(wave * STILES * 16 * 4));
  [[maybe_unused]] const uint32_t pf_lds = xtile_off + (uint32_t)(PKV ? (32 * XS) * NPCK * 2 : 0);      // the prefetch scratch row (Seq32Map::PFS bytes at the end of the map)

  // ---- the operand of a sequence and step: every B fragment of the wave, resident for all chunks, in PINNED registers ----------------
  [[maybe_unused]] p32_u32x32 op0, op1, op2, op3;      // (the tuples of k-steps >= KS are never touched)
  p32_f32x32 acc0, acc1;
  auto opset = [&](auto sc, auto ic, const p32_u32x4& v) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, i = decltype(ic)::value;
    if constexpr (s == 0) p32_set4<i>(op0, v);
    else if constexpr (s == 1) p32_set4<i>(op1, v);
    else if constexpr (s == 2) p32_set4<i>(op2, v);
    else p32_set4<i>(op3, v);
  };
  auto opget = [&](auto sc, auto ic) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, i = decltype(ic)::value;
    if constexpr (s == 0) return p32_get4<i>(op0);
    else if constexpr (s == 1) return p32_get4<i>(op1);
    else if constexpr (s == 2) return p32_get4<i>(op2);
    else return p32_get4<i>(op3);
  };

  for (int b = (int)blockIdx.x; b < B; b += (int)gridDim.x) {
  {
    // the sequence's first operand (h0, x_0), as gcrnn_fused_seq32.h loads it
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.hfirst), 0, B * (NP * F * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.x0), 0, B * (NP * G * 2), 0x00020000);
    const int ln0 = lane_now(), qo = ln0 >> 4;
    int sw[STILES];
    slot_words(ln0, sw);
    p32_forn<KS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      p32_forn<STILES>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int w = sw[i];
        if constexpr (s < HS)
          opset(sc, ic, __builtin_bit_cast(p32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, (w >> 16) * (F * 2) + 16 * qo + 64 * s, b * (NP * F * 2), 0)));
        else
          opset(sc, ic, __builtin_bit_cast(p32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (w >> 16) * (G * 2) + 16 * qo + 64 * (s - HS), b * (NP * G * 2), 0)));
      });
    });
  }
#pragma unroll 1
  for (int step = 0; step < a.nsteps; ++step) {
    uint16_t* hout = a.out0 + (int64_t)step * a.ostride;
    const uint16_t* aux1 = (USERV && a.a1) ? (a.a1_last_only ? (step == a.nsteps - 1 ? a.a1 : nullptr) : a.a1 + (int64_t)step * a.a1stride) : nullptr;
    const bool pk_any = PKV && a.pk_src0 != nullptr;
    const int ubstride = a.ubstride;
    const int64_t pk_soff = (int64_t)b * a.pk_stride;
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(hout, 0, B * (NP * F * 2), 0x00020000);
    const bool more = step + 1 < a.nsteps;
    // the next step's operand: h_t (this step's output; its first HS - 1 chunks are re-read, the last is handed over in registers) and x_{t+1}
    const p32_u32x4 rs_hn = p32_rsrc(hout, more ? (uint32_t)(B * (NP * F * 2)) : 0u);
    const p32_u32x4 rs_xn = p32_rsrc(a.x0 + (int64_t)(step + 1) * a.xstride, more ? (uint32_t)(B * (NP * G * 2)) : 0u);
    const uint32_t so_h = (uint32_t)(b * (NP * F * 2)), so_x = (uint32_t)(b * (NP * G * 2));
    [[maybe_unused]] const bool stamp_on = (step == (a.nsteps > 2 ? a.nsteps - 3 : 0)) && b == (int)blockIdx.x;      // a typical step (diagnostic builds)
    GCRNN_STAMP32(0);

    // acc[i][h] += W_tap(chunk c, half h) [h|x]^T over the wave's 8 tiles (the seed's tap; the hops' taps live inside the asm block)
    auto taps = [&](int tap) {
      const uint32_t wofs = (uint32_t)WOFF + (uint32_t)lane_now() * 16u + (uint32_t)(tap * 2 * KS * 1024);
      p32_forn<2 * KS>([&](auto hsc) {
        constexpr int hs = decltype(hsc)::value, h = hs / KS, s = hs - h * KS;
        const bf16x8 afr = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + wofs + (uint32_t)(hs * 1024)));
        p32_forn<STILES>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const bf16x8 bfr = __builtin_bit_cast(bf16x8, opget(std::integral_constant<int, s>{}, ic));
          if constexpr (h == 0) p32_set4<i>(acc0, __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, p32_get4<i>(acc0), 0, 0, 0));
          else p32_set4<i>(acc1, __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, p32_get4<i>(acc1), 0, 0, 0));
        });
      });
    };
    auto zero_acc = [&]() {
#pragma unroll
      for (int e = 0; e < 32; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    };
    auto put = [&]() {
      int sw[STILES];
      slot_words(lane_now(), sw);
      p32_forn<STILES>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        state_put<true>(reinterpret_cast<float*>(smem), sw[i], p32_get4<i>(acc0));
        state_put<true>(reinterpret_cast<float*>(smem + PL), sw[i], p32_get4<i>(acc1));
      });
    };
    auto seed = [&]() {
      zero_acc();
      taps(K - 1);
      put();
    };
    seed();

#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk) {
      const bool last = chunk == NCH - 1;
      lds_barrier();      // the seed is in the image
      GCRNN_STAMP32(1 + chunk * 24);

      // inline pack, TWO hops ahead: virtual round v of this step -> (target step, round); false: nothing to lay out
      auto pack_target = [&](int v, int& tgt, int& rnd) -> bool {
        v += 2 * RPH;
        const int wrap = v >= NRND ? 1 : 0;
        rnd = v - wrap * NRND;
        tgt = step + 1 + wrap;
        return pk_any && tgt >= 2 && tgt < a.nsteps;
      };
      auto pack_issue = [&](int v) {
        int tgt, rnd;
        if (!pack_target(v, tgt, rnd)) return;
        const uint16_t* pk_src = a.pk_src0 + (int64_t)tgt * a.pksrc_stride;
        constexpr int PPR = NPCK / 8, PIECES = PKROWS * PPR;
        static_assert(PIECES % STHREADS == 0, "whole pieces per thread");
        const uint16_t* xsrc = pk_src + pk_soff + rnd * NPCK;      // (wave-uniform)
        const int tl = wave * 64 + lane_now();
#pragma unroll
        for (int i = 0; i < PIECES / STHREADS; ++i) {
          const int id = i * STHREADS + tl;
          const int row = id / PPR, cs = id - row * PPR;
          const int col = (cs - (row >> 3)) & (PPR - 1);
          if (rnd * NPCK + col * 8 < N)
            p32_dma16(xsrc, (uint32_t)((row * N + col * 8) * 2), __builtin_amdgcn_readfirstlane(xtile_off + (uint32_t)((i * STHREADS + wave * 64) * 16)));
        }
      };
      auto pack_drain = [&](int v) -> bool {
        int tgt, rnd;
        if (!pack_target(v, tgt, rnd)) return false;
        uint16_t* pk_dst = a.pk_dst0 + (int64_t)tgt * a.pkdst_stride;
        constexpr int PCS = PKROWS / 8, RI = PCS * NPCK / STHREADS;
        static_assert(PCS * NPCK % STHREADS == 0 && (RI == 1 || RI == 2), "one or two row pieces per thread");
        const __amdgpu_buffer_rsrc_t rsrc_pk = __builtin_amdgcn_make_buffer_rsrc(pk_dst, 0, B * (NP * PKROWS * 2), 0x00020000);
        const int tl = wave * 64 + lane_now();
        p32_u32x4 vv[RI];
        uint32_t h16[RI][8];
        uint32_t sa[RI];
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          const int id = i * STHREADS + tl;
          const int nl = id / PCS, pc = id - nl * PCS;
          sa[i] = xtile_off + (uint32_t)((pc * 8) * (NPCK * 2) + ((nl + 8 * pc) & (NPCK - 1)) * 2);
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
          vv[i] = p32_u32x4{ok ? w4[0] : 0u, ok ? w4[1] : 0u, ok ? w4[2] : 0u, ok ? w4[3] : 0u};
        }
#pragma unroll
        for (int i = 0; i < RI; ++i) asm volatile("" : "+v"(vv[i]));          // every piece in its own tuple before the first store (gfx950 store-data hazard)
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          const int id = i * STHREADS + tl;
          const int nl = id / PCS, pc = id - nl * PCS;
          __builtin_amdgcn_raw_buffer_store_b128(vv[i], rsrc_pk, (rnd * NPCK + nl) * (PKROWS * 2) + pc * 16 + b * (NP * PKROWS * 2), 0, 0);
        }
        return true;
      };
      // weight fragments of chunk wc, tap `tap` -> their place in LDS (2 KS pieces of 1 KB), one tap at a time as in gcrnn_fused_seq32.h
      auto weights_issue = [&](int wc, int tap) {
        const char* wsrc = reinterpret_cast<const char*>(a.wpack) + (size_t)wc * WB + (size_t)tap * (2 * KS * 1024);
        const int ln = lane_now();
#pragma unroll
        for (int i = 0; i < (2 * KS + SWAVES - 1) / SWAVES; ++i) {
          const int piece = i * SWAVES + wave;
          if (piece < 2 * KS)
            p32_dma16(wsrc + piece * 1024, (uint32_t)(ln * 16), __builtin_amdgcn_readfirstlane((uint32_t)(WOFF + tap * (2 * KS * 1024) + piece * 1024)));
        }
      };

      bool drained_last = false;      // (wave-uniform) the last hop's write-back stored pack rows behind the operand requests
      bool requested = false;
#pragma unroll 1
      for (int j = 1; j <= K - 1; ++j) {
        const int r0 = (chunk * (K - 1) + (j - 1)) * RPH;      // this hop's first pack round
        zero_acc();
        // this wave's LDS-DMA pieces of the hop, right in front of its block (they have the whole stream to land)
        if (NCH > 1) {
          weights_issue((chunk + 1) % NCH, K - j);
          if (j == 1 && K > 2) weights_issue(chunk, 0);
        }
        if (r0 < NRND) pack_issue(r0);
        const uint32_t wofs = (uint32_t)(WOFF + (K - 1 - j) * (2 * KS * 1024));
        GCRNN_STAMP32(1 + chunk * 24 + 4 * (j - 1) + 1);
        if (GCRNN_P32_PRIO == 1 && wave >= SWAVES / 2) __builtin_amdgcn_s_setprio(1);
        if (GCRNN_P32_PRIO == 2 && wave < SWAVES / 2) __builtin_amdgcn_s_setprio(1);
        if (j == 2 && chunk == 0) GCRNN_STAMP32_WAVE(56);
        if (GCRNN_P32_MODE == 1) {
          const bool stream_first = wave < SWAVES / 2;
          if (!stream_first) taps(K - 1 - j);
          if constexpr (HS == 2 && XS == 2) P32_HOP(GCRNN_HOP_ASM_P32_STREAM_TEXT_2_2, P32_OPS_IN_4, wofs);
          else if constexpr (HS == 2 && XS == 1) P32_HOP(GCRNN_HOP_ASM_P32_STREAM_TEXT_2_1, P32_OPS_IN_3, wofs);
          else P32_HOP(GCRNN_HOP_ASM_P32_STREAM_TEXT_1_1, P32_OPS_IN_2, wofs);
          if (stream_first) taps(K - 1 - j);
        } else if (last && j == K - 1 && more) {
          // (the asm text is a string literal: one statement per (HS, XS), the others are discarded)
          if constexpr (HS == 2 && XS == 2) P32_HOP_LOADS(GCRNN_HOP_ASM_P32_LOADS_TEXT_2_2, P32_OPS_IO_4, wofs, rs_hn, rs_xn, so_h, so_x, slot_wave);
          else if constexpr (HS == 2 && XS == 1) P32_HOP_LOADS(GCRNN_HOP_ASM_P32_LOADS_TEXT_2_1, P32_OPS_IO_3, wofs, rs_hn, rs_xn, so_h, so_x, slot_wave);
          else P32_HOP_LOADS(GCRNN_HOP_ASM_P32_LOADS_TEXT_1_1, P32_OPS_IO_2, wofs, rs_hn, rs_xn, so_h, so_x, slot_wave);
          requested = true;
        } else {
          if constexpr (HS == 2 && XS == 2) P32_HOP(GCRNN_HOP_ASM_P32_TEXT_2_2, P32_OPS_IN_4, wofs);
          else if constexpr (HS == 2 && XS == 1) P32_HOP(GCRNN_HOP_ASM_P32_TEXT_2_1, P32_OPS_IN_3, wofs);
          else P32_HOP(GCRNN_HOP_ASM_P32_TEXT_1_1, P32_OPS_IN_2, wofs);
        }
        if (GCRNN_P32_PRIO) __builtin_amdgcn_s_setprio(0);
        GCRNN_STAMP32(1 + chunk * 24 + 4 * (j - 1) + 2);
        if (j == 2 && chunk == 0) GCRNN_STAMP32_WAVE(64);
        lds_barrier();      // every wave has left the image (and the weights, after the last hop); every piece of the pack tile is in
        GCRNN_STAMP32(1 + chunk * 24 + 4 * (j - 1) + 3);
        if (GCRNN_P32_PREFETCH && !PKV && last && more && j < K - 1) {      // (with the inline pack x_{t+1} has just been written by this CU: no gain measured)
          // L2 prefetch of the next operand, issued in the write-back phase (behind the hop's own wait: the NEXT hop's wait covers it). A CU
          // has ~8 KB of misses in flight, so what it fetches costs 8 KB / latency: ~10 B per clock from HBM, ~29 from L2 (MI355X_MICROARCH.md,
          // "Indexed rows") -- the operand requests of the last hop then hit lines that are already on their way or in L2.
          // x_{t+1}: NP G 2 bytes = LINES 128-byte lines; instruction m = k * 8 + wave touches lines 64 m .. + 63, one dword each.
          constexpr int LINES = NP * G * 2 / 128, NPF = LINES / 64;      // instructions per step and workgroup
          constexpr int H0 = (GCRNN_P32_PREFETCH == 2) ? 1 : (GCRNN_P32_PREFETCH == 3 ? K - 2 : 2), NHOP = K - 1 - H0;      // hops H0 .. K-2 share them
          const uint32_t voff = (uint32_t)lane_now() * 128u;
#pragma unroll
          for (int k = 0; k < (NPF + SWAVES - 1) / SWAVES; ++k) {
            const int m = k * SWAVES + wave;
            const int hm = H0 + (NHOP > 0 ? m * NHOP / NPF : 0);
            if (m < NPF && (K - 1 <= H0 ? j == K - 2 : hm == j)) {
              const uint32_t soff = so_x + (uint32_t)(m * 64 * 128);
              asm volatile("s_mov_b32 m0, %3\n\tbuffer_load_dword %0, %1, %2 offen lds" ::"v"(voff), "s"(rs_xn), "s"(soff), "s"(pf_lds) : "memory");
            }
          }
#if defined(GCRNN_P32_PREFETCH_H)
          // ... and the first HS - 1 chunks of h_t (stored at their chunk's end, a whole chunk ago): the lines of its rows
          if (HS > 1 && j == K - 2) {
            constexpr int HLINES = NP * F * 2 / 128;
#pragma unroll
            for (int k = 0; k < (HLINES / 64 + SWAVES - 1) / SWAVES; ++k) {
              const int m = k * SWAVES + wave;
              if (m < HLINES / 64) {
                const uint32_t soff = so_h + (uint32_t)(m * 64 * 128);
                asm volatile("s_mov_b32 m0, %3\n\tbuffer_load_dword %0, %1, %2 offen lds" ::"v"(voff), "s"(rs_hn), "s"(soff), "s"(pf_lds) : "memory");
              }
            }
          }
#endif
        }
        if (j < K - 1) put();
        if (K == 2 && NCH > 1) weights_issue((chunk + 1) % NCH, 0);
        if (r0 < NRND) drained_last = pack_drain(r0);
#pragma unroll
        for (int e = 1; e < RPH; ++e) {       // (fewer hops than rounds: the extra rounds are not hidden behind a stream)
          if (pk_any && r0 + e < NRND) {
            lds_barrier();
            pack_issue(r0 + e);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            pack_drain(r0 + e);
            drained_last = false; requested = false;      // (everything has been waited for)
          }
        }
        if (j < K - 1) lds_barrier();      // the image is complete (and the pack tile read)
        GCRNN_STAMP32(1 + chunk * 24 + 4 * (j - 1) + 4);
      }

      // ---- epilogue: + 2 b, tanh, bf16; lane (r, q) holds features 32 c + 8 q .. + 7 of its node: ONE 16-byte store per tile -------------
      const int lane = lane_now(), q = lane >> 4, tl = wave * 64 + lane;
      float bs[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[h][e] = 2.f * lbias[chunk * 32 + q * 8 + h * 4 + e];      // the one bias is added by both filters (graphML.py:2420-2421)
      int swe[STILES];
      slot_words(lane, swe);
      char* tst = smem;      // transposed user-layout tile [feature pair][node] (nobody reads the image any more: the last hop's barrier)
      p32_forn<STILES>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int node = swe[i] >> 16;
        p32_u32x4 p{0u, 0u, 0u, 0u};
        if (node < N) {
          const auto a0 = p32_get4<i>(acc0), a1 = p32_get4<i>(acc1);
          p[0] = pack2bf(fast_tanh(a0[0] + bs[0][0]), fast_tanh(a0[1] + bs[0][1]));
          p[1] = pack2bf(fast_tanh(a0[2] + bs[0][2]), fast_tanh(a0[3] + bs[0][3]));
          p[2] = pack2bf(fast_tanh(a1[0] + bs[1][0]), fast_tanh(a1[1] + bs[1][1]));
          p[3] = pack2bf(fast_tanh(a1[2] + bs[1][2]), fast_tanh(a1[3] + bs[1][3]));
        }
        __builtin_amdgcn_raw_buffer_store_b128(p, rsrc_o, node * (F * 2) + (chunk * 32 + q * 8) * 2, b * (NP * F * 2), 0);
        if (aux1) {
          char* ra = tst + (4 * q) * RS2 + node * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<uint32_t*>(ra + k * RS2) = p[k];
        }
        // h_t's last 32 features ARE the lanes' B fragments of k-step HS-1: handed to the next step in registers
        if (last) opset(std::integral_constant<int, HS - 1>{}, ic, p);
      });
      // the next operand's requests (issued inside the last hop) are waited for HERE, counted: younger are the 8 state stores and, when the
      // last hop's write-back drained a pack round, its row pieces
      if (GCRNN_P32_MODE == 1 && last && more) {
        // the next step's operand, requested behind the state stores (round 4's place): x_{t+1} and the state features of the earlier chunks;
        // hipcc counts these loads and waits where the seed's MFMAs first use them
        const __amdgpu_buffer_rsrc_t rsrc_hn = __builtin_amdgcn_make_buffer_rsrc(hout, 0, B * (NP * F * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_xn = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.x0 + (int64_t)(step + 1) * a.xstride), 0, B * (NP * G * 2), 0x00020000);
        p32_forn<KS>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if constexpr (s != HS - 1) {
            p32_forn<STILES>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              const int node = swe[i] >> 16;
              if constexpr (s < HS)
                opset(sc, ic, __builtin_bit_cast(p32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_hn, node * (F * 2) + 16 * q + 64 * s, b * (NP * F * 2), 0)));
              else
                opset(sc, ic, __builtin_bit_cast(p32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_xn, node * (G * 2) + 16 * q + 64 * (s - HS), b * (NP * G * 2), 0)));
            });
          }
        });
      }
      GCRNN_STAMP32(1 + chunk * 24 + 17);
      auto wait_requests = [&]() {
        if (!requested) return;
        constexpr int RIp = (PKROWS / 8) * NPCK / STHREADS;
        if (GCRNN_P32_WAIT_AT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (drained_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + RIp) : "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      };
      if (aux1) {
        lds_barrier();
        if (GCRNN_P32_WAIT_AT == 0) wait_requests();
        GCRNN_STAMP32(1 + chunk * 24 + 18);
        const int segs = N >> 3;
        uint16_t* ub = const_cast<uint16_t*>(aux1) + (int64_t)b * ubstride + (int64_t)(chunk * 32) * N;
        const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(ub, 0, 32 * N * 2, 0x00020000);
        // 32 threads per feature-pair row, 16-byte segments l32, l32 + 32, ..: no division by a run-time N (its reciprocal would have to live
        // -- or be spilled -- across the hops), and the same number of row stores in every wave
        const int fp = tl >> 5;
        for (int sg = tl & 31; sg < segs; sg += 32) {
          const p32_u32x4 w0 = *reinterpret_cast<const p32_u32x4*>(tst + fp * RS2 + sg * 32);
          const p32_u32x4 w1 = *reinterpret_cast<const p32_u32x4*>(tst + fp * RS2 + sg * 32 + 16);
          const p32_u32x4 ev = {__builtin_amdgcn_perm(w0[1], w0[0], 0x05040100u), __builtin_amdgcn_perm(w0[3], w0[2], 0x05040100u),
                                __builtin_amdgcn_perm(w1[1], w1[0], 0x05040100u), __builtin_amdgcn_perm(w1[3], w1[2], 0x05040100u)};
          const p32_u32x4 od = {__builtin_amdgcn_perm(w0[1], w0[0], 0x07060302u), __builtin_amdgcn_perm(w0[3], w0[2], 0x07060302u),
                                __builtin_amdgcn_perm(w1[1], w1[0], 0x07060302u), __builtin_amdgcn_perm(w1[3], w1[2], 0x07060302u)};
          __builtin_amdgcn_raw_buffer_store_b128(ev, rsrc_u, ((2 * fp) * N + sg * 8) * 2, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(od, rsrc_u, ((2 * fp + 1) * N + sg * 8) * 2, 0, 0);
        }
        if (GCRNN_P32_WAIT_AT == 1) wait_requests();
      } else {
        wait_requests();
      }
      GCRNN_STAMP32(1 + chunk * 24 + 19);
      // K = 2 only: the next chunk's tap 0 (LDS-DMA behind the last hop) has landed
      if (K == 2 && NCH > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      GCRNN_STAMP32(1 + chunk * 24 + 20);
      if (chunk + 1 < NCH) seed();
    }  // chunks
  }  // steps
  }  // sequences
  GCRNN_STAMP32_FLUSH();
}
