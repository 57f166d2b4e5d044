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
//     u_{K-1} goes to LDS, u_0..u_{K-2} stay in registers (8 tiles x K x 4 fp32 per lane).
// Phase 2 (LDS gather): K-1 hops acc'[n] = sum_m P[n,m] acc[m] + u_k[n] on ONE fp32 [1024][16] image in LDS
//     (64-byte rows): a hop's results stay in the tap's registers until every wave has finished reading, then
//     are written back (two barriers per hop). That leaves room to keep the graph itself in LDS: a
//     degree-sorted sliced ELL (16 nodes per slice, entries [e][16], u16 column + f32 weight), so the
//     neighbour loop is wave-uniform, no gather ever waits on global memory, and each gather is one
//     ds_read_b128 + 4 FMAs per lane. Graphs whose ELL does not fit are read from global memory instead.
// Epilogue: + bias, tanh, bf16 store of the chunk into the node-major state h_t[b][n][c*16 .. +15].
//
// HBM traffic per (sequence, step): read x_t and h_{t-1} (each N*64*2 B; the 4 chunk workgroups of a sequence
// are placed on one XCD so that three of the four reads hit its L2), write h_t: the compulsory
// T*s*N*(G+2F) of SURVEY.md section 8d.
#pragma once
#include "gcrnn_common.h"
#ifndef GCRNN_HOP_ASM_INC
#define GCRNN_HOP_ASM_INC "gcrnn_hop_asm.inc"      // (A/B builds: -DGCRNN_HOP_ASM_INC='"/tmp/variant.inc"' with another output of tools/gen_hop_asm.py)
#endif
#include GCRNN_HOP_ASM_INC

#ifndef GCRNN_STORE_POLICY
#define GCRNN_STORE_POLICY 0   // cache policy of the state stores: 0 plain (default), 16 = sc1 (write-through, line not kept in L2), 2 = nt.
#endif                         // Measured (tools/store_policy_ab.sh, one box, B = 256): plain 119.4 us per step, sc1 124.7, nt 125.4 -- rejected.
#ifndef GCRNN_HOP_ASM
#define GCRNN_HOP_ASM 1      // 1: the hop gather stream is ONE asm block, three groups deep (gcrnn_hop_asm.inc); 0: the two-deep macro stream (A/B)
#endif

#if (!GCRNN_HOP_ASM || (defined(GCRNN_STEP_WAVES) && GCRNN_STEP_WAVES != 8)) && !defined(GCRNN_DIAGNOSTIC_STREAMS)
#error "the compiler-scheduled asm hop pipelines (GCRNN_HOP_ASM=0 / GCRNN_STEP_WAVES != 8) are A/B builds: a spill inside them corrupts results silently (cdna_hip_programming.md 5.7). Build with -DGCRNN_DIAGNOSTIC_STREAMS and check ScratchSize == 0 (tools/kernel_resources.sh)."
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#ifndef GCRNN_HOP16_SPARSE
#define GCRNN_HOP16_SPARSE 1     // bf16-image hop sums: ONE 2:4-sparse v_smfmac_f32_16x16x64_bf16 per four entries (one-hot A, both gathers as its K = 64 operand) instead of two dense MFMAs; 0: the dense pair (A/B)
#endif
#ifndef GCRNN_P1_AHEAD
#define GCRNN_P1_AHEAD 0           // tiles (of 8 per wave) of the next sequence's operand requested during the last hop (experiment: every depth spills)
#endif
namespace {
constexpr int FC = 16;          // output features per workgroup
constexpr int WAVES = 8;        // weight-gradient kernel: 8 waves x 8 tiles
constexpr int TILES = 8;        // node tiles (16 nodes) per wave
constexpr int NP = WAVES * TILES * 16;   // 1024 padded nodes
#ifndef GCRNN_STEP_WAVES
#define GCRNN_STEP_WAVES 8
#endif
#ifndef GCRNN_P1_GROUP
#define GCRNN_P1_GROUP 2      // tiles that share a weight fragment in phase 1 of the step kernel (1, 2 or 4)
#endif
constexpr int SWAVES = GCRNN_STEP_WAVES;      // step kernel: waves per workgroup ...
constexpr int STILES = NP / 16 / SWAVES;      // ... and node tiles per wave
constexpr int STHREADS = 64 * SWAVES;
}

__device__ __forceinline__ uint16_t f2bf(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN-safe
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// two floats -> one dword {bf16(a), bf16(b) << 16}: ONE v_cvt_pk_bf16_f32 (f2bf(a) | f2bf(b) << 16 compiles to two conversions, a shift and
// an or -- 8 instead of 2 vector instructions for a lane's four features, at every hop's write-back and in every epilogue)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2v_t;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2v_t{a, b}, bf16x2v_t));
}

// tanh(x) = 1 - 2 / (1 + exp(2x)) on the hardware exp2/rcp units: abs error < 3e-7 for all x (inf-safe: exp -> inf
// gives 1, exp -> 0 gives -1), far below the bf16 rounding of the stored state.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);      // exp(2x)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
// Write this lane's 4 features of a node into the LDS hop image: fp32 (16 bytes of a 64-byte row) or, IMG16, bf16 (8 bytes of a
// 32-byte row; the hop then sums gathered rows on the matrix cores, GCRNN_HOP_ASM_UNI16_STREAM).
template <bool IMG16>
__device__ __forceinline__ void state_put(float* state, int wv, const f32x4& v) {
  char* p = reinterpret_cast<char*>(state) + (wv & 0xffff);
  if constexpr (IMG16) {
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int u32x2_;
    *reinterpret_cast<u32x2_*>(p) = u32x2_{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
  } else {
    *reinterpret_cast<f32x4*>(p) = v;
  }
}
// Workgroup barrier for LDS hand-offs only: __syncthreads() is a workgroup-scope fence too, for which hipcc drains EVERY outstanding
// vector-memory operation first (s_waitcnt vmcnt(0)) -- at the barriers of the epilogue that is the drain of the state stores just
// issued (measured in the sequence-resident kernel: ~1.5 k cycles per chunk). Where only LDS contents change hands, waiting for this
// wave's LDS operations is enough; the places that hand GLOBAL data between waves (LDS-DMA tiles, a step boundary inside a launch) wait
// for vmcnt(0) explicitly.
#ifdef GCRNN_FENCED_BARRIERS      // A/B switch (tools/ab_build.sh): the compiler's fenced barrier everywhere, as in rounds 1-2
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// ------------------------------------------------------------------------------------------
// the fused step
// ------------------------------------------------------------------------------------------
// Hand-pipelined LDS reads for the hop loop (cdna_hip_programming.md section 5.7): hipcc does not count asm loads,
// so every wait below is ours. LDS ops of one wave return in order, hence lgkmcnt(N) = "all but the N youngest".
#define DS_READ_B64(dst, addr) asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr))
#define DS_READ_B128(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n)                                              \
  do {                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);                            \
  } while (0)
#define KEEP_ALIVE(v) asm volatile("" ::"v"(v))

// One gather trip of the resident pipeline: group g has its weights in VC and its 4 gathered quads in XC*;
// issue cols(g+2) -> CC, then (after cols(g+1) = CN landed) vals(g+1) -> VN and the gathers of g+1 -> XN*;
// then wait for VC / XC* (issued one trip earlier) and do the 16 FMAs. lgkmcnt(6): the 6 reads just issued may
// stay in flight. E and O name the two ping-pong register sets. Uses the locals g, gwend, gwlast, colb, valb, lds0, qx.
#define GCRNN_TRIP_WAIT LGKM_WAIT(6)
#define GCRNN_TRIP_WAIT2 LGKM_WAIT(6)
#define GCRNN_TRIP(ACC, CC, CN, VC, VN, XC0, XC1, XC2, XC3, XN0, XN1, XN2, XN3)                    \
  do {                                                                                             \
    const int g1_ = (g + 1 < gwend) ? g + 1 : gwlast, g2_ = (g + 2 < gwend) ? g + 2 : gwlast;      \
    DS_READ_B64(CC, colb + g2_ * 128);                                                             \
    GCRNN_TRIP_WAIT;                                                                               \
    DS_READ_B128(VN, valb + g1_ * 256);                                                            \
    DS_READ_B128(XN0, lds0 + (((uint32_t)CN & 0xffffu) ^ qx));                                     \
    DS_READ_B128(XN1, lds0 + ((((uint32_t)CN >> 16) & 0xffffu) ^ qx));                             \
    DS_READ_B128(XN2, lds0 + (((uint32_t)(CN >> 32) & 0xffffu) ^ qx));                             \
    DS_READ_B128(XN3, lds0 + ((uint32_t)(CN >> 48) ^ qx));                                         \
    GCRNN_TRIP_WAIT2;                                                                              \
    ACC += VC[0] * XC0;                                                                            \
    ACC += VC[1] * XC1;                                                                            \
    ACC += VC[2] * XC2;                                                                            \
    ACC += VC[3] * XC3;                                                                            \
  } while (0)
#define GCRNN_TRIP_E(ACC) GCRNN_TRIP(ACC, cE, cO, vE, vO, xE0, xE1, xE2, xE3, xO0, xO1, xO2, xO3)
#define GCRNN_TRIP_O(ACC) GCRNN_TRIP(ACC, cO, cE, vO, vE, xO0, xO1, xO2, xO3, xE0, xE1, xE2, xE3)

// One hop of one wave over the LDS-resident graph: acc_i = INIT(i) + sum over the neighbours of tile i's slots, for
// the wave's TILES tiles. The tiles are stored back to back, so the whole hop is ONE continuous stream of groups:
// the pipeline is primed once, runs across tile boundaries (only the accumulator changes) and is drained once.
// Uses the locals tbeg, tend, lds_col, lds_val, lds0, qx, r.
#define GCRNN_HOP_STREAM(INIT, STORE)                                                              \
  do {                                                                                             \
    const int gwbeg = tbeg[0] >> 2, gwend = tend[HT - 1] >> 2, gwlast = gwend - 1;              \
    const uint32_t colb = lds_col + r * 8, valb = lds_val + r * 16;                                \
    uint64_t cE = 0, cO = 0;                                                                       \
    f32x4 vE, vO, xE0, xE1, xE2, xE3, xO0, xO1, xO2, xO3;                                          \
    int g = gwbeg;                                                                                 \
    int par = 0; /* 0: the current group sits in set E, 1: in set O (wave-uniform) */              \
    if (gwbeg < gwend) {                                                                           \
      DS_READ_B64(cE, colb + gwbeg * 128);                                                         \
      DS_READ_B64(cO, colb + ((gwbeg + 1 < gwend) ? gwbeg + 1 : gwlast) * 128);                    \
      DS_READ_B128(vE, valb + gwbeg * 256);                                                        \
      LGKM_WAIT(2);                                                                                \
      DS_READ_B128(xE0, lds0 + (((uint32_t)cE & 0xffffu) ^ qx));                                   \
      DS_READ_B128(xE1, lds0 + ((((uint32_t)cE >> 16) & 0xffffu) ^ qx));                           \
      DS_READ_B128(xE2, lds0 + (((uint32_t)(cE >> 32) & 0xffffu) ^ qx));                           \
      DS_READ_B128(xE3, lds0 + ((uint32_t)(cE >> 48) ^ qx));                                       \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < HT; ++i) {                                               \
      const int ge = tend[i] >> 2;                                                                 \
      f32x4 acc = INIT(i);                                                                         \
      if (g < ge) {                                                                                \
        if (par) { GCRNN_TRIP_O(acc); ++g; par = 0; }                                              \
        while (g < ge) {                                                                           \
          GCRNN_TRIP_E(acc);                                                                       \
          if (++g >= ge) { par = 1; break; }                                                       \
          GCRNN_TRIP_O(acc);                                                                       \
          ++g;                                                                                     \
        }                                                                                          \
      }                                                                                            \
      STORE(i, acc);                                                                               \
    }                                                                                              \
    if (gwbeg < gwend) {                                                                           \
      LGKM_WAIT(0); /* drain the tail prefetches before their registers may be reused */           \
      KEEP_ALIVE(cE); KEEP_ALIVE(cO); KEEP_ALIVE(vE); KEEP_ALIVE(vO);                              \
      KEEP_ALIVE(xE0); KEEP_ALIVE(xE1); KEEP_ALIVE(xE2); KEEP_ALIVE(xE3);                          \
      KEEP_ALIVE(xO0); KEEP_ALIVE(xO1); KEEP_ALIVE(xO2); KEEP_ALIVE(xO3);                          \
    }                                                                                              \
  } while (0)

// The same hop as ONE asm block (generated: gcrnn_hop_asm.inc / tools/gen_hop_asm.py): every in-flight register is named
// inside the block, so no compiler copy can sit between an LDS read and its counted wait, and the stream runs THREE groups
// deep -- 12 LDS reads of groups n+1, n+2 in flight while group n is consumed, one s_waitcnt per trip. The accumulators go in
// and out as 64-bit halves (v_pk_fma_f32 operands). Needs 8 tiles per wave and the dynamic LDS segment at LDS address 0.
#define GCRNN_HOP_ASM_STREAM(INIT, STORE)                                                          \
  do {                                                                                             \
    static_assert(HT == 8, "the asm hop stream is generated for 8 tiles per wave");                \
    const int gwbeg = tbeg[0] >> 2, gwend = tend[HT - 1] >> 2;                                      \
    f32x2 al_[8], ah_[8];                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                \
      const f32x4 a_ = INIT(i);                                                                    \
      al_[i] = f32x2{a_[0], a_[1]};                                                                \
      ah_[i] = f32x2{a_[2], a_[3]};                                                                \
    }                                                                                              \
    if (gwbeg < gwend) {                                                                           \
      const uint32_t colb = lds_col + r * 8, valb = lds_val + r * 16;                              \
      asm volatile(GCRNN_HOP_ASM_TEXT                                                              \
                   : "+v"(al_[0]), "+v"(ah_[0]), "+v"(al_[1]), "+v"(ah_[1]), "+v"(al_[2]), "+v"(ah_[2]), "+v"(al_[3]), "+v"(ah_[3]),  \
                     "+v"(al_[4]), "+v"(ah_[4]), "+v"(al_[5]), "+v"(ah_[5]), "+v"(al_[6]), "+v"(ah_[6]), "+v"(al_[7]), "+v"(ah_[7])   \
                   : "s"(tend[0] >> 2), "s"(tend[1] >> 2), "s"(tend[2] >> 2), "s"(tend[3] >> 2), "s"(tend[4] >> 2),                   \
                     "s"(tend[5] >> 2), "s"(tend[6] >> 2), "s"(tend[7] >> 2), "s"(gwbeg), "s"(gwend - 1), "v"(colb), "v"(valb), "v"(qx) \
                   : GCRNN_HOP_ASM_CLOBBERS);                                                      \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) STORE(i, (f32x4{al_[i][0], al_[i][1], ah_[i][0], ah_[i][1]}));                   \
  } while (0)

// Uniform-weight graphs (every non-zero of S equal: the reference drivers' W / lambda_max): no weight image in LDS, a tile's
// gathered rows are summed and enter its accumulator once, acc = init + w * sum (18 instead of 22 LDS cycles per 4 entries).
// Padding entries point at padding rows, whose state is always zero (gcrnn_ell_fill_z).
#define GCRNN_HOP_ASM_UNI_STREAM(INIT, STORE)                                                      \
  do {                                                                                             \
    static_assert(HT == 8, "the asm hop stream is generated for 8 tiles per wave");                \
    const int gwbeg = tbeg[0] >> 2, gwend = tend[HT - 1] >> 2;                                      \
    f32x2 al_[8], ah_[8];                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                \
      const f32x4 a_ = INIT(i);                                                                    \
      al_[i] = f32x2{a_[0], a_[1]};                                                                \
      ah_[i] = f32x2{a_[2], a_[3]};                                                                \
    }                                                                                              \
    if (gwbeg < gwend) {                                                                           \
      const uint32_t colb = lds_col + r * 8;                                                       \
      const f32x2 wp_ = f32x2{uni_w, uni_w};                                                       \
      asm volatile(GCRNN_HOP_ASM_UNI_TEXT                                                          \
                   : "+v"(al_[0]), "+v"(ah_[0]), "+v"(al_[1]), "+v"(ah_[1]), "+v"(al_[2]), "+v"(ah_[2]), "+v"(al_[3]), "+v"(ah_[3]),  \
                     "+v"(al_[4]), "+v"(ah_[4]), "+v"(al_[5]), "+v"(ah_[5]), "+v"(al_[6]), "+v"(ah_[6]), "+v"(al_[7]), "+v"(ah_[7])   \
                   : "s"(tend[0] >> 2), "s"(tend[1] >> 2), "s"(tend[2] >> 2), "s"(tend[3] >> 2), "s"(tend[4] >> 2),                   \
                     "s"(tend[5] >> 2), "s"(tend[6] >> 2), "s"(tend[7] >> 2), "s"(gwbeg), "s"(gwend - 1), "v"(colb), "v"(qx), "v"(wp_)   \
                   : GCRNN_HOP_ASM_UNI_CLOBBERS);                                                  \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) STORE(i, (f32x4{al_[i][0], al_[i][1], ah_[i][0], ah_[i][1]}));                   \
  } while (0)

// Same hop, but the pipeline is primed and drained per tile: nothing is in flight at the control-flow joins between
// the unrolled tiles. Needed where register pressure makes hipcc insert copies of the ping-pong sets at those joins
// (a copy of a register whose asm load has not landed yet captures stale data: cdna_hip_programming.md 5.7 item 1).
#define GCRNN_HOP_TILED(INIT, STORE)                                                               \
  do {                                                                                             \
    const uint32_t colb = lds_col + r * 8, valb = lds_val + r * 16;                                \
    _Pragma("unroll") for (int i = 0; i < HT; ++i) {                                               \
      const int gwbeg = tbeg[i] >> 2, gwend = tend[i] >> 2, gwlast = gwend - 1;                    \
      f32x4 acc = INIT(i);                                                                         \
      if (gwbeg < gwend) {                                                                         \
        uint64_t cE, cO;                                                                           \
        f32x4 vE, vO, xE0, xE1, xE2, xE3, xO0, xO1, xO2, xO3;                                      \
        DS_READ_B64(cE, colb + gwbeg * 128);                                                       \
        DS_READ_B64(cO, colb + ((gwbeg + 1 < gwend) ? gwbeg + 1 : gwlast) * 128);                  \
        DS_READ_B128(vE, valb + gwbeg * 256);                                                      \
        LGKM_WAIT(2);                                                                              \
        DS_READ_B128(xE0, lds0 + (((uint32_t)cE & 0xffffu) ^ qx));                                 \
        DS_READ_B128(xE1, lds0 + ((((uint32_t)cE >> 16) & 0xffffu) ^ qx));                         \
        DS_READ_B128(xE2, lds0 + (((uint32_t)(cE >> 32) & 0xffffu) ^ qx));                         \
        DS_READ_B128(xE3, lds0 + ((uint32_t)(cE >> 48) ^ qx));                                     \
        int g = gwbeg;                                                                             \
        while (true) {                                                                             \
          GCRNN_TRIP_E(acc);                                                                       \
          if (++g >= gwend) break;                                                                 \
          GCRNN_TRIP_O(acc);                                                                       \
          if (++g >= gwend) break;                                                                 \
        }                                                                                          \
        LGKM_WAIT(0);                                                                              \
        KEEP_ALIVE(cE); KEEP_ALIVE(cO); KEEP_ALIVE(vE); KEEP_ALIVE(vO);                            \
        KEEP_ALIVE(xE0); KEEP_ALIVE(xE1); KEEP_ALIVE(xE2); KEEP_ALIVE(xE3);                        \
        KEEP_ALIVE(xO0); KEEP_ALIVE(xO1); KEEP_ALIVE(xO2); KEEP_ALIVE(xO3);                        \
      }                                                                                            \
      STORE(i, acc);                                                                               \
    }                                                                                              \
  } while (0)

// LDS map (dynamic): state [NP][16] fp32 (64 KiB) | weight fragments K*KS KiB | RESIDENT: lval4 (f32x4), lcol4 (u16x4).
// Tiles hold 16 nodes of similar degree: tile_nodes[p] lists the node of every slot p (degree-sorted order,
// padded with node ids >= N that have no edges); memory rows are in natural node order.
// EPI: 0 = state epilogue (bias, tanh, bf16 store), 1 = time-gate pre-pass (dot-reduce; optionally also stores the
// gate cell's state c = tanh(pre) for its BPTT), 2 = BPTT data-gradient step (optionally scaled by the forget gate),
// 3 = gate-gradient pass: sum_{f,n} (filter output + b) * dpre of every item (the gradient w.r.t. a scalar time gate),
// 4 = filter-output pass: stores (filter output + b) of every item, bf16 sequence-major (the input filter A(S)x_t + b of the
//     node-gated cell for all t at once: it does not depend on the recurrence),
// 5 = node-gated step (graphML.py:2379-2407), state-only operand (XS = 0):
//     h_t = tanh(gi ni_t[n] Yx_t[n][f] + gf nf_t[n] (B(S)h_{t-1} + b)[n][f]);  Yx_t from the EPI 4 pass (aux0), node gates [2][B][N]
//     fp32 in gate_w (ni then nf), scalar time gates gi / gf [B] or null (= 1); optionally stores Yh = B(S)h_{t-1} + b (for BPTT)
// 6 = the state epilogue of EPI 0 plus a fused output head Linear(F -> 1) (gate_w = its weights [F], gate_out = partials [B][F/16][N]);
//     no user-layout copy of h_t
// The uniform-weight hop on a bf16 image, summed on the matrix cores (generator: gen_uniform16): a lane gathers 16 bytes = half a
// neighbour's 32-byte row, v_mfma_f32_16x16x32_bf16 with a one-hot A operand adds two neighbours per instruction into D, whose layout
// is the accumulators'. A operand of lane (i = lane & 15, kg = lane >> 4): A[i][8 kg + s] = 1 iff i == 8 (kg & 1) + s.
// (tile ranges are wave-uniform; should they have been parked in vector registers, the "s" operands get them back through v_readfirstlane)
#define GCRNN_SGPR(x) __builtin_amdgcn_readfirstlane(x)
#define GCRNN_HOP_ASM_UNI16_STREAM(INIT, STORE)                                                    \
  do {                                                                                             \
    static_assert(HT == 8, "the asm hop stream is generated for 8 tiles per wave");                \
    const int gwbeg = tbeg[0] >> 2, gwend = tend[HT - 1] >> 2;                                      \
    f32x2 al_[8], ah_[8];                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                \
      const f32x4 a_ = INIT(i);                                                                    \
      al_[i] = f32x2{a_[0], a_[1]};                                                                \
      ah_[i] = f32x2{a_[2], a_[3]};                                                                \
    }                                                                                              \
    if (gwbeg < gwend) {                                                                           \
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4a_;     \
      const uint32_t colb = lds_col + r * 8 + (q >> 1) * 4;       /* this lane's own column dword of a slot's pair */ \
      const f32x2 wp_ = f32x2{uni_w, uni_w};                                                       \
      const int hit_ = ((r >> 3) == (q & 1)) ? (r & 7) : 8;       /* position of this lane's 1.0 among its 8 A elements, or none */ \
      const uint32_t one_ = (hit_ & 1) ? 0x3f800000u : 0x00003f80u;                                \
      u32x4a_ aop_ = {(hit_ >> 1) == 0 ? one_ : 0u, (hit_ >> 1) == 1 ? one_ : 0u, (hit_ >> 1) == 2 ? one_ : 0u, (hit_ >> 1) == 3 ? one_ : 0u}; \
      const uint32_t qh_ = (uint32_t)(q & 1) << 4;                                                 \
      if constexpr (GCRNN_HOP16_SPARSE) {                                                            \
        asm volatile(GCRNN_HOP_ASM_UNI16_SPARSE_TEXT                                               \
                     : "+v"(al_[0]), "+v"(ah_[0]), "+v"(al_[1]), "+v"(ah_[1]), "+v"(al_[2]), "+v"(ah_[2]), "+v"(al_[3]), "+v"(ah_[3]),  \
                       "+v"(al_[4]), "+v"(ah_[4]), "+v"(al_[5]), "+v"(ah_[5]), "+v"(al_[6]), "+v"(ah_[6]), "+v"(al_[7]), "+v"(ah_[7])   \
                     : "s"(GCRNN_SGPR(tend[0] >> 2)), "s"(GCRNN_SGPR(tend[1] >> 2)), "s"(GCRNN_SGPR(tend[2] >> 2)), "s"(GCRNN_SGPR(tend[3] >> 2)), "s"(GCRNN_SGPR(tend[4] >> 2)), \
                       "s"(GCRNN_SGPR(tend[5] >> 2)), "s"(GCRNN_SGPR(tend[6] >> 2)), "s"(GCRNN_SGPR(tend[7] >> 2)), "s"(GCRNN_SGPR(gwbeg)), "s"(GCRNN_SGPR(gwend - 1)), "v"(colb), "v"(qh_), "v"(wp_) \
                     : GCRNN_HOP_ASM_UNI16_SPARSE_CLOBBERS);                                        \
      } else {                                                                                     \
        asm volatile(GCRNN_HOP_ASM_UNI16_TEXT                                                        \
                   : "+v"(al_[0]), "+v"(ah_[0]), "+v"(al_[1]), "+v"(ah_[1]), "+v"(al_[2]), "+v"(ah_[2]), "+v"(al_[3]), "+v"(ah_[3]),  \
                     "+v"(al_[4]), "+v"(ah_[4]), "+v"(al_[5]), "+v"(ah_[5]), "+v"(al_[6]), "+v"(ah_[6]), "+v"(al_[7]), "+v"(ah_[7])   \
                   : "s"(GCRNN_SGPR(tend[0] >> 2)), "s"(GCRNN_SGPR(tend[1] >> 2)), "s"(GCRNN_SGPR(tend[2] >> 2)), "s"(GCRNN_SGPR(tend[3] >> 2)), "s"(GCRNN_SGPR(tend[4] >> 2)), \
                     "s"(GCRNN_SGPR(tend[5] >> 2)), "s"(GCRNN_SGPR(tend[6] >> 2)), "s"(GCRNN_SGPR(tend[7] >> 2)), "s"(GCRNN_SGPR(gwbeg)), "s"(GCRNN_SGPR(gwend - 1)), "v"(colb), "v"(qh_), "v"(wp_), \
                     "v"(aop_)                                                                     \
                   : GCRNN_HOP_ASM_UNI16_CLOBBERS);                                                        \
      }                                                \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) STORE(i, (f32x4{al_[i][0], al_[i][1], ah_[i][0], ah_[i][1]}));                   \
  } while (0)

// The summing variant (generator: gen_uniform16(sums=True)): tile i's gathered rows are summed straight into D[i] (f32x4, zeroed here);
// the caller applies acc = init + w * D afterwards. A tile exit only switches tiles -- the per-tile matrix-core -> VALU wait states,
// packed FMAs and re-zeroing of the stream above are gone (they cost 6.5 % of the forward launch: tools/hop16_exit_experiment.sh).
// Used by the sequence-resident kernel, whose taps are evaluated per hop anyway (gcrnn_fused_seq.h).
#define GCRNN_HOP_ASM_UNI16_SUMS_STREAM(D) GCRNN_HOP_ASM_UNI16_SUMS_STREAM_IMG(D, false)
// timing experiment (WRONG results): -DGCRNN_EXPERIMENT_STREAM_WAVES="&& (wave & 1) == 0" lets only some waves stream -- does a wave's
// trip time depend on how many other waves gather? (tools/ab_build.sh)
#ifndef GCRNN_EXPERIMENT_STREAM_WAVES
#define GCRNN_EXPERIMENT_STREAM_WAVES
#elif !defined(GCRNN_SEQ_STAMPS)
#error "GCRNN_EXPERIMENT_STREAM_WAVES gives wrong results by construction: diagnostic stamp builds (tools/seq_stamps.py) only"
#endif
// IMGB_ (compile-time): the gathers read the second hop image, GCRNN_HOP_IMAGE_B_OFFSET bytes behind the first (sequence-resident kernel)
#define GCRNN_HOP_ASM_UNI16_SUMS_ASM_(TEXT_, CLOB_, ...)                                           \
        asm volatile(TEXT_                                                                         \
                     : "+v"(D_[0]), "+v"(D_[1]), "+v"(D_[2]), "+v"(D_[3]), "+v"(D_[4]), "+v"(D_[5]), "+v"(D_[6]), "+v"(D_[7])  \
                     : "s"(GCRNN_SGPR(tend[0] >> 2)), "s"(GCRNN_SGPR(tend[1] >> 2)), "s"(GCRNN_SGPR(tend[2] >> 2)), "s"(GCRNN_SGPR(tend[3] >> 2)), "s"(GCRNN_SGPR(tend[4] >> 2)), \
                       "s"(GCRNN_SGPR(tend[5] >> 2)), "s"(GCRNN_SGPR(tend[6] >> 2)), "s"(GCRNN_SGPR(tend[7] >> 2)), "s"(GCRNN_SGPR(gwbeg)), "s"(GCRNN_SGPR(gwend - 1)), "v"(colb), "v"(qh_) __VA_ARGS__ \
                     : CLOB_)
#define GCRNN_HOP_ASM_UNI16_SUMS_STREAM_IMG(D, IMGB_)                                              \
  do {                                                                                             \
    static_assert(HT == 8, "the asm hop stream is generated for 8 tiles per wave");                \
    const int gwbeg = tbeg[0] >> 2, gwend = tend[HT - 1] >> 2;                                      \
    f32x4 (&D_)[8] = D;                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) D_[i] = f32x4{0.f, 0.f, 0.f, 0.f};              \
    if (gwbeg < gwend GCRNN_EXPERIMENT_STREAM_WAVES) {                                             \
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4a_;     \
      const uint32_t colb = lds_col + r * 8 + (q >> 1) * 4;       /* this lane's own column dword of a slot's pair */ \
      const uint32_t qh_ = (uint32_t)(q & 1) << 4;                                                 \
      if constexpr (GCRNN_HOP16_SPARSE) {                                                          \
        if constexpr (IMGB_) GCRNN_HOP_ASM_UNI16_SUMS_ASM_(GCRNN_HOP_ASM_UNI16_SUMS_SPARSE_TEXT_B, GCRNN_HOP_ASM_UNI16_SUMS_SPARSE_CLOBBERS);   \
        else GCRNN_HOP_ASM_UNI16_SUMS_ASM_(GCRNN_HOP_ASM_UNI16_SUMS_SPARSE_TEXT, GCRNN_HOP_ASM_UNI16_SUMS_SPARSE_CLOBBERS);                     \
      } else {                                                                                     \
        const int hit_ = ((r >> 3) == (q & 1)) ? (r & 7) : 8;     /* position of this lane's 1.0 among its 8 A elements, or none */ \
        const uint32_t one_ = (hit_ & 1) ? 0x3f800000u : 0x00003f80u;                              \
        u32x4a_ aop_ = {(hit_ >> 1) == 0 ? one_ : 0u, (hit_ >> 1) == 1 ? one_ : 0u, (hit_ >> 1) == 2 ? one_ : 0u, (hit_ >> 1) == 3 ? one_ : 0u}; \
        if constexpr (IMGB_) GCRNN_HOP_ASM_UNI16_SUMS_ASM_(GCRNN_HOP_ASM_UNI16_SUMS_TEXT_B, GCRNN_HOP_ASM_UNI16_SUMS_CLOBBERS, , "v"(aop_));     \
        else GCRNN_HOP_ASM_UNI16_SUMS_ASM_(GCRNN_HOP_ASM_UNI16_SUMS_TEXT, GCRNN_HOP_ASM_UNI16_SUMS_CLOBBERS, , "v"(aop_));                       \
      }                                                                                            \
    }                                                                                              \
  } while (0)

// UNI (RESIDENT only): uniform-weight graph image -- column words only, all non-zeros weigh uni_w (GCRNN_HOP_ASM_UNI_STREAM)
// UNI == 2: the same on a bf16 image of the hop state with matrix-core sums (GCRNN_HOP_ASM_UNI16_STREAM; plan arrays of graph.fused_plan(img16=True))
template <int K, int HS, int XS, bool GATED, bool RESIDENT, int EPI = 0, int UNI = 0>
__global__ __launch_bounds__(STHREADS) void fused_step_kernel(
    const uint16_t* __restrict__ xt,        // [B][NP][G]   bf16
    const uint16_t* __restrict__ hprev,     // [B][NP][F]   bf16
    uint16_t* __restrict__ hout,            // [B][NP][F]   bf16
    const uint4* __restrict__ wpack,        // [F/16][K][KS][64] x 16 B
    const float* __restrict__ bias,         // [F] or null
    const float* __restrict__ gi,           // [B] (GATED)
    const float* __restrict__ gf,           // [B] (GATED); EPI 2: forget gate of the step being back-propagated, or null
    const int32_t* __restrict__ tile_nodes, // [NP] node id of each tile slot
    const int32_t* __restrict__ tile_off,   // [NP/16 + 1], in entries
    const int32_t* __restrict__ ell_col,    // [entries][16] neighbour node id            (used when !RESIDENT)
    const float* __restrict__ ell_val,      // [entries][16]
    const float4* __restrict__ ell_val4,    // [entries/4][16] x 4 weights                  (LDS image, RESIDENT)
    const uint2* __restrict__ ell_col4,     // [entries/4][16] x 4 u16 (row offset | swizzle)
    const float* __restrict__ gate_w,       // GATEOUT: [N][F] node-major weights of the gate's Linear(N*F -> 1)
    float* __restrict__ gate_out,           // GATEOUT: [B][F/16][8] per-(chunk, wave) partials of sum_{n,f} tanh(pre) * gate_w
    const uint16_t* __restrict__ aux0,      // EPI 2: upstream gradient dH_{t-1} [B][NP][F] bf16 (or null); EPI 3: dpre [B][NP][F] bf16
    const uint16_t* __restrict__ aux1,      // EPI 2: state h_{t-1} [B][NP][F] bf16;  EPI 0: user-layout output H[.][t][F][N] (or null)
    int ubstride,                           // EPI 0: elements between consecutive sequences of the user-layout output (T*F*N)
    int entries, int B, int hmod, int N,
    const int32_t* __restrict__ flags,     // EPI 1 (or null): flags[0] != 0 = the state operand h0 is all zeros -> its loads and MFMAs are skipped
    float uni_w,                           // UNI: the one weight of every non-zero
    const uint16_t* __restrict__ pk_src,   // UNI, EPI 0 / EPI 2, inline pack of the NEXT launch's operand (or null): its USER-layout block of sequence 0
                                           // (EPI 0: x_{t+1} = X[0][t+1], G rows of N; EPI 2: dH_{t-2} = dH[0][t-2], F rows of N), ...
    uint16_t* __restrict__ pk_dst,         // ... the sequence-major array it is laid out into here ([B][NP][rows]) ...
    int pk_stride) {                       // ... and the elements between consecutive sequences of the user-layout tensor
  static_assert(!UNI || (RESIDENT && GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8), "the uniform-weight stream is the asm stream on the resident graph");
  constexpr int KS = HS + XS;
  constexpr int F = 32 * HS, G = 32 * XS;
  constexpr int NCH = F / FC;
  constexpr bool GATEOUT = (EPI == 1 || EPI == 3);      // per-item scalar outputs (partials per chunk and wave)
  // rows of the operand an inline pack lays out for the next launch (0: this instantiation has none)
  constexpr int PKROWS = (UNI != 0 && (EPI == 0 || EPI == 6) && XS > 0 && !GATED) ? G : ((UNI != 0 && EPI == 2 && XS == 0) ? F : 0);   // (the gated cell's pre-passes need every x_t up front)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* state = reinterpret_cast<float*>(smem);
  uint4* wl = reinterpret_cast<uint4*>(smem + NP * FC * 4);
  // resident graph: per group of 4 entries and tile slot r:  lval4[g][r] = 4 weights, lcol4[g][r] = 4 x u16 = (col * 64)
  float4* lval4 = reinterpret_cast<float4*>(smem + NP * FC * 4 + K * KS * 1024);
  uint2* lcol4 = reinterpret_cast<uint2*>(lval4 + ((RESIDENT && !UNI) ? entries * 4 : 0));

  // XCD-aware placement: the NCH chunk workgroups of one sequence get block ids that are equal mod 8,
  // i.e. one XCD under round-robin dispatch (speed only; nothing depends on it).
  // Each workgroup stays on its CU for the whole launch and walks the sequences b0, b0 + SEQ_SLOTS, ...: weights and
  // graph are staged into LDS once per launch, not once per sequence.
  const int L = blockIdx.x;
  const int grp = L / (8 * NCH), rem = L - grp * (8 * NCH);
  const int chunk = rem >> 3, b0 = grp * 8 + (rem & 7);
  const int seq_slots = (gridDim.x / (8 * NCH)) * 8;
  if (b0 >= B) return;

  constexpr int HT = STILES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform -> scalar loads below
  const int r = lane & 15, q = lane >> 4;

#ifdef GCRNN_STAGGER_US      // experiment (tools/stagger_ab.sh): every second group of workgroups starts late, so that the chip's CUs are not all in the same phase
  if (grp & 1) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)(GCRNN_STAGGER_US) * 100u) __builtin_amdgcn_s_sleep(8);
  }
#endif
  for (int i = tid; i < K * KS * 64; i += STHREADS) wl[i] = wpack[(int64_t)chunk * K * KS * 64 + i];
  if (RESIDENT) {
    const int n = (entries >> 2) * 16;                         // the host packed the LDS image: straight copies,
    for (int i0 = 0; i0 < n; i0 += STHREADS * 4) {                  // 4 loads in flight per lane before the first LDS store
      const int i_0 = i0 + tid, i_1 = i_0 + STHREADS, i_2 = i_0 + 2 * STHREADS, i_3 = i_0 + 3 * STHREADS, nl = n - 1;
      const uint2 tc0 = ell_col4[i_0 < n ? i_0 : nl], tc1 = ell_col4[i_1 < n ? i_1 : nl];
      const uint2 tc2 = ell_col4[i_2 < n ? i_2 : nl], tc3 = ell_col4[i_3 < n ? i_3 : nl];
      if (!UNI) {
        const float4 tv0 = ell_val4[i_0 < n ? i_0 : nl], tv1 = ell_val4[i_1 < n ? i_1 : nl];
        const float4 tv2 = ell_val4[i_2 < n ? i_2 : nl], tv3 = ell_val4[i_3 < n ? i_3 : nl];
        if (i_0 < n) lval4[i_0] = tv0;
        if (i_1 < n) lval4[i_1] = tv1;
        if (i_2 < n) lval4[i_2] = tv2;
        if (i_3 < n) lval4[i_3] = tv3;
      }
      if (i_0 < n) lcol4[i_0] = tc0;
      if (i_1 < n) lcol4[i_1] = tc1;
      if (i_2 < n) lcol4[i_2] = tc2;
      if (i_3 < n) lcol4[i_3] = tc3;
    }
  }
  // per-wave tile ranges, fetched once through the scalar path
  int tbeg[STILES], tend[STILES];
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
    tbeg[i] = tile_off[wave * STILES + i];
    tend[i] = tile_off[wave * STILES + i + 1];
  }
  f32x4 u[STILES][K - 1];   // taps 0..K-2 (tap K-1 seeds the LDS state directly); later: the hop results
  int woff[STILES];      // low 16 bits: byte offset of this lane's quad in the (swizzled) state row of its node; high: node id
#pragma unroll
  for (int i = 0; i < STILES; ++i)      // slot = node << 16 | row << 6 | swz << 4;  UNI == 2 (bf16 image): node << 16 | row << 5 | hswz << 4, a lane's 8 bytes = half q >> 1, piece q & 1
    woff[i] = tile_nodes[(wave * STILES + i) * 16 + r] ^ (UNI == 2 ? (((q >> 1) << 4) | ((q & 1) << 3)) : (q << 4));
  float bvec[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) bvec[c] = bias[chunk * FC + q * 4 + c];
  }
  __syncthreads();

  // Row traffic goes through buffer instructions: descriptor in SGPRs (built from kernel arguments only, so provably
  // wave-uniform), 32-bit lane offset, per-sequence base as the scalar offset -- no per-lane 64-bit pointers to keep
  // alive (and spill) across the sequence loop.
  const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(hprev), 0, hmod * (NP * F * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xt), 0, B * (NP * G * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(hout, 0, hout ? B * (NP * F * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(aux0), 0, ((EPI == 2 || EPI == 3 || EPI == 5) && aux0) ? B * (NP * F * 2) : 0, 0x00020000);
  // EPI 5 (XS = 0: no input operand): the xt argument carries the optional Yh output [B][NP][F] instead
  // EPI 2 on a state-only operand: xt is the optional second output of the node-gated BPTT (the next launch's operand)
  const __amdgpu_buffer_rsrc_t rsrc_yh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xt), 0, ((EPI == 5 || (EPI == 2 && XS == 0)) && xt) ? B * (NP * F * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(aux1), 0, (EPI == 2 && aux1) ? B * (NP * F * 2) : 0, 0x00020000);

  // gate pre-pass with an all-zero initial state (every training loop of the reference starts from h0 = 0, train_rnn.py:256): the
  // state half of the operand contributes exactly nothing -- skip its loads and MFMAs (wave-uniform)
  const bool skip_h = (EPI == 1) && flags && flags[0] != 0;
  // Cross-item fragment prefetch (GCRNN_P1_AHEAD tiles of the NEXT sequence's B operand): requested when the LAST hop of the current
  // sequence starts -- the registers of the taps already folded in are free then, the hop is LDS-bound and does not use the L2 -> CU
  // path -- so that most of phase 1's operand transfer overlaps the hops instead of preceding them.
  constexpr int PT = (GCRNN_P1_AHEAD < 0) ? 0 : (GCRNN_P1_AHEAD > STILES ? STILES : GCRNN_P1_AHEAD);
  constexpr int PTE = (EPI == 1 || EPI == 2 || EPI == 5) && KS == 4 ? (PT > 4 ? 4 : PT) : PT;      // instantiations whose epilogue prefetch also needs registers
  bf16x8 pf[PTE > 0 ? PTE : 1][KS];
  auto load_ahead = [&](int bn) {
    const int sh = (bn % hmod) * (NP * F * 2), sx = bn * (NP * G * 2);
#pragma unroll
    for (int i = 0; i < PTE; ++i) {
      int w = woff[i];
      asm volatile("" : "+v"(w));
      const int roh = (w >> 16) * (F * 2) + 16 * q, rox = (w >> 16) * (G * 2) + 16 * q;
#pragma unroll
      for (int s = 0; s < HS; ++s)
        pf[i][s] = skip_h ? __builtin_bit_cast(bf16x8, uint4{0u, 0u, 0u, 0u})
                          : __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, roh + 64 * s, sh, 0));
#pragma unroll
      for (int s = 0; s < XS; ++s)
        pf[i][HS + s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, rox + 64 * s, sx, 0));
    }
  };
  if (PTE > 0) load_ahead(b0);
  for (int b = b0; b < B; b += seq_slots) {
  const int soff_h = (b % hmod) * (NP * F * 2);     // hmod < B: every item of the gate pre-pass reads h0[b]
  const int soff_x = b * (NP * G * 2);
  float gin = 1.f, gfo = 1.f;
  float gratio = 1.f;
  if (GATED) { gin = gi[b]; gfo = gf[b]; gratio = gfo / fmaxf(gin, 1e-30f); }
  if (EPI == 5 && gi) { gin = gi[b]; gfo = gf[b]; }      // scalar time gates on top of the node gates (applied in the epilogue)

  // ---- phase 1: taps on the matrix cores ------------------------------------------------------
#ifdef GCRNN_ABLATE_PHASE1      // profiling builds only (tools/ablate.sh): results are wrong by construction
#pragma unroll
  for (int i = 0; i < STILES; ++i)
#pragma unroll
    for (int tap = 0; tap < K - 1; ++tap) u[i][tap] = f32x4{0.f, 0.f, 0.f, (float)woff[i]};
#else
  // all B-operand fragments of the wave (8 tiles x 4 x 16 B per lane) are requested before the first MFMA: one
  // memory latency per sequence instead of one per tile; the registers are free again before the taps fill up.
  bf16x8 bfr[STILES][KS];
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
    if (i < PTE) {                   // requested during the previous sequence's last hop (or before the loop)
#pragma unroll
      for (int s = 0; s < KS; ++s) bfr[i][s] = pf[i][s];
      continue;
    }
    int w = woff[i];
    asm volatile("" : "+v"(w));      // opaque per iteration: keeps hipcc from hoisting (and spilling) 16+ row offsets
    const int roh = (w >> 16) * (F * 2) + 16 * q, rox = (w >> 16) * (G * 2) + 16 * q;
#pragma unroll
#ifdef GCRNN_ABLATE_P1_LOADS
    for (int s = 0; s < KS; ++s) bfr[i][s] = __builtin_bit_cast(bf16x8, uint4{(unsigned)roh, (unsigned)rox, (unsigned)s, 1u});
#else
    for (int s = 0; s < HS; ++s)
      bfr[i][s] = skip_h ? __builtin_bit_cast(bf16x8, uint4{0u, 0u, 0u, 0u})
                         : __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, roh + 64 * s, soff_h, 0));
#pragma unroll
    for (int s = 0; s < XS; ++s)
      bfr[i][HS + s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, rox + 64 * s, soff_x, 0));
#endif
  }
#if !defined(GCRNN_ABLATE_P1_MFMA)
  // GP tiles share every weight fragment: 1/GP of the A-operand LDS reads and GP independent MFMA chains per tap
  // (GP = 2: 248 VGPRs, -3 us per launch; GP = 4 spills in some instantiations -- tools/p1pair_ab.sh)
  {
    constexpr int GP = GCRNN_P1_GROUP;
    static_assert(STILES % GP == 0, "tile groups");
#pragma unroll
    for (int i = 0; i < STILES; i += GP) {
#pragma unroll
      for (int tap = 0; tap < K; ++tap) {
        f32x4 accg[GP];
#pragma unroll
        for (int p = 0; p < GP; ++p) accg[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!skip_h) {
#pragma unroll
          for (int s = 0; s < HS; ++s) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, wl[(tap * KS + s) * 64 + lane]);
#pragma unroll
            for (int p = 0; p < GP; ++p) accg[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[i + p][s], accg[p], 0, 0, 0);
          }
        }
        // time-gated cell: gi (x W_x) + gf (h W_h) on ONE accumulator: h-chain, scale by gf/gi, continue the chain with x, scale
        // by gi (gi = sigmoid(.) > 0; the wave-uniform guard covers an underflowed gate)
        const bool xpart = !GATED || gin > 1e-30f;
        if (GATED) {
#pragma unroll
          for (int p = 0; p < GP; ++p) accg[p] *= (xpart ? gratio : gfo);
        }
        if (xpart) {
#pragma unroll
          for (int s = HS; s < KS; ++s) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, wl[(tap * KS + s) * 64 + lane]);
#pragma unroll
            for (int p = 0; p < GP; ++p) accg[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[i + p][s], accg[p], 0, 0, 0);
          }
          if (GATED) {
#pragma unroll
            for (int p = 0; p < GP; ++p) accg[p] *= gin;
          }
        }
#pragma unroll
        for (int p = 0; p < GP; ++p) {
          if (tap == K - 1) {
            int wv = woff[i + p];
            asm volatile("" : "+v"(wv));      // opaque: the masked LDS offsets are not hoisted out of the tile loop
            state_put<UNI == 2>(state, wv, accg[p]);
          } else {
            u[i + p][tap] = accg[p];
          }
        }
      }
    }
  }
#else      // profiling build (tools/ablate.sh): the MFMAs replaced by a few adds on the loaded fragments; results are wrong
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
#pragma unroll
    for (int tap = 0; tap < K; ++tap) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) { const f32x4 t = __builtin_bit_cast(f32x4, bfr[i][s]); acc += t * (float)(tap + 1); }
      if (tap == K - 1) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        state_put<UNI == 2>(state, wv, acc);
      }
      else u[i][tap] = acc;
    }
  }
#endif
#endif
  lds_barrier();

  // L2 prefetch of the NEXT sequence of this workgroup: while the hops keep the LDS busy, every thread touches one
  // 128-byte row (= one cache line) of [h | x]; the NCH chunk workgroups of a sequence share an XCD and split the
  // 2 * NP rows between them. Phase 1 of the next sequence then streams from L2 instead of stalling on HBM.
  uint32_t prefetched = 0;
#ifndef GCRNN_EPI_PREFETCH
#define GCRNN_EPI_PREFETCH 1       // 0: epilogue operands (EPI 1, 2, 3, 5) are loaded in the epilogue instead of at the start of the last hop (A/B)
#endif
#ifndef GCRNN_PREFETCH_AT
#define GCRNN_PREFETCH_AT 1      // 0: no L2 prefetch, 1: at the start of the hops (default), 2: before the last hop (A/B: tools/prefetch_ab.sh)
#endif
  auto prefetch_next = [&]() {
    if (b + seq_slots < B) {
      constexpr int LINES = (XS > 0 ? 2 : 1) * NP;                         // rows of [h | x] (x absent in the BPTT step)
      const int line = chunk * (LINES / NCH) + tid;                        // LINES / NCH == 512 for F = G = 64
      if (tid < LINES / NCH) {
        prefetched = (line < NP)
            ? __builtin_amdgcn_raw_buffer_load_b32(rsrc_h, line * (F * 2), ((b + seq_slots) % hmod) * (NP * F * 2), 0)
            : __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, (line - NP) * (G * 2), (b + seq_slots) * (NP * G * 2), 0);
      }
    }
  };
  if (GCRNN_PREFETCH_AT == 1 || (GCRNN_PREFETCH_AT == 2 && K <= 2)) prefetch_next();
  // EPI 2 (BPTT step): the epilogue's operands h_{t-1} (last touched a whole forward ago) and dH_{t-1} come cold from HBM, and every
  // CU asks for them at the same moment (the start of its last hop): 2 x 32.8 MB in one hop's time at B = 256. Touch their lines
  // now -- the chunk workgroups of a sequence split them, one line per thread -- so that the register prefetch below hits L2.
#ifndef GCRNN_EPI_L2_PREFETCH
#define GCRNN_EPI_L2_PREFETCH 1      // A/B switch
#endif
  // Inline pack: the rows of the user layout that the LDS-DMA of the last hop will fetch (x_{t+1}[b] / dH_{t-2}[b], never touched
  // before) are cold as well: touch this workgroup's 512 bytes of every row now (128-byte steps plus the last dword of the range).
  uint32_t prefetched_pk = 0;
  if constexpr (PKROWS > 0 && GCRNN_EPI_L2_PREFETCH) {
    constexpr int NPC = NP / NCH;
    const int prow = tid >> 3, pj = tid & 7;
    if (pk_src && prow < PKROWS && pj < 5 && chunk * NPC < N) {
      int node = chunk * NPC + (pj < 4 ? pj * 64 : NPC - 2);
      node = node < N - 2 ? node : N - 2;                                     // N % 8 == 0 on this path: an even element index, inside the row
      prefetched_pk = *reinterpret_cast<const uint32_t*>(pk_src + (int64_t)b * pk_stride + (int64_t)prow * N + node);
    }
  }
  uint32_t prefetched_epi = 0;
  if constexpr ((EPI == 2 || EPI == 5) && GCRNN_EPI_L2_PREFETCH) {          // EPI 5 (node-gated step): Yx_t of the all-steps pass (rsrc_a0)
    constexpr int ELINES = NP * F * 2 / 128 / NCH;                          // 128-byte lines of one operand per chunk workgroup
    const int idx = tid < ELINES ? tid : tid - ELINES;
    if (tid < 2 * ELINES) {
      const int eo = (chunk * ELINES + idx) * 128;
      if (EPI == 5) {
        if (tid >= ELINES) prefetched_epi = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a0, eo, b * (NP * F * 2), 0);
      } else {
        prefetched_epi = tid < ELINES ? __builtin_amdgcn_raw_buffer_load_b32(rsrc_a1, eo, b * (NP * F * 2), 0)      // zero-length descriptors when absent
                                      : __builtin_amdgcn_raw_buffer_load_b32(rsrc_a0, eo, b * (NP * F * 2), 0);
      }
    }
  }

  // ---- phase 2: Horner hops, state image in LDS -------------------------------------------------
  const char* sbytes = reinterpret_cast<const char*>(state);
  const int qoff = q * 16;
  // 32-bit LDS byte addresses for the asm reads (low half of the flat LDS address = offset in the allocation)
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
#if GCRNN_HOP_ASM
  if (lds0 != 0) __builtin_trap();        // the asm stream forms gather addresses as (column word ^ q << 4): the state image must sit at LDS address 0
#endif
  const uint32_t qx = (uint32_t)qoff;     // stored column = (col << 6) | (swizzle << 4);  ^ (q << 4) selects this lane's quad
  const uint32_t lds_val = lds0 + NP * FC * 4 + K * KS * 1024;
  const uint32_t lds_col = lds_val + ((RESIDENT && !UNI) ? entries * 64 : 0);
#ifdef GCRNN_ABLATE_HOPS
#define GCRNN_HOP_FIRST K
#else
#define GCRNN_HOP_FIRST 1
#endif
  // EPI 2: the epilogue's operands (h_{t-1} for tanh', the upstream gradient dH_{t-1}) are requested when the LAST hop starts -- by
  // then the registers of the taps already folded in are free -- and land while it runs (their latency used to sit in the epilogue)
  u32x2 eph[(EPI == 2 && GCRNN_EPI_PREFETCH) ? STILES : 1], epg[((EPI == 2 || EPI == 5) && GCRNN_EPI_PREFETCH) ? STILES : 1];
  float epn[(EPI == 5 && GCRNN_EPI_PREFETCH) ? STILES : 1][2];      // EPI 5: the node's input / forget gate (Yx_t goes through epg)
  float4 epw[(EPI == 1 && GCRNN_EPI_PREFETCH) ? STILES : 1];        // EPI 1: the gate read-out's weights of this lane's (node, 4 features)
  u32x2 epd[(EPI == 3 && GCRNN_EPI_PREFETCH) ? STILES : 1];         // EPI 3: dpre of this lane's (node, 4 features)
#pragma unroll
  for (int j = GCRNN_HOP_FIRST; j < K; ++j) {
    if (GCRNN_PREFETCH_AT == 2 && K > 2 && j == K - 1) prefetch_next();
    if constexpr (PTE > 0) {
      if (j == K - 1 && b + seq_slots < B) load_ahead(b + seq_slots);
    }
    if constexpr (EPI == 2 && GCRNN_EPI_PREFETCH) {
      if (j == K - 1) {
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int eoff = (wv >> 16) * (F * 2) + (chunk * FC + q * 4) * 2;
          eph[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_a1, eoff, b * (NP * F * 2), 0);      // zero-length descriptor when aux1 is null: 0
          epg[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, eoff, b * (NP * F * 2), 0);
        }
      }
    }
    if constexpr ((EPI == 1 || EPI == 3) && GCRNN_EPI_PREFETCH) {
      if (j == K - 1) {
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int node = wv >> 16;
          if constexpr (EPI == 1) {
            epw[i] = node < N ? *reinterpret_cast<const float4*>(gate_w + (int64_t)node * F + chunk * FC + q * 4) : float4{0.f, 0.f, 0.f, 0.f};
          } else {
            epd[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
          }
        }
      }
    }
    if constexpr (EPI == 5 && GCRNN_EPI_PREFETCH) {
      if (j == K - 1) {
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int node = wv >> 16;
          epg[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
          epn[i][0] = node < N ? gate_w[(int64_t)b * N + node] : 0.f;
          epn[i][1] = node < N ? gate_w[(int64_t)(B + b) * N + node] : 0.f;
        }
      }
    }
    if constexpr (PKROWS > 0) {
      // Inline pack (uniform graphs leave LDS room next to the column image): this workgroup's node range [chunk NPC, +NPC) of
      // the next launch's operand (EPI 0: x_{t+1}[b]; EPI 2: the upstream gradient dH_{t-2}[b]), all feature rows of the USER layout,
      // is requested by LDS-DMA (no registers) when the LAST hop starts and lands while that hop keeps the LDS busy; after the
      // epilogue it is read back transposed and stored sequence-major -- the separate pack pass over X (dH) disappears (only the
      // first step(s) are packed by the caller).
      if (j == K - 1 && pk_src) {
        constexpr int NPC = NP / NCH, PPR = NPC / 8, PIECES = PKROWS * PPR;
        static_assert(PIECES % STHREADS == 0, "whole pieces per thread");
        char* xtile = smem + NP * FC * 4 + K * KS * 1024 + entries * 32;
        const uint16_t* xsrc = pk_src + (int64_t)b * pk_stride + chunk * NPC;
#pragma unroll
        for (int i = 0; i < PIECES / STHREADS; ++i) {
          // LDS slot id = (row, cs) receives the row's 8-node piece col = (cs - (row >> 3)) mod PPR: every group of 8 feature rows is
          // rotated by one more 16-byte slot, so that the read-back below (8 lanes = the 8 feature groups of one node) is conflict-free
          const int id = i * STHREADS + tid;
          const int row = id / PPR, cs = id - row * PPR;
          const int col = (cs - (row >> 3)) & (PPR - 1);
          if (chunk * NPC + col * 8 < N)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + (int64_t)row * N + col * 8),
                                             (__attribute__((address_space(3))) void*)(xtile + (i * STHREADS + wave * 64) * 16), 16, 0, 0);
        }
      }
    }
    if (RESIDENT) {
#define GCRNN_FWD_INIT(i) u[i][K - 1 - j]
#define GCRNN_FWD_STORE(i, a) u[i][K - 1 - j] = a   /* the new value lives in the tap's registers until every wave has read `state` */
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
      if (UNI == 2) GCRNN_HOP_ASM_UNI16_STREAM(GCRNN_FWD_INIT, GCRNN_FWD_STORE);
      else if (UNI) GCRNN_HOP_ASM_UNI_STREAM(GCRNN_FWD_INIT, GCRNN_FWD_STORE);
      else GCRNN_HOP_ASM_STREAM(GCRNN_FWD_INIT, GCRNN_FWD_STORE);
#else
      GCRNN_HOP_STREAM(GCRNN_FWD_INIT, GCRNN_FWD_STORE);
#endif
#undef GCRNN_FWD_INIT
#undef GCRNN_FWD_STORE
    } else {
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        const int beg = tbeg[i], end = tend[i];
        f32x4 acc = u[i][K - 1 - j];
        for (int e = beg; e < end; e += 4) {      // entry counts are padded to multiples of 4
          int cc[4]; float vv[4]; f32x4 xv[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) { cc[p] = ell_col[(e + p) * 16 + r]; vv[p] = ell_val[(e + p) * 16 + r]; }
#pragma unroll
          for (int p = 0; p < 4; ++p)
            xv[p] = *reinterpret_cast<const f32x4*>(sbytes + (cc[p] ^ qoff));      // ell_col = node_addr of the neighbour
#pragma unroll
          for (int p = 0; p < 4; ++p) acc += vv[p] * xv[p];
        }
        u[i][K - 1 - j] = acc;
      }
    }
    if (j < K - 1) {
      lds_barrier();
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        state_put<UNI == 2>(state, wv, u[i][K - 1 - j]);
      }
      lds_barrier();
    }
  }

  if constexpr (PKROWS > 0) {
    // inline pack: this wave's LDS-DMA pieces have had the whole last hop to land; wait for them HERE, before any barrier of the
    // epilogue -- hipcc only waits (vmcnt) before a wave's OWN aliasing LDS reads, i.e. after the barrier that is supposed to
    // publish the pieces to the other waves
    if (pk_src) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- epilogue: bias, tanh, bf16 store into the node-major state h_t ------------------------------
  float bsum[4];
  {
    const float bs = gin + gfo;     // the one bias is added by both filters (graphML.py:2420-2421)
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[c] = bs * bvec[c];
  }
  if (EPI == 3) {
    // gate-gradient pass: the hops produced the filter output of this item's chunk; its inner product with dpre (one
    // bias: this is ONE filter, A(S)x + b or B(S)h + b) is the chunk's share of d loss / d gate (graphML.py:2420-2421)
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < STILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      const int node = wv >> 16;
      const u32x2 d2 = GCRNN_EPI_PREFETCH ? epd[GCRNN_EPI_PREFETCH ? i : 0]
                                            : __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
      const f32x4 acc = u[i][0];
      if (node < N)
        part += (acc[0] + bvec[0]) * bf2f((uint16_t)(d2[0] & 0xffffu)) + (acc[1] + bvec[1]) * bf2f((uint16_t)(d2[0] >> 16)) +
                (acc[2] + bvec[2]) * bf2f((uint16_t)(d2[1] & 0xffffu)) + (acc[3] + bvec[3]) * bf2f((uint16_t)(d2[1] >> 16));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) gate_out[(int64_t)b * (NCH * SWAVES) + chunk * SWAVES + wave] = part;   // fixed-order sum by the caller
  } else if (EPI == 4) {
    // filter-output pass: the hops produced this item's chunk of A(S)x_t (or any one filter); + b, bf16, sequence-major
#pragma unroll
    for (int i = 0; i < STILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      const int node = wv >> 16;
      uint2 pk{0u, 0u};
      if (node < N) {
        const f32x4 acc = u[i][0];
        pk.x = pack2bf(acc[0] + bvec[0], acc[1] + bvec[1]);
        pk.y = pack2bf(acc[2] + bvec[2], acc[3] + bvec[3]);
      }
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk.x, pk.y}, rsrc_o, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
    }
  } else if (GATEOUT) {
    // gate pre-pass: partial dot product of tanh(pre) with the gate's linear weights over this chunk, one partial per wave;
    // with hout the gate cell's state c = tanh(pre) is also stored (bf16, sequence-major) for the gate's BPTT
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < STILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      const int node = wv >> 16;
      uint2 pk{0u, 0u};
      if (node < N) {
        const float4 w4 = GCRNN_EPI_PREFETCH ? epw[GCRNN_EPI_PREFETCH ? i : 0] : *reinterpret_cast<const float4*>(gate_w + (int64_t)node * F + chunk * FC + q * 4);
        const f32x4 acc = u[i][0];
        const float o0 = fast_tanh(acc[0] + bsum[0]), o1 = fast_tanh(acc[1] + bsum[1]);
        const float o2 = fast_tanh(acc[2] + bsum[2]), o3 = fast_tanh(acc[3] + bsum[3]);
        part = __builtin_fmaf(o3, w4.w, __builtin_fmaf(o2, w4.z, __builtin_fmaf(o1, w4.y, __builtin_fmaf(o0, w4.x, part))));      // (explicit chain, as the chain's partials)
        pk.x = pack2bf(o0, o1);
        pk.y = pack2bf(o2, o3);
      }
      if (hout) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk.x, pk.y}, rsrc_o, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) gate_out[(int64_t)b * (NCH * SWAVES) + chunk * SWAVES + wave] = part;   // fixed-order sum by the caller
  } else if (EPI == 2) {
    // BPTT data-gradient step: the hops just applied sum_k (S)^k (dpre_t W_k) = d h_{t-1} (recurrent part); add the
    // upstream gradient of h_{t-1} and go through tanh':  dpre_{t-1} = (acc + dH_{t-1}) * (1 - h_{t-1}^2).
    // With aux0 == null the raw state gradient is stored (d h0). Time-gated cell: the recurrent part carries the forget
    // gate of the step it came through, gf_t[b] (the adjoint chain is linear, so the scale is applied here).
    const float gsc = gf ? gf[b] : 1.f;
    // With gate_out the launch also emits <h_{t-1}, sum_k (S)^k (dpre_t B_k)> = <B(S) h_{t-1}, dpre_t> (adjoint identity): the
    // bias-free part of d loss / d gf_t -- the forget gate's gradient without a pass of its own.
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < STILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      const int node = wv >> 16;
      const int eoff = node * (F * 2) + (chunk * FC + q * 4) * 2;
      const f32x4 raw = u[i][0];
      f32x4 o = raw * gsc;
      float hv0 = 0.f, hv1 = 0.f, hv2 = 0.f, hv3 = 0.f;
      if (aux1) {
        const u32x2 h2 = (GCRNN_EPI_PREFETCH && K > 1) ? eph[GCRNN_EPI_PREFETCH ? i : 0] : __builtin_amdgcn_raw_buffer_load_b64(rsrc_a1, eoff, b * (NP * F * 2), 0);
        hv0 = bf2f((uint16_t)(h2[0] & 0xffffu)); hv1 = bf2f((uint16_t)(h2[0] >> 16));
        hv2 = bf2f((uint16_t)(h2[1] & 0xffffu)); hv3 = bf2f((uint16_t)(h2[1] >> 16));
      }
      if (gate_out) part = __builtin_fmaf(raw[3], hv3, __builtin_fmaf(raw[2], hv2, __builtin_fmaf(raw[1], hv1, __builtin_fmaf(raw[0], hv0, part))));      // (explicit chain: with -ffp-contract=fast the association of a*b + c*d + .. is the compiler's choice, per instantiation)     // rows >= N of h are zero
      if (aux0) {
        const u32x2 g2 = (GCRNN_EPI_PREFETCH && K > 1) ? epg[GCRNN_EPI_PREFETCH ? i : 0] : __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, eoff, b * (NP * F * 2), 0);
        const float g0 = bf2f((uint16_t)(g2[0] & 0xffffu)), g1 = bf2f((uint16_t)(g2[0] >> 16));
        const float g2f = bf2f((uint16_t)(g2[1] & 0xffffu)), g3 = bf2f((uint16_t)(g2[1] >> 16));
        o[0] = (o[0] + g0) * (1.f - hv0 * hv0);
        o[1] = (o[1] + g1) * (1.f - hv1 * hv1);
        o[2] = (o[2] + g2f) * (1.f - hv2 * hv2);
        o[3] = (o[3] + g3) * (1.f - hv3 * hv3);
      }
      uint2 pk;
      if (node < N) {
        pk.x = pack2bf(o[0], o[1]);
        pk.y = pack2bf(o[2], o[3]);
      } else {
        pk.x = 0u; pk.y = 0u;
      }
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk.x, pk.y}, rsrc_o, eoff, b * (NP * F * 2), 0);     // dropped when hout is null
      if (XS == 0 && xt) {
        // node-gated cell: the operand of the NEXT launch of the chain is d(B(S)h + b) = (gf nf)[n] . dpre, scaled per node here
        // (a per-node scale does not commute with the graph shifts, so it cannot wait for the epilogue of that launch)
        uint2 pg{0u, 0u};
        if (node < N) {
          const float gn = gate_w[(int64_t)b * N + node];
          pg.x = pack2bf(o[0] * gn, o[1] * gn);
          pg.y = pack2bf(o[2] * gn, o[3] * gn);
        }
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pg.x, pg.y}, rsrc_yh, eoff, b * (NP * F * 2), 0);
      }
    }
    if (gate_out) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
      if (lane == 0) gate_out[(int64_t)b * (NCH * SWAVES) + chunk * SWAVES + wave] = part;   // fixed-order sum by the caller
    }
  } else {
  float hp[STILES];                                    // EPI 6 (state epilogue + fused output head): this lane's per-tile partials ...
  float hw[4] = {0.f, 0.f, 0.f, 0.f};                  // ... and its four weights (gate_w = [F]), fetched per item (EPI 6 is an
  if constexpr (EPI == 6) {                            // instantiation of its own: the plain state epilogue keeps its register allocation)
#pragma unroll
    for (int c = 0; c < 4; ++c) hw[c] = gate_w[chunk * FC + q * 4 + c];
  }
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
    int wv = woff[i];
    asm volatile("" : "+v"(wv));
    const int node = wv >> 16;
    const f32x4 acc = u[i][0];
    uint2 pk;
    if (node < N) {
      float o0, o1, o2, o3;
      if (EPI == 5) {
        // node-gated cell: the x part comes from the all-steps pass, both parts are scaled per node (and per sequence)
        const int eoff = node * (F * 2) + (chunk * FC + q * 4) * 2;
        const u32x2 y2 = GCRNN_EPI_PREFETCH ? epg[GCRNN_EPI_PREFETCH ? i : 0] : __builtin_amdgcn_raw_buffer_load_b64(rsrc_a0, eoff, b * (NP * F * 2), 0);
        const float ni = gin * (GCRNN_EPI_PREFETCH ? epn[GCRNN_EPI_PREFETCH ? i : 0][0] : gate_w[(int64_t)b * N + node]);
        const float nf = gfo * (GCRNN_EPI_PREFETCH ? epn[GCRNN_EPI_PREFETCH ? i : 0][1] : gate_w[(int64_t)(B + b) * N + node]);
        const float yh0 = acc[0] + bvec[0], yh1 = acc[1] + bvec[1], yh2 = acc[2] + bvec[2], yh3 = acc[3] + bvec[3];
        if (xt) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2bf(yh0, yh1), pack2bf(yh2, yh3)},
                                                     rsrc_yh, eoff, b * (NP * F * 2), 0);
        o0 = fast_tanh(ni * bf2f((uint16_t)(y2[0] & 0xffffu)) + nf * yh0);
        o1 = fast_tanh(ni * bf2f((uint16_t)(y2[0] >> 16)) + nf * yh1);
        o2 = fast_tanh(ni * bf2f((uint16_t)(y2[1] & 0xffffu)) + nf * yh2);
        o3 = fast_tanh(ni * bf2f((uint16_t)(y2[1] >> 16)) + nf * yh3);
      } else {
        o0 = fast_tanh(acc[0] + bsum[0]); o1 = fast_tanh(acc[1] + bsum[1]);
        o2 = fast_tanh(acc[2] + bsum[2]); o3 = fast_tanh(acc[3] + bsum[3]);
      }
      pk.x = pack2bf(o0, o1);
      pk.y = pack2bf(o2, o3);
    } else {
      pk.x = 0u; pk.y = 0u;          // padded rows stay zero
      if (EPI == 5 && xt) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, rsrc_yh, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), 0);
    }
    // GCRNN_STORE_POLICY: an experiment with write-through stores (sc1: the line is not kept in the XCD's L2, whose 4 MiB the
    // [h | x] operands of the sequences in flight and the prefetched next ones need) -- slower than plain stores, see the define
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk.x, pk.y}, rsrc_o, node * (F * 2) + (chunk * FC + q * 4) * 2, b * (NP * F * 2), GCRNN_STORE_POLICY);
    u[i][0] = f32x4{__uint_as_float(pk.x), __uint_as_float(pk.y), 0.f, 0.f};      // keep the packed bf16 for the user-layout copy
    if constexpr (EPI == 6) {
      // Output head fused onto the h_t store (reference architectures.py:1616-1627, `multipMlp` with one output: the same
      // Linear(F -> 1) on every node): this chunk's share  sum_{f in chunk} w[f] h_t[n][f]  on the bf16-rounded state, summed over the four
      // feature quads of the node (lanes r, r + 16, r + 32, r + 48); the caller adds the F / 16 chunk partials and the bias. With it the
      // user-layout copy of H can be dropped altogether (aux1 = null): inference of the regression model never materialises H.
      float part = hw[0] * bf2f((uint16_t)(pk.x & 0xffffu)) + hw[1] * bf2f((uint16_t)(pk.x >> 16)) +
                   hw[2] * bf2f((uint16_t)(pk.y & 0xffffu)) + hw[3] * bf2f((uint16_t)(pk.y >> 16));
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      hp[i] = part;
    }
  }
  if constexpr (EPI == 6) {
    // the tile nodes are degree-ranked, i.e. scattered: stage the per-node partials in the (now dead) state region and store them
    // in node order, coalesced
    float* hstage = state;
    lds_barrier();                                   // every wave has finished reading `state` in the last hop
    if (q == 0) {
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        hstage[wv >> 16] = hp[i];
      }
    }
    lds_barrier();
    float* go = gate_out + ((int64_t)b * NCH + chunk) * N;
    for (int n = tid; n < N; n += STHREADS) go[n] = hstage[n];
  }
  if ((EPI == 0 || EPI == 5) && aux1) {
    // the state is also delivered in the USER layout H[b][t][f][:] (node-contiguous rows): transposed bf16 tile in LDS
    // (row stride 2080 B), then 16-byte coalesced row stores -- replaces a separate unpack pass over the whole sequence.
    constexpr int RS = 2 * NP + 32;
    char* tst = reinterpret_cast<char*>(state);
    lds_barrier();                                   // every wave has finished reading `state` in the last hop
#pragma unroll
    for (int i = 0; i < STILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      const int node = wv >> 16;
      const uint32_t p0 = __float_as_uint(u[i][0][0]), p1 = __float_as_uint(u[i][0][1]);
      // odd quads write their rows in the order 2, 3, 0, 1: RS = 8 banks mod 32, so the two quads of a 32-lane store group would
      // otherwise hit one bank for one node in every instruction; with the rotation they sit 16 banks apart, and a tile whose nodes
      // differ in (node >> 1) & 15 (graph.spread_tile_classes) scatters conflict-free
      const uint32_t pa = (q & 1) ? p1 : p0, pb = (q & 1) ? p0 : p1;
      char* ra = tst + (q * 4 + ((q & 1) ? 2 : 0)) * RS + node * 2;
      char* rb = tst + (q * 4 + ((q & 1) ? 0 : 2)) * RS + node * 2;
      *reinterpret_cast<uint16_t*>(ra) = (uint16_t)(pa & 0xffffu);
      *reinterpret_cast<uint16_t*>(ra + RS) = (uint16_t)(pa >> 16);
      *reinterpret_cast<uint16_t*>(rb) = (uint16_t)(pb & 0xffffu);
      *reinterpret_cast<uint16_t*>(rb + RS) = (uint16_t)(pb >> 16);
    }
    lds_barrier();
    const int segs = N >> 3;                           // 16-byte segments per row (N % 8 == 0 checked by the host)
    uint16_t* ub = const_cast<uint16_t*>(aux1) + (int64_t)b * ubstride + (int64_t)(chunk * FC) * N;
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(ub, 0, FC * N * 2, 0x00020000);      // this item's 16 rows of H[b][t]
    for (int idx = tid; idx < FC * segs; idx += STHREADS) {
      const int f = idx / segs, sg = idx - f * segs;
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(tst + f * RS + sg * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_u, (f * N + sg * 8) * 2, 0, GCRNN_STORE_POLICY);
    }
  }
  }
  if constexpr (PKROWS > 0) {
    if (pk_src) {
      // second half of the inline pack: the [rows][NPC] tile (every wave waited for its DMA pieces before the epilogue) -> rows of
      // 8-feature pieces; the pieces of a node sit in consecutive lanes (whole rows per store), rows >= N are zeros
      constexpr int NPC = NP / NCH, PCS = PKROWS / 8;
      const char* xtile = smem + NP * FC * 4 + K * KS * 1024 + entries * 32;
      if (!((EPI == 0 && aux1) || EPI == 6)) lds_barrier();   // (EPI 0 with the user-layout output / EPI 6: their two barriers have already passed)
      const __amdgpu_buffer_rsrc_t rsrc_xn = __builtin_amdgcn_make_buffer_rsrc(pk_dst, 0, B * (NP * PKROWS * 2), 0x00020000);
      // gfx950: a 16-byte buffer store whose soffset is an SGPR followed IMMEDIATELY by a VALU write of its first data register
      // loses that dword in a few lanes, rarely (hipcc models no hazard for this form and happily reuses one register tuple for
      // consecutive stores). So: all LDS reads first; the RI pieces are built in RI DISTINCT tuples that stay reserved until every
      // store has retired (s_waitcnt vmcnt(0) below); the stores take an immediate soffset (the form hipcc does pad).
      constexpr int RI = PCS * NPC / STHREADS;
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;
      u32x4_t vv[RI];
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int id = i * STHREADS + tid;
        const int nl = id / PCS, pc = id - nl * PCS;
        const char* src = xtile + (pc * 8) * (NPC * 2) + ((nl + 8 * pc) & (NPC - 1)) * 2;      // rows 8 pc .. 8 pc + 7 share the rotation
        uint32_t w4[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const uint32_t lo = *reinterpret_cast<const uint16_t*>(src + (2 * jj) * (NPC * 2));
          const uint32_t hi = *reinterpret_cast<const uint16_t*>(src + (2 * jj + 1) * (NPC * 2));
          w4[jj] = lo | (hi << 16);
        }
        const bool ok = chunk * NPC + nl < N;
        vv[i] = u32x4_t{ok ? w4[0] : 0u, ok ? w4[1] : 0u, ok ? w4[2] : 0u, ok ? w4[3] : 0u};
      }
#pragma unroll
      for (int i = 0; i < RI; ++i) asm volatile("" : "+v"(vv[i]));          // every piece materialised in its own tuple before the first store
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int id = i * STHREADS + tid;
        const int nl = id / PCS, pc = id - nl * PCS;
        __builtin_amdgcn_raw_buffer_store_b128(vv[i], rsrc_xn, (chunk * NPC + nl) * (PKROWS * 2) + pc * 16 + b * (NP * PKROWS * 2), 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(vv[0]), "v"(vv[RI - 1]) : "memory");      // the tuples are released only after the stores retired
#pragma unroll
      for (int i = 1; i + 1 < RI; ++i) asm volatile("" ::"v"(vv[i]));
    }
  }
  asm volatile("" ::"v"(prefetched), "v"(prefetched_epi), "v"(prefetched_pk));      // the prefetch loads retire here at the latest
  lds_barrier();     // the last hop's reads of `state` are done before the next sequence overwrites it
  }  // sequences
}

#include "gcrnn_fused_seq.h"

typedef void (*fused_kern_t)(const uint16_t*, const uint16_t*, uint16_t*, const uint4*, const float*, const float*,
                             const float*, const int32_t*, const int32_t*, const int32_t*, const float*, const float4*,
                             const uint2*, const float*, float*, const uint16_t*, const uint16_t*, int, int, int, int, int, const int32_t*, float, const uint16_t*, uint16_t*, int);

struct FusedGraphArgs {
  const int32_t* tile_nodes; const int32_t* tile_off; const int32_t* ell_col; const float* ell_val;
  const void* ell_val4; const void* ell_col4; int64_t entries;
  float uniform_w = 0.f;      // != 0: every non-zero carries this weight and the padding entries point at zero rows (gcrnn_ell_fill_z)
  int img16 = 0;              // != 0: tile_nodes / ell_col4 address a bf16 hop image (32-byte rows, graph.fused_plan_img16): matrix-core sums
};

template <int K, int HS, int XS>
int fused_launch_t(int mode /*0 plain, 1 gated, 2 gate pre-pass, 3 BPTT data gradient, 4 gate-gradient pass, 5 filter-output pass, 6 node-gated steps, 7 node-gated BPTT data chain, 8 one BPTT step*/, const void* xs,
                          const void* h0, void* hs, const void* wpack, const float* bias, const float* gi, const float* gf,
                          const float* gate_w, float* gate_out, const FusedGraphArgs& ga, int64_t B, int64_t T, int64_t N,
                          hipStream_t st, const void* bw_dHs = nullptr, const void* bw_hs = nullptr, const void* bw_h0 = nullptr,
                          void* bw_dh0 = nullptr, void* huser = nullptr, void* const* step_events = nullptr, int huser_last_only = 0,
                          const int32_t* hzero_flag = nullptr) {
  constexpr int F = 32 * HS, G = 32 * XS, KS = HS + XS;
  const size_t base = (size_t)NP * FC * 4 + (size_t)K * KS * 1024;
  // uniform-weight plan: column words only in LDS (2 instead of 6 bytes per slot and entry)
  const bool uni = ga.uniform_w != 0.f && ga.ell_col4 && GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8 &&
                   base + (size_t)ga.entries * 16 * 2 <= 160 * 1024;
  const size_t resident_bytes = base + (size_t)ga.entries * 16 * (uni ? 2 : 6);
  const bool resident = resident_bytes <= 160 * 1024 && ga.ell_val4 && ga.ell_col4;
  // inline pack of the next step's input (mode 0 with the user-layout X in bw_dHs): needs the uniform image and room for a
  // [G][NP / NCH] bf16 tile behind it
  // (mode 3: the BPTT chain lays out dH_{t-2} the same way, the user-layout dH arrives in xs and the tile has F rows)
  const bool inline_bw = (mode == 3 && xs != nullptr) || (mode == 7 && bw_h0 != nullptr) || (mode == 8 && xs != nullptr);      // (mode 7: the user-layout dH arrives in bw_h0)
  const size_t xtile_bytes = (size_t)(inline_bw ? F : G) * (NP / (F / FC)) * 2;
  const bool prepass_pack = (mode == 2 && bw_dHs != nullptr);       // gate pre-pass that also lays out X (sequence-resident kernel only)
  const bool inline_pack = (mode == 0 && bw_dHs != nullptr) || inline_bw || prepass_pack;
  if (inline_pack && !prepass_pack && !((XS > 0 || inline_bw) && uni && resident && N % 8 == 0 && resident_bytes + xtile_bytes <= 160 * 1024 &&
                                        T * (inline_bw ? F : G) * N <= 2147483647LL))
    return GCRNN_ERR_UNSUPPORTED;
  if (prepass_pack && !(XS > 0 && uni && N % 8 == 0 && T * G * N <= 2147483647LL)) return GCRNN_ERR_UNSUPPORTED;
  const size_t lds = (resident ? resident_bytes : base) + ((inline_pack && !prepass_pack) ? xtile_bytes : 0);
  if (ga.img16 && !(uni && resident)) return GCRNN_ERR_UNSUPPORTED;      // the bf16-image plan needs the uniform-weight asm stream on the LDS-resident graph (every mode has its UNI == 2 instantiation)
  fused_kern_t kern;
  const bool head = (mode == 0 || mode == 1) && gate_w != nullptr;      // fused output head: EPI 6 instantiations
  if (head) {
    if constexpr (XS > 0) {
      if (huser) return GCRNN_ERR_BAD_SHAPE;                             // (the head replaces the user-layout copy of H)
      if (mode == 1) kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, true, true, 6> : (fused_kern_t)fused_step_kernel<K, HS, XS, true, false, 6>;
      else kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 6> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, false, 6>;
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
      if (uni && resident) kern = mode == 1 ? (fused_kern_t)fused_step_kernel<K, HS, XS, true, true, 6, 1> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 6, 1>;
      if (uni && resident && ga.img16) kern = mode == 1 ? (fused_kern_t)fused_step_kernel<K, HS, XS, true, true, 6, 2> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 6, 2>;
#endif
    } else {
      return GCRNN_ERR_UNSUPPORTED;
    }
  } else if (mode == 6) {
    if constexpr (XS == 0) {
      kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 5> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, false, 5>;
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
      if (uni && resident) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 5, 1>;
      if (uni && resident && ga.img16) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 5, 2>;
#endif
    }
    else return GCRNN_ERR_UNSUPPORTED;
  }
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
  else if (mode == 5 && uni && resident && ga.img16) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 4, 2>;
  else if (mode == 5 && uni && resident) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 4, 1>;
  else if ((mode == 3 || mode == 7 || mode == 8) && uni && resident && ga.img16) {
    if constexpr (XS == 0) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 2, 2>;
    else return GCRNN_ERR_UNSUPPORTED;
  }
  else if ((mode == 3 || mode == 7 || mode == 8) && uni && resident) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 2, 1>;
  else if (mode == 2 && uni && resident && ga.img16) {
    if constexpr (XS > 0) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 1, 2>;
    else return GCRNN_ERR_UNSUPPORTED;
  }
  else if (mode == 2 && uni && resident) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 1, 1>;
  else if (mode == 4 && uni && resident && ga.img16) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 3, 2>;
  else if (mode == 4 && uni && resident) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 3, 1>;
#endif
  else if (mode == 5) kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 4> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, false, 4>;
  else if (mode == 4) kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 3> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, false, 3>;
  else if (mode == 3 || mode == 7 || mode == 8) kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 2> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, false, 2>;
  else if (mode == 2) kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 1> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, false, 1>;
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
  else if (mode == 1 && uni && resident && ga.img16) {
    if constexpr (XS > 0) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, true, true, 0, 2>;
    else return GCRNN_ERR_UNSUPPORTED;
  }
  else if (mode == 1 && uni && resident) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, true, true, 0, 1>;
  else if (mode == 0 && uni && resident && ga.img16) {
    if constexpr (XS > 0) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 0, 2>;
    else return GCRNN_ERR_UNSUPPORTED;
  }
  else if (mode == 0 && uni && resident) kern = (fused_kern_t)fused_step_kernel<K, HS, XS, false, true, 0, 1>;
#endif
  else if (mode == 1) kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, true, true> : (fused_kern_t)fused_step_kernel<K, HS, XS, true, false>;
  else                kern = resident ? (fused_kern_t)fused_step_kernel<K, HS, XS, false, true> : (fused_kern_t)fused_step_kernel<K, HS, XS, false, false>;
  const int NCH = F / FC;
  const uint16_t* x = (const uint16_t*)xs;
  uint16_t* h = (uint16_t*)hs;
  const int64_t xstep = B * NP * G, hstep = B * NP * F;
#if GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8
  // Sequence-resident kernel (gcrnn_fused_seq.h): one workgroup per sequence keeps the operand in registers for all chunks -- the
  // un-gated forward steps and the plain BPTT data chain on uniform-weight bf16-image plans, when the batch fills the chip.
  if (ga.uniform_w != 0.f && ga.img16 && ga.ell_col4 && !head && !step_events && (mode == 0 || mode == 1 || mode == 2 || mode == 3 || mode == 5 || mode == 6) &&
      fused_seq_wanted((mode == 2 || mode == 5) ? B * T : B, NCH)) {
    const size_t tap_extra = (mode == 2 && bw_hs) ? (size_t)huser_last_only * NP * 4 + (size_t)NCH * 3 * 512 : 0;      // per-item tap accumulators + staged fragments
    const size_t slds = fused_seq_lds<K, HS, XS>(ga.entries, inline_pack, mode == 3 ? F : G, tap_extra);
    const unsigned sgrid = (unsigned)(B < GCRNN_SEQ_MAX_GRID ? B : GCRNN_SEQ_MAX_GRID);
    const bool persist = fused_seq_persistent();
    SeqArgs sa{};
    sa.wpack = (const uint4*)wpack; sa.tile_nodes = ga.tile_nodes; sa.tile_off = ga.tile_off; sa.ell_col4 = (const uint2*)ga.ell_col4;
    sa.entries = (int)ga.entries; sa.B = (int)B; sa.N = (int)N; sa.uni_w = ga.uniform_w;
    if (mode == 5 && slds) {
      // filter output A(S) operand + b of every (t, b) item, one workgroup per item (time-chunked like mode 2): the operand is one
      // state-like array (XS == 0) or [0 | x_t] with the zero half skipped
      auto sk = fused_seq_kernel<K, HS, XS, 4>;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds) != hipSuccess)
        return GCRNN_ERR_LAUNCH;
      const int64_t row_bytes = (int64_t)NP * (F > G ? F : G) * 2;
      int64_t tchunk = (2147483647LL / row_bytes) / B;
      if (tchunk < 1) return GCRNN_ERR_BAD_SHAPE;
      if (tchunk > T) tchunk = T;
      GCRNN_PRE_LAUNCH();
      for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
        const int64_t nt = (T - t0 < tchunk) ? T - t0 : tchunk, items = nt * B;
        SeqArgs s1 = sa;
        s1.bias = bias; s1.B = (int)items; s1.nsteps = 1;
        if (XS == 0) s1.hfirst = (const uint16_t*)h0 + t0 * hstep;
        else s1.x0 = x + t0 * xstep;
        s1.out0 = h + t0 * hstep;
        sk<<<(unsigned)(items < GCRNN_SEQ_MAX_GRID ? items : GCRNN_SEQ_MAX_GRID), STHREADS, slds, st>>>(s1);
      }
      GCRNN_CHECK_LAUNCH();
      return GCRNN_OK;
    }
    if constexpr (XS > 0) {
      if (mode == 2 && bw_hs != nullptr) {
        // (the pre-pass with fused F -> 1 tap dots exists on this kernel only; bw_hs = tap fragments, bw_dh0 = [items][ntaps][N] fp32,
        //  huser_last_only carries the tap count)
        if (!slds || !bw_dh0 || huser_last_only < 1 || huser_last_only > 8) return GCRNN_ERR_UNSUPPORTED;
      }
      if (mode == 2 && slds) {
        // gate pre-pass: every (t, b) item of one gate in one launch (split over whole time steps where the 32-bit buffer offsets of
        // items * NP * max(F, G) * 2 bytes would overflow), one workgroup per item
        auto sk = fused_seq_kernel<K, HS, XS, 1>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds) != hipSuccess)
          return GCRNN_ERR_LAUNCH;
        const int64_t row_bytes = (int64_t)NP * (F > G ? F : G) * 2;
        int64_t tchunk = (2147483647LL / row_bytes) / B;
        if (tchunk < 1) return GCRNN_ERR_BAD_SHAPE;
        if (tchunk > T) tchunk = T;
        GCRNN_PRE_LAUNCH();
        for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
          const int64_t nt = (T - t0 < tchunk) ? T - t0 : tchunk, items = nt * B;
          SeqArgs s1 = sa;
          s1.bias = bias; s1.B = (int)items; s1.hmod = (int)B; s1.nsteps = 1;
          s1.x0 = x + t0 * xstep; s1.hfirst = (const uint16_t*)h0;
          s1.out0 = h ? h + t0 * hstep : nullptr;
          s1.gw = gate_w; s1.go0 = gate_out ? gate_out + t0 * B * (NCH * SWAVES) : nullptr; s1.flags = hzero_flag;
          if (bw_hs) {
            if (tchunk != T) return GCRNN_ERR_UNSUPPORTED;
            s1.tapf = (const uint2*)bw_hs; s1.taps_out = (float*)bw_dh0; s1.ntaps = huser_last_only;
            s1.sacc_off = (int)(slds - tap_extra); s1.tapf_off = s1.sacc_off + huser_last_only * NP * 4;
          }
          if (prepass_pack) {
            // every item lays out the operand of the workgroup's next item from the user-layout X [B][T][G][N] (the caller laid out
            // the first min(items, 256)): the pass over X that packed the whole input goes away
            if (tchunk != T) return GCRNN_ERR_UNSUPPORTED;
            s1.pk_src0 = (const uint16_t*)bw_dHs; s1.pksrc_stride = G * N; s1.pk_stride = (int)(T * G * N);
            s1.pk_dst0 = const_cast<uint16_t*>(x);
          }
          sk<<<(unsigned)(items < GCRNN_SEQ_MAX_GRID ? items : GCRNN_SEQ_MAX_GRID), STHREADS, slds, st>>>(s1);
        }
        GCRNN_CHECK_LAUNCH();
        return GCRNN_OK;
      }
      if (mode == 1 && slds) {
        // time-gated recurrence: the gates of every (t, b) are known before the first step (they read (x_t, h0)); one persistent launch
        auto sk = fused_seq_kernel<K, HS, XS, 0, true>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds) != hipSuccess)
          return GCRNN_ERR_LAUNCH;
        GCRNN_PRE_LAUNCH();
        sa.bias = bias;
        sa.x0 = x; sa.xstride = xstep;
        sa.hfirst = (const uint16_t*)h0; sa.hrest = h; sa.hstride = hstep;
        sa.out0 = h; sa.ostride = hstep;
        sa.gi0 = gi; sa.gf0 = gf; sa.gfstride = B;
        sa.a1 = (const uint16_t*)huser; sa.a1stride = F * N; sa.a1_last_only = huser_last_only ? 1 : 0;
        sa.ubstride = (int)((huser_last_only ? 1 : T) * F * N);
        if (persist) {
          sa.nsteps = (int)T;
          sk<<<sgrid, STHREADS, slds, st>>>(sa);
        } else {
          for (int64_t t = 0; t < T; ++t) {
            SeqArgs s1 = sa;
            s1.nsteps = 1;
            s1.x0 = x + t * xstep;
            s1.hfirst = (t == 0) ? (const uint16_t*)h0 : h + (t - 1) * hstep;
            s1.out0 = h + t * hstep;
            s1.gi0 = gi + t * B; s1.gf0 = gf + t * B;
            s1.a1 = !huser ? nullptr : (!huser_last_only ? (const uint16_t*)huser + t * F * N : (t == T - 1 ? (const uint16_t*)huser : nullptr));
            s1.a1_last_only = 0;
            sk<<<sgrid, STHREADS, slds, st>>>(s1);
          }
        }
        GCRNN_CHECK_LAUNCH();
        return GCRNN_OK;
      }
      if (mode == 0 && slds) {
        auto sk = fused_seq_kernel<K, HS, XS, 0>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds) != hipSuccess)
          return GCRNN_ERR_LAUNCH;
        GCRNN_PRE_LAUNCH();
        sa.bias = bias;
        sa.x0 = x; sa.xstride = xstep;
        sa.hfirst = (const uint16_t*)h0; sa.hrest = h; sa.hstride = hstep;
        sa.out0 = h; sa.ostride = hstep;
        sa.a1 = (const uint16_t*)huser; sa.a1stride = F * N; sa.a1_last_only = huser_last_only ? 1 : 0;
        sa.ubstride = (int)((huser_last_only ? 1 : T) * F * N);
        // A/B (GCRNN_SEQ_PK_AHEAD=2): the inline pack two steps ahead + the x half of the next operand requested during the last hop.
        // Measured on one box (profiles/r03_x_prefetch_ab.txt): 68.3-69.3 vs 66.2-67.6 us per step -- SLOWER (the 16 extra requests per lane
        // join the LDS-DMA pieces and the stores of the busiest moment of the step); default 1.
        const char* pka_env = getenv("GCRNN_SEQ_PK_AHEAD");
        const int pka = (pka_env && pka_env[0] == '2') ? 2 : 1;
        if (inline_pack && T > pka) {
          // step t lays out x_{t+2} (the caller packed x_0 and x_1): x_{t+1} is complete while step t runs, the kernel requests it early
          sa.pk_src0 = (const uint16_t*)bw_dHs + pka * G * N; sa.pksrc_stride = G * N;
          sa.pk_dst0 = const_cast<uint16_t*>(x) + pka * xstep; sa.pkdst_stride = xstep;
          sa.pk_stride = (int)(T * G * N);
          sa.pk_ahead = pka;
        }
        sa.xprefetch = (pka == 2) ? 1 : 0;
        if (persist) {
          sa.nsteps = (int)T;
          sk<<<sgrid, STHREADS, slds, st>>>(sa);
        } else {
          for (int64_t t = 0; t < T; ++t) {         // the same kernel, one step per launch
            SeqArgs s1 = sa;
            s1.nsteps = 1; s1.pk_all = 1;
            s1.x0 = x + t * xstep;
            s1.hfirst = (t == 0) ? (const uint16_t*)h0 : h + (t - 1) * hstep;
            s1.out0 = h + t * hstep;
            s1.a1 = !huser ? nullptr : (!huser_last_only ? (const uint16_t*)huser + t * F * N : (t == T - 1 ? (const uint16_t*)huser : nullptr));
            s1.a1_last_only = 0;
            const bool pkt = sa.pk_src0 && t + pka < T;
            s1.pk_src0 = pkt ? sa.pk_src0 + t * sa.pksrc_stride : nullptr;
            s1.pk_dst0 = pkt ? sa.pk_dst0 + t * sa.pkdst_stride : nullptr;
            sk<<<sgrid, STHREADS, slds, st>>>(s1);
          }
        }
        GCRNN_CHECK_LAUNCH();
        return GCRNN_OK;
      }
    } else {
      if (mode == 6 && slds) {
        // node-gated recurrence (graphML.py:2379-2407, 2420-2423): every gate is known before step 0, so ONE persistent launch walks
        // the T steps; Yx_t = A(S)x_t + b from the all-items pass (bw_dHs), node gates gate_w [T][2][B][N], optional Yh output (bw_dh0)
        auto sk = fused_seq_kernel<K, HS, 0, 5>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds) != hipSuccess)
          return GCRNN_ERR_LAUNCH;
        GCRNN_PRE_LAUNCH();
        sa.bias = bias;
        sa.hfirst = (const uint16_t*)h0; sa.hrest = h; sa.hstride = hstep;
        sa.out0 = h; sa.ostride = hstep;
        sa.a0 = (const uint16_t*)bw_dHs; sa.a0stride = hstep;
        sa.ng0 = gate_w; sa.ngstride = 2 * B * N;
        sa.gi0 = gi; sa.gf0 = gi ? gf : nullptr; sa.gfstride = B;
        sa.yh0 = (uint16_t*)bw_dh0; sa.yhstride = hstep;
        sa.a1 = (const uint16_t*)huser; sa.a1stride = F * N; sa.a1_last_only = huser_last_only ? 1 : 0;
        sa.ubstride = (int)((huser_last_only ? 1 : T) * F * N);
        if (persist) {
          sa.nsteps = (int)T;
          sk<<<sgrid, STHREADS, slds, st>>>(sa);
        } else {
          for (int64_t t = 0; t < T; ++t) {
            SeqArgs s1 = sa;
            s1.nsteps = 1;
            s1.hfirst = (t == 0) ? (const uint16_t*)h0 : h + (t - 1) * hstep;
            s1.out0 = h + t * hstep;
            s1.a0 = sa.a0 + t * hstep;
            s1.ng0 = gate_w + t * 2 * B * N;
            s1.gi0 = gi ? gi + t * B : nullptr; s1.gf0 = gi ? gf + t * B : nullptr;
            s1.yh0 = sa.yh0 ? sa.yh0 + t * hstep : nullptr;
            s1.a1 = !huser ? nullptr : (!huser_last_only ? (const uint16_t*)huser + t * F * N : (t == T - 1 ? (const uint16_t*)huser : nullptr));
            s1.a1_last_only = 0;
            sk<<<sgrid, STHREADS, slds, st>>>(s1);
          }
        }
        GCRNN_CHECK_LAUNCH();
        return GCRNN_OK;
      }
      if (mode == 3 && slds) {
        auto sk = fused_seq_kernel<K, HS, 0, 2>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds) != hipSuccess)
          return GCRNN_ERR_LAUNCH;
        GCRNN_PRE_LAUNCH();
        const uint16_t* dH = (const uint16_t*)bw_dHs;
        const uint16_t* hst = (const uint16_t*)bw_hs;
        const int64_t gstep = B * (NCH * SWAVES);
        if (T >= 2) {
          // steps i = 0 .. T-2 walk t = T-1 .. 1: operand dpre_t, output dpre_{t-1}, epilogue operands dH_{t-1} / h_{t-1}; step i lays out dH_{t-2}
          sa.hfirst = h + (T - 1) * hstep; sa.hrest = h + (T - 2) * hstep; sa.hstride = -hstep;
          sa.out0 = h + (T - 2) * hstep; sa.ostride = -hstep;
          sa.gf0 = gf ? gf + (T - 1) * B : nullptr; sa.gfstride = -B;
          sa.go0 = gate_out ? gate_out + (T - 1) * gstep : nullptr; sa.gostride = -gstep;
          sa.a0 = dH + (T - 2) * hstep; sa.a0stride = -hstep;
          sa.a1 = hst + (T - 2) * hstep; sa.a1stride = -hstep;
          if (inline_bw && T >= 3) {
            sa.pk_src0 = (const uint16_t*)xs + (T - 3) * F * N; sa.pksrc_stride = -(F * N);
            sa.pk_dst0 = const_cast<uint16_t*>(dH) + (T - 3) * hstep; sa.pkdst_stride = -hstep;
            sa.pk_stride = (int)(T * F * N);
          }
          if (persist) {
            sa.nsteps = (int)(T - 1);
            sk<<<sgrid, STHREADS, slds, st>>>(sa);
          } else {
            for (int64_t i = 0; i + 1 < T; ++i) {
              SeqArgs s1 = sa;
              s1.nsteps = 1; s1.pk_all = 1;
              s1.hfirst = sa.hfirst + i * sa.hstride;
              s1.out0 = sa.out0 + i * sa.ostride;
              s1.gf0 = sa.gf0 ? sa.gf0 + i * sa.gfstride : nullptr;
              s1.go0 = sa.go0 ? sa.go0 + i * sa.gostride : nullptr;
              s1.a0 = sa.a0 + i * sa.a0stride; s1.a1 = sa.a1 + i * sa.a1stride;
              const bool pkt = sa.pk_src0 && i + 2 < T;
              s1.pk_src0 = pkt ? sa.pk_src0 + i * sa.pksrc_stride : nullptr;
              s1.pk_dst0 = pkt ? sa.pk_dst0 + i * sa.pkdst_stride : nullptr;
              sk<<<sgrid, STHREADS, slds, st>>>(s1);
            }
          }
        }
        if (bw_dh0 || gate_out) {
          // d h0 (and the forget gate's partials of step 0): the raw state gradient of dpre_0, no tanh' and no upstream term
          SeqArgs s0{};
          s0.wpack = sa.wpack; s0.tile_nodes = sa.tile_nodes; s0.tile_off = sa.tile_off; s0.ell_col4 = sa.ell_col4;
          s0.entries = sa.entries; s0.B = sa.B; s0.N = sa.N; s0.uni_w = sa.uni_w;
          s0.hfirst = h; s0.out0 = (uint16_t*)bw_dh0; s0.gf0 = gf; s0.go0 = gate_out;
          s0.a1 = gate_out ? (const uint16_t*)bw_h0 : nullptr;
          s0.nsteps = 1;
          sk<<<sgrid, STHREADS, slds, st>>>(s0);
        }
        GCRNN_CHECK_LAUNCH();
        return GCRNN_OK;
      }
    }
  }
#endif
  if (prepass_pack || (mode == 2 && bw_hs)) return GCRNN_ERR_UNSUPPORTED;      // (gcrnn_fused_gate_prepass_lays_out / _taps_supported tell the caller beforehand)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  // one workgroup per CU: 256 / NCH sequence slots (rounded to the 8 XCDs), fewer when the batch is small
  auto grid_for = [&](int64_t items) {
    int64_t slots = cdiv(items, 8) * 8;
    const int64_t max_slots = (256 / NCH) / 8 * 8 > 0 ? (256 / NCH) / 8 * 8 : 8;
    if (slots > max_slots) slots = max_slots;
    return (unsigned)(slots * NCH);
  };
  GCRNN_PRE_LAUNCH();
  if (mode == 7) {
    // BPTT data chain of the node-gated cell: hs = dpre [T][B][NP][F] (slot T-1 seeded), xs = dyh [T][B][NP][F] = (gf nf) . dpre (slot
    // T-1 seeded; every launch reads its operand from it and writes the next one), gate_w = gf nf [T][B][N] fp32
    if constexpr (XS == 0) {
      const unsigned grid = grid_for(B);
      const uint16_t* dH = (const uint16_t*)bw_dHs;
      const uint16_t* hst = (const uint16_t*)bw_hs;
      uint16_t* dyh = (uint16_t*)const_cast<void*>(xs);
      for (int64_t t = T - 1; t >= 1; --t) {
        const uint16_t* dun = (inline_bw && t >= 2) ? (const uint16_t*)bw_h0 + (t - 2) * F * N : nullptr;      // inline pack of dH_{t-2}, as in mode 3
        kern<<<grid, STHREADS, lds, st>>>(dyh + (t - 1) * hstep, dyh + t * hstep, h + (t - 1) * hstep, (const uint4*)wpack, nullptr, nullptr,
                                     nullptr, ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val, (const float4*)ga.ell_val4,
                                     (const uint2*)ga.ell_col4, gate_w + (t - 1) * B * N, nullptr, dH + (t - 1) * hstep, hst + (t - 1) * hstep, 0,
                                     (int)ga.entries, (int)B, (int)B, (int)N, nullptr, uni ? ga.uniform_w : 0.f,
                                     dun, dun ? const_cast<uint16_t*>(dH) + (t - 2) * hstep : nullptr, (int)(T * F * N));
      }
    } else {
      return GCRNN_ERR_UNSUPPORTED;
    }
  } else if (mode == 6) {
    // node-gated recurrence: h = hs, Yx_t = bw_dHs [T][B][NP][F], node gates gate_w [T][2][B][N], optional Yh output bw_dh0 [T][B][NP][F]
    const unsigned grid = grid_for(B);
    const uint16_t* yx = (const uint16_t*)bw_dHs;
    uint16_t* yh = (uint16_t*)bw_dh0;
    for (int64_t t = 0; t < T; ++t) {
      const uint16_t* hp = (t == 0) ? (const uint16_t*)h0 : h + (t - 1) * hstep;
      kern<<<grid, STHREADS, lds, st>>>(yh ? yh + t * hstep : nullptr, hp, h + t * hstep, (const uint4*)wpack, bias, gi ? gi + t * B : nullptr,
                                   gi ? gf + t * B : nullptr, ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val,
                                   (const float4*)ga.ell_val4, (const uint2*)ga.ell_col4, gate_w + t * 2 * B * N, nullptr, yx + t * hstep,
                                   !huser ? nullptr : (!huser_last_only ? (const uint16_t*)huser + t * F * N : (t == T - 1 ? (const uint16_t*)huser : nullptr)),
                                   (int)((huser_last_only ? 1 : T) * F * N), (int)ga.entries, (int)B, (int)B, (int)N, nullptr, uni ? ga.uniform_w : 0.f, nullptr, nullptr, 0);
    }
  } else if (mode == 5) {
    // filter output of every (t, b) item in one launch (split over whole time steps like mode 2 / 4): hs receives A(S)x_t + b
    const int64_t row_bytes = (int64_t)NP * (F > G ? F : G) * 2;
    int64_t tchunk = (2147483647LL / row_bytes) / B;
    if (tchunk < 1) return GCRNN_ERR_BAD_SHAPE;
    if (tchunk > T) tchunk = T;
    for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
      const int64_t nt = (T - t0 < tchunk) ? T - t0 : tchunk, items = nt * B;
      if (XS == 0)        // operand = one [T*B][NP][F] array (an input with G == F, packed like a state)
        kern<<<grid_for(items), STHREADS, lds, st>>>(nullptr, (const uint16_t*)h0 + t0 * hstep, h + t0 * hstep, (const uint4*)wpack, bias, nullptr, nullptr,
                                     ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val, (const float4*)ga.ell_val4, (const uint2*)ga.ell_col4,
                                     nullptr, nullptr, nullptr, nullptr, 0, (int)ga.entries, (int)items, (int)items, (int)N, nullptr, uni ? ga.uniform_w : 0.f, nullptr, nullptr, 0);
      else                // operand [0 | x_t]: ONE all-zero state block shared by every item (hmod = 1)
        kern<<<grid_for(items), STHREADS, lds, st>>>(x + t0 * xstep, (const uint16_t*)h0, h + t0 * hstep, (const uint4*)wpack, bias, nullptr, nullptr,
                                     ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val, (const float4*)ga.ell_val4, (const uint2*)ga.ell_col4,
                                     nullptr, nullptr, nullptr, nullptr, 0, (int)ga.entries, (int)items, 1, (int)N, nullptr, uni ? ga.uniform_w : 0.f, nullptr, nullptr, 0);
    }
  } else if (mode == 2 || mode == 4) {
    // no recurrence: all (t, b) items in one launch -- split over whole time steps where the 32-bit buffer offsets of
    // one launch (items * NP * max(F, G) * 2 bytes) would overflow
    const int64_t row_bytes = (int64_t)NP * (F > G ? F : G) * 2;
    int64_t tchunk = (2147483647LL / row_bytes) / B;
    if (tchunk < 1) return GCRNN_ERR_BAD_SHAPE;
    if (tchunk > T) tchunk = T;
    for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
      const int64_t nt = (T - t0 < tchunk) ? T - t0 : tchunk, items = nt * B;
      float* go = gate_out + t0 * B * (NCH * SWAVES);
      if (mode == 2)      // operands [h0 | x_t]; optional store of c_t = tanh(pre) into hs
        kern<<<grid_for(items), STHREADS, lds, st>>>(x + t0 * xstep, (const uint16_t*)h0, h ? h + t0 * hstep : nullptr, (const uint4*)wpack, bias,
                                     nullptr, nullptr, ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val, (const float4*)ga.ell_val4,
                                     (const uint2*)ga.ell_col4, gate_w, go, nullptr, nullptr, 0, (int)ga.entries, (int)items, (int)B, (int)N, hzero_flag, uni ? ga.uniform_w : 0.f, nullptr, nullptr, 0);
      else if (XS == 0)   // operand = one [T*B][NP][F] array, per-item dpre in bw_dHs
        kern<<<grid_for(items), STHREADS, lds, st>>>(nullptr, (const uint16_t*)h0 + t0 * hstep, nullptr, (const uint4*)wpack, bias, nullptr, nullptr,
                                     ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val, (const float4*)ga.ell_val4,
                                     (const uint2*)ga.ell_col4, nullptr, go, (const uint16_t*)bw_dHs + t0 * hstep, nullptr, 0,
                                     (int)ga.entries, (int)items, (int)items, (int)N, nullptr, uni ? ga.uniform_w : 0.f, nullptr, nullptr, 0);
      else                // input filter with G != F: operand [0 | x_t] -- ONE all-zero state block shared by every item (hmod = 1)
        kern<<<grid_for(items), STHREADS, lds, st>>>(x + t0 * xstep, (const uint16_t*)h0, nullptr, (const uint4*)wpack, bias, nullptr, nullptr,
                                     ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val, (const float4*)ga.ell_val4,
                                     (const uint2*)ga.ell_col4, nullptr, go, (const uint16_t*)bw_dHs + t0 * hstep, nullptr, 0,
                                     (int)ga.entries, (int)items, 1, (int)N, nullptr, uni ? ga.uniform_w : 0.f, nullptr, nullptr, 0);
    }
  } else if (mode == 8) {
    // ONE BPTT step with explicit arrays (edge-gated cell: the operand of step t is the attention backward's output):
    // h0 = operand [B][NP][F], hs = dpre_{t-1} (out), bw_dHs = dH_{t-1}, bw_hs = h_{t-1}
    // inline pack: xs = the user-layout block dH[0][t-2] (or null), bw_dh0 = dHs[t-2] to lay it out into, T = the sequence length
    kern<<<grid_for(B), STHREADS, lds, st>>>(nullptr, (const uint16_t*)h0, h, (const uint4*)wpack, nullptr, nullptr, nullptr, ga.tile_nodes,
                                 ga.tile_off, ga.ell_col, ga.ell_val, (const float4*)ga.ell_val4, (const uint2*)ga.ell_col4, nullptr, nullptr,
                                 (const uint16_t*)bw_dHs, (const uint16_t*)bw_hs, 0, (int)ga.entries, (int)B, (int)B, (int)N, nullptr,
                                 uni ? ga.uniform_w : 0.f, (const uint16_t*)xs, (uint16_t*)bw_dh0, (int)(T * F * N));
  } else if (mode == 3) {
    // BPTT: hs (= dpre, [T][B][NP][F]) already holds dpre_{T-1}; walk t = T-1 .. 1, then optionally d h0.
    // gf (time-gated cell, [T][B]): step t's recurrent gradient is scaled by its forget gate gf_t.
    const unsigned grid = grid_for(B);
    const uint16_t* dH = (const uint16_t*)bw_dHs;
    const uint16_t* hst = (const uint16_t*)bw_hs;
    const int64_t gstep = B * (NCH * SWAVES);             // gate_out (or null): [T][B][NCH*SWAVES] partials of <h_{t-1}, adjoint chain of dpre_t>
    for (int64_t t = T - 1; t >= 1; --t) {
      // inline pack: this launch consumes dHs[t-1] and lays out dHs[t-2] from the user-layout dH (the caller packed the last two steps)
      const uint16_t* dun = (inline_bw && t >= 2) ? (const uint16_t*)xs + (t - 2) * F * N : nullptr;
      kern<<<grid, STHREADS, lds, st>>>(nullptr, h + t * hstep, h + (t - 1) * hstep, (const uint4*)wpack, nullptr, nullptr,
                                   gf ? gf + t * B : nullptr, ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val,
                                   (const float4*)ga.ell_val4, (const uint2*)ga.ell_col4, nullptr, gate_out ? gate_out + t * gstep : nullptr,
                                   dH + (t - 1) * hstep, hst + (t - 1) * hstep, 0, (int)ga.entries, (int)B, (int)B, (int)N, nullptr, uni ? ga.uniform_w : 0.f,
                                   dun, dun ? const_cast<uint16_t*>(dH) + (t - 2) * hstep : nullptr, (int)(T * F * N));
    }
    if (bw_dh0 || gate_out)
      kern<<<grid, STHREADS, lds, st>>>(nullptr, h, (uint16_t*)bw_dh0, (const uint4*)wpack, nullptr, nullptr, gf, ga.tile_nodes,
                                   ga.tile_off, ga.ell_col, ga.ell_val, (const float4*)ga.ell_val4, (const uint2*)ga.ell_col4,
                                   nullptr, gate_out, nullptr, gate_out ? (const uint16_t*)bw_h0 : nullptr, 0, (int)ga.entries,
                                   (int)B, (int)B, (int)N, nullptr, uni ? ga.uniform_w : 0.f, nullptr, nullptr, 0);
  } else {
    const unsigned grid = grid_for(B);
    for (int64_t t = 0; t < T; ++t) {
      const uint16_t* hp = (t == 0) ? (const uint16_t*)h0 : h + (t - 1) * hstep;
      // step_events[t] (or null): the launch of step t first waits for that event -- x_t is being packed on another stream
      if (step_events && step_events[t] && hipStreamWaitEvent(st, (hipEvent_t)step_events[t], 0) != hipSuccess) return GCRNN_ERR_LAUNCH;
      // inline pack (mode 0, uniform graph): launch t also lays out x_{t+1} from the user-layout X (bw_dHs) into xs[t+1]
      const uint16_t* xun = (inline_pack && t + 1 < T) ? (const uint16_t*)bw_dHs + (t + 1) * G * N : nullptr;
      // modes 0 / 1 with gate_w: the fused output head (weights [F]); gate_out = its partials [T][B][F/16][N]
      kern<<<grid, STHREADS, lds, st>>>(x + t * xstep, hp, h + t * hstep, (const uint4*)wpack, bias, mode == 1 ? gi + t * B : nullptr,
                                   mode == 1 ? gf + t * B : nullptr, ga.tile_nodes, ga.tile_off, ga.ell_col, ga.ell_val,
                                   (const float4*)ga.ell_val4, (const uint2*)ga.ell_col4, gate_w, gate_w ? gate_out + t * B * NCH * N : nullptr, nullptr,
                                   !huser ? nullptr : (!huser_last_only ? (const uint16_t*)huser + t * F * N : (t == T - 1 ? (const uint16_t*)huser : nullptr)),
                                   (int)((huser_last_only ? 1 : T) * F * N),
                                   (int)ga.entries, (int)B, (int)B, (int)N, nullptr, uni ? ga.uniform_w : 0.f,
                                   xun, xun ? const_cast<uint16_t*>(x) + (t + 1) * xstep : nullptr, (int)(T * G * N));
    }
  }
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// The step kernel's instantiations are spread over several translation units (gcrnn_fused_step_k*.hip) so that the library
// builds in parallel; gcrnn_fused.hip only declares them.
#define GCRNN_STEP_SIG (int, const void*, const void*, void*, const void*, const float*, const float*, const float*, const float*, float*, const FusedGraphArgs&, int64_t, int64_t, int64_t, hipStream_t, const void*, const void*, const void*, void*, void*, void* const*, int, const int32_t*)
#define GCRNN_STEP_DECLARE(K, HS, XS) extern template int fused_launch_t<K, HS, XS> GCRNN_STEP_SIG;
#define GCRNN_STEP_DEFINE(K, HS, XS) template int fused_launch_t<K, HS, XS> GCRNN_STEP_SIG;
// the (HS, XS) operand shapes built for a tap count K: [h | x] with F = G = 64 / 32, F = 64 with G <= 32, and the state-only
// operands of the BPTT data-gradient steps and gate-gradient passes
#define GCRNN_STEP_FOR_K5(M) M(5, 2, 2) M(5, 1, 1) M(5, 2, 1) M(5, 2, 0) M(5, 1, 0)
#define GCRNN_STEP_FOR_K4(M) M(4, 2, 2) M(4, 1, 1) M(4, 2, 1) M(4, 2, 0) M(4, 1, 0)
#define GCRNN_STEP_FOR_K3(M) M(3, 2, 2) M(3, 1, 1) M(3, 2, 1) M(3, 2, 0) M(3, 1, 0)
#define GCRNN_STEP_FOR_K2(M) M(2, 2, 2) M(2, 1, 1) M(2, 2, 1) M(2, 2, 0) M(2, 1, 0)
