// Edge gates of the fused path (reference Utils/graphML.py:2409-2416): the edge-gated cell is
//     h_t = tanh( gi att_in(A(S)x_t + b) + gf att_f(B(S)h_{t-1} + b) ),    att = GraphAttentional(F, F, 1 head) (graphML.py:1999-2128)
//     att(y)[n] = relu( sum_m z_m (S+I)[m][n] alpha[m][n] ),   z = W y,   alpha[m][.] = softmax over the support row m of
//                 LeakyReLU(a1.z_n + a2.z_m)                               (graphAttention, graphML.py:521-627)
// The mixing matrix W acts on features and the shift on nodes, so z = W(sum_k S^k u C_k + b) = sum_k S^k u (C_k W^T) + W b is a
// filter output with composite taps: the step kernel's filter-output pass (gcrnn_fused_filter_output_bf16) produces z directly and
// this file only holds what is left -- the attention itself, one workgroup per item (t, b):
//   load    z [N][F] bf16 -> LDS, scores s1[n] = a1.z_n, s2[n] = a2.z_n on the way (8 lanes per row, xor-shuffle reduction)
//   phase A per support row m: running max and sum of exp of e[m][n] (online softmax, one thread per row)
//   phase B per node n: o[n] = sum over the in-edges (m -> n) of v alpha z_m -- F/8 lanes per node, each owning 8 features; the
//           edge records {m, v} of a node are fetched by its lanes together (one coalesced load) and broadcast with shuffles;
//           alpha is recomputed from s1, s2 and the row statistics (nothing of size nnz is stored in inference)
//   epilogue  pre-pass: relu(o) (bf16, sequence-major);  step: h = tanh(gi gx + gf relu(o)) stored sequence-major for the next
//           step's filter pass and, through a transposed LDS tile, in the user layout H[b][t][f][:].
// Every sum is a gather in a fixed order: results are deterministic.
#include "gcrnn_common.h"

// In-kernel phase stamps (diagnostic builds only, -DGCRNN_EDGE_STAMPS; tools/edge_att_stamps.py): thread 0 of every workgroup records s_memtime at
// the attention kernel's phase boundaries (the last launch's values stay).
#if defined(GCRNN_EDGE_STAMPS)
static __device__ unsigned long long gcrnn_edge_stamps[1024 * 16];
#define EA_STAMP(slot)                                                                                         \
  do {                                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                               \
      unsigned long long tv_;                                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tv_)::"memory");                              \
      gcrnn_edge_stamps[blockIdx.x * 16 + (slot)] = tv_;                                                       \
    }                                                                                                          \
  } while (0)
extern "C" int gcrnn_debug_read_edge_stamps(void* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(gcrnn_edge_stamps), sizeof(unsigned long long) * 1024 * 16) == hipSuccess ? 0 : 1;
}
#else
#define EA_STAMP(slot) do {} while (0)
#endif

namespace {

#ifndef GCRNN_EDGE_THREADS
#define GCRNN_EDGE_THREADS 1024
#endif
constexpr int ETHREADS = GCRNN_EDGE_THREADS;      // 16 waves: the aggregation is a chain of dependent LDS reads, occupancy hides it

__device__ __forceinline__ float ebf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t ef2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t epack2bf(float a, float b) {      // one v_cvt_pk_bf16_f32 for the pair
  typedef __attribute__((ext_vector_type(2))) float f2v_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2v_t;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v_t{a, b}, b2v_t));
}
__device__ __forceinline__ float etanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float eexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ void unpack8(const uint4 v, float* o) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* o) {
  uint4 v;
  v.x = epack2bf(o[0], o[1]);
  v.y = epack2bf(o[2], o[3]);
  v.z = epack2bf(o[4], o[5]);
  v.w = epack2bf(o[6], o[7]);
  return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Softmax statistics (max, 1 / sum of exp) of every support row, one thread per row. The row's records and the scores they point at
// are fetched eight at a time (eight independent global loads, then eight LDS reads) -- a serial walk costs one global and one LDS
// latency PER EDGE, and at one workgroup per CU nothing else hides it.
// PRE (round 5): the bounds of row `tid` and the column indices of its first eight records were requested by the caller before it loaded z
// (they depend on the graph only): the statistics of the first pass start without a global round trip.
struct RowPre { int j0, j1; int nn[8]; };
template <int THREADS>
__device__ __forceinline__ RowPre row_softmax_prefetch(const int32_t* __restrict__ rowptr, const int2* __restrict__ r_edge, int N, int tid) {
  RowPre o;
  o.j0 = o.j1 = 0;
  if (tid < N) { o.j0 = rowptr[tid]; o.j1 = rowptr[tid + 1]; }
#pragma unroll
  for (int u = 0; u < 8; ++u) o.nn[u] = (o.j0 + u < o.j1) ? r_edge[o.j0 + u].x : -1;
  return o;
}
template <int THREADS, bool PRE = false>
__device__ __forceinline__ void row_softmax_stats(float4* __restrict__ sc, const int32_t* __restrict__ rowptr, const int2* __restrict__ r_edge,
                                                  int N, float slope, int tid, const RowPre* pre = nullptr) {
  for (int m = tid; m < N; m += THREADS) {
    const bool first = PRE && m == tid;
    const int j0 = first ? pre->j0 : rowptr[m], j1 = first ? pre->j1 : rowptr[m + 1];
    const float s2m = sc[m].y;
    float mx = -1e30f, sum = 0.f;
    for (int j = j0; j < j1; j += 8) {
      int nn[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) nn[u] = (first && j == j0) ? pre->nn[u] : ((j + u < j1) ? r_edge[j + u].x : -1);
      float ev[8];
      float cm = -1e30f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float e = (nn[u] >= 0 ? sc[nn[u]].x : 0.f) + s2m;
        e = e > 0.f ? e : slope * e;
        ev[u] = nn[u] >= 0 ? e : -1e30f;
        cm = fmaxf(cm, ev[u]);
      }
      const float nm = fmaxf(mx, cm);
      float add = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) add += nn[u] >= 0 ? eexp(ev[u] - nm) : 0.f;
      sum = sum * eexp(mx - nm) + add;
      mx = nm;
    }
    sc[m].z = mx;
    sc[m].w = sum > 0.f ? 1.f / sum : 0.f;
  }
}

// LDS carve-up: rows of 8-feature pieces, then the per-node scalars sc[n] = {s1, s2, row max, 1 / row sum}; once the aggregation
// is done the same bytes hold the transposed image of h for the user-layout store: 32-bit words [feature pair][node] = {h[n][2q], h[n][2q+1]}
// (round 5; the 8-node blocks of row q sit at block ^ (q >> 2 & 3), which spreads the pieces of one node over the banks)
template <int F> struct EdgeLds {
  static constexpr int LPN = F / 8;                  // lanes per node = 16-byte pieces of a bf16 row
  static constexpr int NPP = ETHREADS / LPN;         // nodes per pass
  static constexpr int MAXN = 1024;                  // the fused path's padded node count: a lane keeps one 16-byte piece of h per pass in registers
  static constexpr int MAXPASS = MAXN / NPP;
  static constexpr int WPITCH = MAXN;                // words per row of the transposed image [feature pair][node]
  static size_t fwd_bytes(int N) {
    const size_t a = (size_t)N * F * 2 + (size_t)N * 16, b = (size_t)(F / 2) * WPITCH * 4;
    return a > b ? a : b;
  }
};

// MODE 0: out = relu(att(z));  MODE 1: h = tanh(gi gx + gf relu(att(z)))
template <int F, int MODE>
__global__ __launch_bounds__(ETHREADS) void edge_att_fwd_kernel(
    const uint16_t* __restrict__ z,          // [items][NPad][F] bf16: the filter output with the mixing matrix folded into the taps
    const float* __restrict__ a12,           // [2][F]: a1 (scores the receiving node n), a2 (scores the row m)
    const uint16_t* __restrict__ gx,         // MODE 1: [items][NPad][F] bf16, the other branch (already relu'ed)
    const float* __restrict__ gi, const float* __restrict__ gf,      // MODE 1: per-item scalar time gates or null (= 1)
    const int32_t* __restrict__ rowptr, const int2* __restrict__ r_edge,        // support rows m: {n, bits of (S+I)[m][n]}
    const int32_t* __restrict__ t_rowptr, const int2* __restrict__ t_edge,      // support columns n: {m, bits of (S+I)[m][n]}
    const int32_t* __restrict__ t_order,     // the nodes by descending in-degree: the LPN-lane groups of a wave get equal trip counts
    uint16_t* __restrict__ out_seq,          // [items][NPad][F] bf16 (rows >= N are written as zeros); must not alias z when Huser is given
    uint16_t* __restrict__ r_out,            // MODE 1, training: relu(att(z)) [items][NPad][F] bf16, or null
    uint16_t* __restrict__ Huser,            // MODE 1: user-layout output of item 0, element (f, n) at f * N + n; or null
    int64_t hu_stride,                       // elements between the user-layout blocks of consecutive items
    int N, int NPad, float slope) {
  using L = EdgeLds<F>;
  constexpr int LPN = L::LPN, NPP = L::NPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* zl = reinterpret_cast<uint4*>(smem);
  float4* sc = reinterpret_cast<float4*>(smem + (size_t)N * F * 2);

  const int tid = threadIdx.x;
  const int p = tid % LPN, nl = tid / LPN;
  const int64_t item = blockIdx.x;
  const uint4* zsrc = reinterpret_cast<const uint4*>(z + item * NPad * F);

  EA_STAMP(0);
  // ---- (round 5) the requests, in the order their answers are needed (memory returns in order, and a dependent load waits for everything older):
  // 1. the heads of the graph-only chains (row bounds of phase A, the nodes of phase B's first two passes); 2. z, eight 16-byte loads in flight per
  // lane (N <= 1024 rows are one trip); 3. the chains' second level (phase A's first eight records, the in-edge bounds of the two passes), which
  // waits for 1. only; the records of phase B's first pass follow the scores.
  float giv = 1.f, gfv = 1.f;
  if (MODE == 1 && gi) { giv = gi[item]; gfv = gf[item]; }
  uint4* oseq = reinterpret_cast<uint4*>(out_seq + item * NPad * F);
  const uint4* gsrc = MODE == 1 ? reinterpret_cast<const uint4*>(gx + item * NPad * F) : nullptr;
  uint4* rdst = (MODE == 1 && r_out) ? reinterpret_cast<uint4*>(r_out + item * NPad * F) : nullptr;
  uint16_t* hu = (MODE == 1 && Huser) ? Huser + item * hu_stride : nullptr;
  RowPre rpre;
  rpre.j0 = rpre.j1 = 0;
  if (tid < N) { rpre.j0 = rowptr[tid]; rpre.j1 = rowptr[tid + 1]; }
  int na = (nl < N) ? t_order[nl] : -1, nb = (NPP + nl < N) ? t_order[NPP + nl] : -1;
  __builtin_amdgcn_sched_barrier(0);
  constexpr int ZU = 8;
  const int ztotal = N * LPN;
  uint4 zv[ZU];
#pragma unroll
  for (int u = 0; u < ZU; ++u) {
    const int idx = tid + u * ETHREADS;
    zv[u] = idx < ztotal ? zsrc[idx] : uint4{0, 0, 0, 0};
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) rpre.nn[u] = (rpre.j0 + u < rpre.j1) ? r_edge[rpre.j0 + u].x : -1;
  auto in_edges = [&](int n, int& q0, int& deg) {
    q0 = 0; deg = 0;
    if (n >= 0) { q0 = t_rowptr[n]; deg = t_rowptr[n + 1] - q0; }
  };
  auto bounds = [&](int base, int& n, int& q0, int& deg) {
    n = (base + nl < N) ? t_order[base + nl] : -1;
    in_edges(n, q0, deg);
  };
  // records of a pass: lane p of a node's group holds the in-edges p, p + LPN, ... (MAXC chunks cover in-degrees <= 32 from
  // registers; the records of the NEXT pass are requested before this pass computes, so their latency is never exposed)
  constexpr int MAXC = 32 / LPN, GPC = LPN / 4;          // chunks per pass in registers; groups of 4 records per chunk
  auto load_recs = [&](int q0, int deg, int2* r) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) r[c] = (c * LPN + p < deg) ? t_edge[q0 + c * LPN + p] : int2{0, 0};
  };
  int q0a, dega, q0b, degb;
  in_edges(na, q0a, dega);
  in_edges(nb, q0b, degb);
  __builtin_amdgcn_sched_barrier(0);
  // ---- z -> LDS, scores ----------------------------------------------------------------------------------------------------------
  {
    f32x2 ar[8];                                      // (a1[f], a2[f]) pairs of this lane's 8 features
#pragma unroll
    for (int j = 0; j < 8; ++j) ar[j] = f32x2{a12[p * 8 + j], a12[F + p * 8 + j]};
    const int total = ztotal;
    for (int i0 = tid; i0 < total; i0 += ZU * ETHREADS) {
      uint4 v[ZU];
#pragma unroll
      for (int u = 0; u < ZU; ++u) {
        const int idx = i0 + u * ETHREADS;
        v[u] = (i0 == tid) ? zv[u] : (idx < total ? zsrc[idx] : uint4{0, 0, 0, 0});
      }
#pragma unroll
      for (int u = 0; u < ZU; ++u) {
        const int idx = i0 + u * ETHREADS;
        float f8[8];
        unpack8(v[u], f8);
        f32x2 d = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) d += ar[j] * f8[j];
#pragma unroll
        for (int off = 1; off < LPN; off <<= 1) { d.x += __shfl_xor(d.x, off, 64); d.y += __shfl_xor(d.y, off, 64); }
        if (idx < total) {
          zl[idx] = v[u];
          if (p == 0) sc[idx / LPN] = float4{d.x, d.y, 0.f, 0.f};
        }
      }
    }
  }
  int2 reca[MAXC];
  load_recs(q0a, dega, reca);
  uint4 gva = uint4{0, 0, 0, 0};
  if (MODE == 1 && na >= 0) gva = gsrc[na * LPN + p];
  __syncthreads();
  EA_STAMP(1);
  // ---- phase A: softmax statistics of every support row -----------------------------------------------------------------
  row_softmax_stats<ETHREADS, true>(sc, rowptr, r_edge, N, slope, tid, &rpre);
  __syncthreads();
  EA_STAMP(2);
  // ---- phase B: aggregation over the in-edges, epilogue; slot base + nl of a pass is node t_order[base + nl] ---------------
  uint4 keep[L::MAXPASS];      // MODE 1 with the user-layout copy: this lane's piece of h of every pass (no global round trip in the epilogue)
#pragma unroll
  for (int ps = 0; ps < L::MAXPASS; ++ps) keep[ps] = uint4{0, 0, 0, 0};
#pragma unroll
  for (int ps = 0; ps < L::MAXPASS; ++ps) {
    const int base = ps * NPP;
    if (base >= N) break;      // (workgroup-uniform)
    int2 recb[MAXC];
    load_recs(q0b, degb, recb);
    uint4 gvb = uint4{0, 0, 0, 0};
    if (MODE == 1 && nb >= 0) gvb = gsrc[nb * LPN + p];
    int nc, q0c, degc;
    bounds(base + 2 * NPP, nc, q0c, degc);
    const int n = na;
    const bool valid = n >= 0;
    const float s1n = valid ? sc[n].x : 0.f;
    int dmax = dega;
#pragma unroll
    for (int off = LPN; off < 64; off <<= 1) dmax = max(dmax, __shfl_xor(dmax, off, 64));
    dmax = __builtin_amdgcn_readfirstlane(dmax);
    f32x2 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x2{0.f, 0.f};
    // coefficient v alpha of this lane's records: once per record, broadcast below together with the row index
    auto coef = [&](const int2 r) {
      const float4 s = sc[r.x];
      float e = s1n + s.y;
      e = e > 0.f ? e : slope * e;
      const float v = __int_as_float(r.y);
      return (v != 0.f) ? v * eexp(e - s.z) * s.w : 0.f;
    };
    // (round 5: the broadcasts on the VALU's DPP path instead -- quad_perm, then row_shr / row_shl:4 into the other quad's bank -- measured no
    //  faster, 725 against 698 units for the phase: it is bound by vector issue, not by the LDS instructions)
    auto group4 = [&](const int mx, const float cx, const int off) {       // four records: shuffles, then the row reads, then the FMAs
      int mm[4];
      float cc[4];
      uint4 zr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { mm[i] = __shfl(mx, off + i, LPN); cc[i] = __shfl(cx, off + i, LPN); }
#pragma unroll
      for (int i = 0; i < 4; ++i) zr[i] = zl[mm[i] * LPN + p];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0] += cc[i] * f32x2{__uint_as_float(zr[i].x << 16), __uint_as_float(zr[i].x & 0xffff0000u)};
        acc[1] += cc[i] * f32x2{__uint_as_float(zr[i].y << 16), __uint_as_float(zr[i].y & 0xffff0000u)};
        acc[2] += cc[i] * f32x2{__uint_as_float(zr[i].z << 16), __uint_as_float(zr[i].z & 0xffff0000u)};
        acc[3] += cc[i] * f32x2{__uint_as_float(zr[i].w << 16), __uint_as_float(zr[i].w & 0xffff0000u)};
      }
    };
    float cm[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) cm[c] = (c * LPN < dmax) ? coef(reca[c]) : 0.f;
#pragma unroll
    for (int g = 0; g < MAXC * GPC; ++g)
      if (g * 4 < dmax) group4(reca[g / GPC].x, cm[g / GPC], (g % GPC) * 4);
    for (int e0 = MAXC * LPN; e0 < dmax; e0 += LPN) {                      // in-degrees beyond the register chunks (hubs)
      const int2 r = (e0 + p < dega) ? t_edge[q0a + e0 + p] : int2{0, 0};
      const float cx = coef(r);
#pragma unroll
      for (int g = 0; g < GPC; ++g) group4(r.x, cx, g * 4);
    }
    float o8[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { o8[2 * j] = fmaxf(acc[j].x, 0.f); o8[2 * j + 1] = fmaxf(acc[j].y, 0.f); }   // the layer's ReLU (graphML.py:2101)
    if (MODE == 0) {
      if (valid) oseq[n * LPN + p] = pack8(o8);
    } else {
      if (valid && rdst) rdst[n * LPN + p] = pack8(o8);
      float g8[8];
      unpack8(gva, g8);
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] = etanh(giv * g8[j] + gfv * o8[j]);
      const uint4 hv = pack8(o8);
      if (valid) { oseq[n * LPN + p] = hv; keep[ps] = hv; }
    }
    na = nb; q0a = q0b; dega = degb; gva = gvb; nb = nc; q0b = q0c; degb = degc;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) reca[c] = recb[c];
  }
  EA_STAMP(3);
  if (MODE == 1 && hu) {
    // ---- user layout H[f][n] (round 5): the pieces kept in registers go into a transposed LDS image over the dead z / sc region as 32-bit words
    // {feature 2q, feature 2q + 1} of a node -- a piece is four such words as it is --, then every feature row is read back along n (two
    // 16-byte reads + 4 v_perm_b32 pick the row's half of 8 words) and stored in 16-byte pieces. (Rounds 2-4: the rows just stored were read
    // back from global memory behind a draining barrier and scattered with 8 ds_write_b16 per piece: 325 of the kernel's 1,075 units.)
    constexpr int WP = L::WPITCH;
    uint32_t* imgw = reinterpret_cast<uint32_t*>(smem);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has left z and sc (LDS only: the stores above stay in flight)
    EA_STAMP(4);
#pragma unroll
    for (int ps = 0; ps < L::MAXPASS; ++ps) {
      const int slot = ps * NPP + nl;
      if (slot < N) {
        const int n = t_order[slot];
        const uint32_t w4[4] = {keep[ps].x, keep[ps].y, keep[ps].z, keep[ps].w};
        const int nsw = n ^ (8 * (p & 3));                  // row q = 4 p + j: q >> 2 = p
#pragma unroll
        for (int j = 0; j < 4; ++j) imgw[(4 * p + j) * WP + nsw] = w4[j];
      }
    }
    __syncthreads();
    EA_STAMP(5);
    const int ppr = N / 8;                                                 // 16-byte pieces per feature row
    for (int idx = tid; idx < F * ppr; idx += ETHREADS) {
      const int f = idx / ppr, pc = idx - f * ppr;
      const int q2 = f >> 1;
      const uint4* src = reinterpret_cast<const uint4*>(imgw + q2 * WP + 8 * (pc ^ ((q2 >> 2) & 3)));
      const uint4 w0 = src[0], w1 = src[1];
      const uint32_t sel = (f & 1) ? 0x07060302u : 0x05040100u;
      uint4 o;
      o.x = __builtin_amdgcn_perm(w0.y, w0.x, sel);
      o.y = __builtin_amdgcn_perm(w0.w, w0.z, sel);
      o.z = __builtin_amdgcn_perm(w1.y, w1.x, sel);
      o.w = __builtin_amdgcn_perm(w1.w, w1.z, sel);
      *reinterpret_cast<uint4*>(hu + (int64_t)f * N + pc * 8) = o;
    }
  }
  EA_STAMP(6);
  for (int idx = N * LPN + tid; idx < NPad * LPN; idx += ETHREADS) {       // padding rows of the sequence-major output: zeros
    oseq[idx] = uint4{0, 0, 0, 0};
    if (rdst) rdst[idx] = uint4{0, 0, 0, 0};
  }
}

template <int F>
static int edge_att_fwd_t(const void* z, const float* a12, const void* gx, const float* gi, const float* gf, const int32_t* rowptr,
                          const void* r_edge, const int32_t* t_rowptr, const void* t_edge, const int32_t* t_order, void* out_seq, void* r_out, void* Huser,
                          int64_t hu_stride, int64_t items, int64_t N, int64_t NPad, float slope, hipStream_t st) {
  const size_t lds = EdgeLds<F>::fwd_bytes((int)N);
  if (lds > 160 * 1024) return GCRNN_ERR_UNSUPPORTED;
  auto k0 = edge_att_fwd_kernel<F, 0>;
  auto k1 = edge_att_fwd_kernel<F, 1>;
  auto kern = gx ? k1 : k0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)items, ETHREADS, lds, st>>>((const uint16_t*)z, a12, (const uint16_t*)gx, gi, gf, rowptr, (const int2*)r_edge, t_rowptr,
                                               (const int2*)t_edge, t_order, (uint16_t*)out_seq, (uint16_t*)r_out, (uint16_t*)Huser, hu_stride,
                                               (int)N, (int)NPad, slope);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}


// ------------------------------------------------------------------------------------------
// Backward of one attention (autograd of graphAttention, graphML.py:585-627, composed with the layer's ReLU and the branch's gate):
//   do[n]      = g dpre[n] . [r[n] > 0]                      r = relu(att(z)) kept by the forward, g the branch's scalar time gate
//   d alpha    = v (z_m . do[n])                              per support edge (m -> n)
//   dz_m       = sum_{n in row m} v alpha do[n]               direct path, the SAME gathered rows do[n] serve both
//   softmax:     dl[m][n] = alpha (d alpha - R_m) lrelu'(s1[n] + s2[m]),  R_m = sum_n alpha d alpha
//   scores:      ds2[m] = sum_n dl[m][n] (row sum),  ds1[n] = sum_m dl[m][n] (column sum, through the per-edge scratch E)
//   dz_m      += a1 ds1[m] + a2 ds2[m];   da1 = sum_n ds1[n] z_n,  da2 = sum_m ds2[m] z_m   (per-item partials)
// One workgroup per item. LDS: the do image [N][F] bf16 and sc[n] = {s1, s2, row max -> ds2, 1 / row sum -> ds1}. Rows are
// visited by descending out-degree; a row's F/8 lanes hold its z piece and its edge records in registers (hub rows with more than 32
// records continue chunk by chunk, d alpha parked in the scratch).
// ------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int LPN> __device__ __forceinline__ float group_sum(float v) {       // sum over the LPN (4 or 8) aligned lanes of a node
  v += dpp_f<0xB1>(v);                 // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);                 // quad_perm [2,3,0,1]
  if (LPN == 8) v += dpp_f<0x141>(v);  // row_half_mirror: the other quad's total
  return v;
}

template <int F, bool HUB>
__global__ __launch_bounds__(HUB ? 512 : 1024) void edge_att_bwd_kernel(
    const uint16_t* __restrict__ dpre,       // [items][NPad][F] bf16
    const uint16_t* __restrict__ r,          // [items][NPad][F] bf16: relu(att(z)) of this branch
    const float* __restrict__ g,             // [items] scalar gate of this branch, or null (= 1)
    const uint16_t* __restrict__ z,          // [items][NPad][F] bf16
    const float* __restrict__ a12,           // [2][F]
    const int32_t* __restrict__ rowptr, const int2* __restrict__ r_edge, const int32_t* __restrict__ r_order,
    const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_pos,
    float* __restrict__ E,                   // [items][nnz] scratch: dl per support edge, row order
    uint16_t* __restrict__ dz,               // [items][NPad][F] bf16 (rows >= N written as zeros)
    float* __restrict__ da_part,             // [items][2][F]
    float* __restrict__ dgate,               // [items]: sum dpre . r (the gradient of g), or null
    int N, int NPad, int nnz, float slope) {
  // HUB: rows with more than 32 records exist: their extra chunks need registers the lean variant (16 waves) does not have -> 8 waves
  constexpr int BT = HUB ? 512 : 1024;
  using L = EdgeLds<F>;
  constexpr int LPN = L::LPN, NPP = BT / LPN;
  constexpr int MAXC = 32 / LPN, GPC = LPN / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* dol = reinterpret_cast<uint4*>(smem);
  float4* sc = reinterpret_cast<float4*>(smem + (size_t)N * F * 2);

  const int tid = threadIdx.x;
  const int p = tid % LPN, nl = tid / LPN;
  const int64_t item = blockIdx.x;
  const uint4* zsrc = reinterpret_cast<const uint4*>(z + item * NPad * F);
  const uint4* dsrc = reinterpret_cast<const uint4*>(dpre + item * NPad * F);
  const uint4* rsrc = reinterpret_cast<const uint4*>(r + item * NPad * F);
  uint4* dzd = reinterpret_cast<uint4*>(dz + item * NPad * F);
  float* Ei = E + item * nnz;
  const float gv = g ? g[item] : 1.f;
  const int total = N * LPN;

  float a1r[8], a2r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a1r[j] = a12[p * 8 + j]; a2r[j] = a12[F + p * 8 + j]; }
  // ---- phase 0: do image, scores, gate gradient -------------------------------------------------------------------------
  float gsum = 0.f;
  for (int i0 = tid; i0 < total; i0 += 2 * BT) {
    uint4 vz[2], vd[2], vr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = i0 + u * BT;
      const bool ok = idx < total;
      vz[u] = ok ? zsrc[idx] : uint4{0, 0, 0, 0};
      vd[u] = ok ? dsrc[idx] : uint4{0, 0, 0, 0};
      vr[u] = ok ? rsrc[idx] : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = i0 + u * BT;
      float zf[8], df[8], rf[8];
      unpack8(vz[u], zf);
      unpack8(vd[u], df);
      unpack8(vr[u], rf);
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        d1 += a1r[j] * zf[j];
        d2 += a2r[j] * zf[j];
        gsum += df[j] * rf[j];
        df[j] = rf[j] > 0.f ? gv * df[j] : 0.f;
      }
      d1 = group_sum<LPN>(d1);
      d2 = group_sum<LPN>(d2);
      if (idx < total) {
        dol[idx] = pack8(df);
        if (p == 0) sc[idx / LPN] = float4{d1, d2, 0.f, 0.f};
      }
    }
  }
  __syncthreads();
  // ---- phase A: softmax statistics of every support row ------------------------------------------------------------------
  row_softmax_stats<BT>(sc, rowptr, r_edge, N, slope, tid);
  __syncthreads();
  // ---- phase R: rows by descending out-degree ----------------------------------------------------------------------------
  auto bounds = [&](int base, int& m, int& j0, int& deg) {
    m = -1; j0 = 0; deg = 0;
    if (base + nl < N) { m = r_order[base + nl]; j0 = rowptr[m]; deg = rowptr[m + 1] - j0; }
  };
  auto load_recs = [&](int j0, int deg, int2* rc) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) rc[c] = (c * LPN + p < deg) ? r_edge[j0 + c * LPN + p] : int2{0, 0};
  };
  int ma, j0a, dega, mb, j0b, degb;
  bounds(0, ma, j0a, dega);
  bounds(NPP, mb, j0b, degb);
  int2 reca[MAXC];
  load_recs(j0a, dega, reca);
  uint4 zva = uint4{0, 0, 0, 0};
  if (ma >= 0) zva = zsrc[ma * LPN + p];
  for (int base = 0; base < N; base += NPP) {
    int2 recb[MAXC];
    load_recs(j0b, degb, recb);
    uint4 zvb = uint4{0, 0, 0, 0};
    if (mb >= 0) zvb = zsrc[mb * LPN + p];
    int mc, j0c, degc;
    bounds(base + 2 * NPP, mc, j0c, degc);
    const int m = ma;
    const bool valid = m >= 0;
    float4 sm = float4{0.f, 0.f, 0.f, 0.f};
    if (valid) sm = sc[m];
    int dmax = dega;
#pragma unroll
    for (int off = LPN; off < 64; off <<= 1) dmax = max(dmax, __shfl_xor(dmax, off, 64));
    dmax = __builtin_amdgcn_readfirstlane(dmax);
    float zf[8];
    unpack8(zva, zf);
    // this lane's records: alpha, v alpha and the LeakyReLU slope factor
    float al[MAXC], cown[MAXC], lr[MAXC], dal[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      al[c] = 0.f; cown[c] = 0.f; lr[c] = 1.f; dal[c] = 0.f;
      if (c * LPN < dmax) {
        const float v = __int_as_float(reca[c].y);
        float e = sc[reca[c].x].x + sm.y;
        lr[c] = e > 0.f ? 1.f : slope;
        e *= lr[c];
        al[c] = (v != 0.f) ? eexp(e - sm.z) * sm.w : 0.f;
        cown[c] = v * al[c];
      }
    }
    f32x2 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
    for (int gq = 0; gq < MAXC * GPC; ++gq) {
      if (gq * 4 < dmax) {
        const int c = gq / GPC, off = (gq % GPC) * 4;
        int nn[4];
        float cc[4];
        uint4 dr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { nn[i] = __shfl(reca[c].x, off + i, LPN); cc[i] = __shfl(cown[c], off + i, LPN); }
#pragma unroll
        for (int i = 0; i < 4; ++i) dr[i] = dol[nn[i] * LPN + p];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float df[8];
          unpack8(dr[i], df);
          float dot = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) dot += zf[j] * df[j];
          dot = group_sum<LPN>(dot);
          if (p == off + i) dal[c] = dot;                     // the lane that owns this record keeps z_m . do[n]
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += cc[i] * f32x2{df[2 * j], df[2 * j + 1]};
        }
      }
    }
    // softmax / LeakyReLU backward on this lane's records
    float rsum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { dal[c] *= __int_as_float(reca[c].y); rsum += al[c] * dal[c]; }
    // hub rows (out-degree beyond the register chunks): the same gather chunk by chunk; d alpha of those records waits in the
    // scratch (each lane reads back only what it wrote itself) until the row sum R is complete
    if constexpr (HUB)
    for (int e0 = MAXC * LPN; e0 < dmax; e0 += LPN) {
      const int2 rc = (e0 + p < dega) ? r_edge[j0a + e0 + p] : int2{0, 0};
      const float v = __int_as_float(rc.y);
      float e = sc[rc.x].x + sm.y;
      e = e > 0.f ? e : slope * e;
      const float alx = (v != 0.f) ? eexp(e - sm.z) * sm.w : 0.f;
      const float cx = v * alx;
      float dmine = 0.f;
#pragma unroll
      for (int i = 0; i < LPN; ++i) {
        const int nn = __shfl(rc.x, i, LPN);
        const float cc = __shfl(cx, i, LPN);
        float df[8];
        unpack8(dol[nn * LPN + p], df);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += zf[j] * df[j];
        dot = group_sum<LPN>(dot);
        if (p == i) dmine = dot;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += cc * f32x2{df[2 * j], df[2 * j + 1]};
      }
      dmine *= v;
      rsum += alx * dmine;
      if (e0 + p < dega) Ei[j0a + e0 + p] = dmine;
    }
    rsum = group_sum<LPN>(rsum);
    float ds2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const float dl = al[c] * (dal[c] - rsum) * lr[c];
      ds2 += dl;
      if (c * LPN + p < dega) Ei[j0a + c * LPN + p] = dl;
    }
    if constexpr (HUB)
    for (int e0 = MAXC * LPN; e0 < dmax; e0 += LPN) {
      if (e0 + p < dega) {
        const int2 rc = r_edge[j0a + e0 + p];
        float e = sc[rc.x].x + sm.y;
        const float lrx = e > 0.f ? 1.f : slope;
        e *= lrx;
        const float alx = eexp(e - sm.z) * sm.w;
        const float dl = alx * (Ei[j0a + e0 + p] - rsum) * lrx;
        ds2 += dl;
        Ei[j0a + e0 + p] = dl;
      }
    }
    ds2 = group_sum<LPN>(ds2);
    if (valid) {
      if (p == 0) sc[m].z = ds2;                              // the row statistics of m are in this group's registers: the slot is free
      float o8[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { o8[2 * j] = acc[j].x + a2r[2 * j] * ds2; o8[2 * j + 1] = acc[j].y + a2r[2 * j + 1] * ds2; }
      dzd[m * LPN + p] = pack8(o8);
    }
    ma = mb; j0a = j0b; dega = degb; zva = zvb; mb = mc; j0b = j0c; degb = degc;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) reca[c] = recb[c];
  }
  __threadfence_block();
  __syncthreads();
  // ---- phase C: ds1[n] = column sums of dl through the scratch (this workgroup's own stores) --------------------------------
  for (int n = tid; n < N; n += BT) {
    float acc1 = 0.f;
    const int q1 = t_rowptr[n + 1];
    for (int q = t_rowptr[n]; q < q1; q += 8) {             // eight positions, then eight scratch reads: two latencies per chunk
      int tp[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) tp[u] = (q + u < q1) ? t_pos[q + u] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc1 += tp[u] >= 0 ? Ei[tp[u]] : 0.f;
    }
    sc[n].w = acc1;
  }
  __syncthreads();
  // ---- phase F: dz += a1 ds1, per-item partials of da ----------------------------------------------------------------------
  f32x2 dacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dacc[j] = f32x2{0.f, 0.f};
  for (int i0 = tid; i0 < total; i0 += 2 * BT) {        // two rows per lane and trip: their loads are issued before the first store
    uint4 vz[2], vp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = i0 + u * BT;
      vz[u] = idx < total ? zsrc[idx] : uint4{0, 0, 0, 0};
      vp[u] = idx < total ? dzd[idx] : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = i0 + u * BT;
      if (idx < total) {
        const float4 s = sc[idx / LPN];
        float zf[8], pf[8];
        unpack8(vz[u], zf);
        unpack8(vp[u], pf);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pf[j] += a1r[j] * s.w;
          dacc[j] += f32x2{s.w, s.z} * zf[j];
        }
        dzd[idx] = pack8(pf);
      }
    }
  }
  for (int idx = total + tid; idx < NPad * LPN; idx += BT) dzd[idx] = uint4{0, 0, 0, 0};
  // block reductions (fixed order): da over the nodes, the gate gradient over everything; the do image is dead
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                 // [BT / LPN][LPN * 16] da partials, then [BT] gate partials
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[(nl * LPN + p) * 16 + j] = dacc[j].x; red[(nl * LPN + p) * 16 + 8 + j] = dacc[j].y; }
  float* gred = red + BT * 16;
  gred[tid] = gsum;
  __syncthreads();
  if (tid < 2 * F) {
    const int which = tid / F, f = tid % F, pp = f / 8, j = f % 8;
    float sum = 0.f;
    for (int k = 0; k < BT / LPN; ++k) sum += red[(k * LPN + pp) * 16 + which * 8 + j];
    da_part[item * 2 * F + tid] = sum;
  }
  if (dgate && tid == 2 * F) {
    float sum = 0.f;
    for (int k = 0; k < BT; ++k) sum += gred[k];
    dgate[item] = sum;
  }
}

template <int F>
static int edge_att_bwd_t(const void* dpre, const void* r, const float* g, const void* z, const float* a12, const int32_t* rowptr,
                          const void* r_edge, const int32_t* r_order, const int32_t* t_rowptr, const int32_t* t_pos, float* E, void* dz,
                          float* da_part, float* dgate, int64_t items, int64_t N, int64_t NPad, int64_t nnz, float slope, bool hub, hipStream_t st) {
  size_t lds = (size_t)N * F * 2 + (size_t)N * 16;
  const int bt = hub ? 512 : 1024;
  const size_t red = (size_t)bt * 17 * 4;
  if (lds < red) lds = red;
  if (lds > 160 * 1024) return GCRNN_ERR_UNSUPPORTED;
  auto kern = hub ? edge_att_bwd_kernel<F, true> : edge_att_bwd_kernel<F, false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)items, bt, lds, st>>>((const uint16_t*)dpre, (const uint16_t*)r, g, (const uint16_t*)z, a12, rowptr, (const int2*)r_edge,
                                               r_order, t_rowptr, t_pos, E, (uint16_t*)dz, da_part, dgate, (int)N, (int)NPad, (int)nnz, slope);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

}  // namespace

extern "C" int gcrnn_fused_edge_attention_supported(int64_t N, int64_t F) {
  if (N <= 0 || N % 8 || N > 1024 || (F != 32 && F != 64)) return 0;      // (N <= 1024: a lane keeps its pieces of h of at most 1024 / (nodes per pass) passes)
  const size_t lds = F == 64 ? EdgeLds<64>::fwd_bytes((int)N) : EdgeLds<32>::fwd_bytes((int)N);
  return lds <= 160 * 1024;
}

extern "C" int gcrnn_fused_edge_attention_bf16(const void* z, const float* a12, const void* gx, const float* gi, const float* gf,
                                               const int32_t* rowptr, const void* r_edge, const int32_t* t_rowptr, const void* t_edge,
                                               const int32_t* t_order, void* out_seq, void* r_out, void* Huser, int64_t huser_item_stride, int64_t items,
                                               int64_t N, int64_t NPad, int64_t F, double negative_slope, void* stream) {
  if (!z || !a12 || !rowptr || !r_edge || !t_rowptr || !t_edge || !t_order || !out_seq) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (!gx && (gi || r_out || Huser)) return GCRNN_ERR_BAD_SHAPE;
  if (items <= 0 || items > 2147483647LL || N <= 0 || N > NPad || N % 8) return GCRNN_ERR_BAD_SHAPE;
  if (Huser && (reinterpret_cast<uintptr_t>(Huser) & 15 || huser_item_stride % 8 || out_seq == z)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
  if (F == 64) return edge_att_fwd_t<64>(z, a12, gx, gi, gf, rowptr, r_edge, t_rowptr, t_edge, t_order, out_seq, r_out, Huser, huser_item_stride, items, N, NPad, (float)negative_slope, st);
  if (F == 32) return edge_att_fwd_t<32>(z, a12, gx, gi, gf, rowptr, r_edge, t_rowptr, t_edge, t_order, out_seq, r_out, Huser, huser_item_stride, items, N, NPad, (float)negative_slope, st);
  return GCRNN_ERR_UNSUPPORTED;
}

// Backward of gcrnn_fused_edge_attention_bf16 for one branch: dpre = d loss / d (pre-activation of the cell) [items][NPad][F] bf16,
// r = relu(att(z)) kept by the forward, g [items] the branch's scalar time gate (or NULL = 1) -> dz [items][NPad][F] bf16 (the
// gradient w.r.t. the composite filter output z), da_part fp32 [items][2][F] (per-item partials of the mixer gradient: the caller
// adds the items in a fixed order), dgate fp32 [items] = sum dpre . r (or NULL). r_order = support rows by descending out-degree,
// t_pos = position of every column-ordered support edge in the row order, E = fp32 scratch [items][nnz]. Rows with more than 32
// support entries (hubs) take a slower chunked loop.
extern "C" int gcrnn_fused_edge_attention_backward_supported(int64_t N, int64_t F, int64_t max_out_degree) {
  (void)max_out_degree;      // rows beyond 32 records take a slower chunked loop (their d alpha waits in the scratch)
  return gcrnn_fused_edge_attention_supported(N, F);
}

extern "C" int gcrnn_fused_edge_attention_backward_bf16(const void* dpre, const void* r, const float* g, const void* z, const float* a12,
                                                        const int32_t* rowptr, const void* r_edge, const int32_t* r_order,
                                                        const int32_t* t_rowptr, const int32_t* t_pos, float* scratch, void* dz,
                                                        float* da_part, float* dgate, int64_t items, int64_t N, int64_t NPad, int64_t F,
                                                        int64_t nnz, int64_t max_out_degree, double negative_slope, void* stream) {
  if (!dpre || !r || !z || !a12 || !rowptr || !r_edge || !r_order || !t_rowptr || !t_pos || !scratch || !dz || !da_part) return GCRNN_ERR_NULL_POINTER;
  if (items <= 0 || items > 2147483647LL || N <= 0 || N > NPad || N % 8 || nnz <= 0 || items * nnz > (1LL << 40)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
  if (F == 64) return edge_att_bwd_t<64>(dpre, r, g, z, a12, rowptr, r_edge, r_order, t_rowptr, t_pos, scratch, dz, da_part, dgate, items, N, NPad, nnz, (float)negative_slope, max_out_degree > 32 || F == 32 /* (the lean F = 32 instantiation would spill: eight record chunks) */, st);
  if (F == 32) return edge_att_bwd_t<32>(dpre, r, g, z, a12, rowptr, r_edge, r_order, t_rowptr, t_pos, scratch, dz, da_part, dgate, items, N, NPad, nnz, (float)negative_slope, max_out_degree > 32 || F == 32 /* (the lean F = 32 instantiation would spill: eight record chunks) */, st);
  return GCRNN_ERR_UNSUPPORTED;
}
