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

namespace {

#ifndef GCRNN_EDGE_THREADS
#define GCRNN_EDGE_THREADS 1024
#endif
constexpr int ETHREADS = GCRNN_EDGE_THREADS;      // 16 waves: the aggregation is a chain of dependent LDS reads, occupancy hides it

__device__ __forceinline__ float ebf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t ef2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
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
  v.x = (uint32_t)ef2bf(o[0]) | ((uint32_t)ef2bf(o[1]) << 16);
  v.y = (uint32_t)ef2bf(o[2]) | ((uint32_t)ef2bf(o[3]) << 16);
  v.z = (uint32_t)ef2bf(o[4]) | ((uint32_t)ef2bf(o[5]) << 16);
  v.w = (uint32_t)ef2bf(o[6]) | ((uint32_t)ef2bf(o[7]) << 16);
  return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS carve-up: rows of 8-feature pieces, then the per-node scalars sc[n] = {s1, s2, row max, 1 / row sum}; once the aggregation
// is done the same bytes hold the transposed image [F][N + 64] of h for the user-layout store (row f shifted by 8 (f >> 3)
// columns, which spreads the eight feature groups of a wave's ds_write_b16 over all banks)
template <int F> struct EdgeLds {
  static constexpr int LPN = F / 8;                  // lanes per node = 16-byte pieces of a bf16 row
  static constexpr int NPP = ETHREADS / LPN;         // nodes per pass
  static size_t fwd_bytes(int N) {
    const size_t a = (size_t)N * F * 2 + (size_t)N * 16, b = (size_t)F * (N + 64) * 2;
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

  // ---- load z, scores (four 16-byte loads in flight per lane) ----------------------------------------------------------
  {
    f32x2 ar[8];                                      // (a1[f], a2[f]) pairs of this lane's 8 features
#pragma unroll
    for (int j = 0; j < 8; ++j) ar[j] = f32x2{a12[p * 8 + j], a12[F + p * 8 + j]};
    const int total = N * LPN;
    for (int i0 = tid; i0 < total; i0 += 4 * ETHREADS) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * ETHREADS;
        v[u] = idx < total ? zsrc[idx] : uint4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
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
  __syncthreads();
  // ---- phase A: softmax statistics of every support row (one thread per row, online max / sum) ----------------------------
  for (int m = tid; m < N; m += ETHREADS) {
    const int j0 = rowptr[m], j1 = rowptr[m + 1];
    const float s2m = sc[m].y;
    float mx = -1e30f, sum = 0.f;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
      float e = sc[r_edge[j].x].x + s2m;
      e = e > 0.f ? e : slope * e;
      const float nm = fmaxf(mx, e);
      sum = sum * eexp(mx - nm) + eexp(e - nm);
      mx = nm;
    }
    sc[m].z = mx;
    sc[m].w = sum > 0.f ? 1.f / sum : 0.f;
  }
  __syncthreads();
  // ---- phase B: aggregation over the in-edges, epilogue; slot base + nl of a pass is node t_order[base + nl] ---------------
  float giv = 1.f, gfv = 1.f;
  if (MODE == 1 && gi) { giv = gi[item]; gfv = gf[item]; }
  uint4* oseq = reinterpret_cast<uint4*>(out_seq + item * NPad * F);
  const uint4* gsrc = MODE == 1 ? reinterpret_cast<const uint4*>(gx + item * NPad * F) : nullptr;
  uint4* rdst = (MODE == 1 && r_out) ? reinterpret_cast<uint4*>(r_out + item * NPad * F) : nullptr;
  uint16_t* hu = (MODE == 1 && Huser) ? Huser + item * hu_stride : nullptr;
  auto bounds = [&](int base, int& n, int& q0, int& deg) {
    n = -1; q0 = 0; deg = 0;
    if (base + nl < N) { n = t_order[base + nl]; q0 = t_rowptr[n]; deg = t_rowptr[n + 1] - q0; }
  };
  // records of a pass: lane p of a node's group holds the in-edges p, p + LPN, ... (MAXC chunks cover in-degrees <= 32 from
  // registers; the records of the NEXT pass are requested before this pass computes, so their latency is never exposed)
  constexpr int MAXC = 32 / LPN, GPC = LPN / 4;          // chunks per pass in registers; groups of 4 records per chunk
  auto load_recs = [&](int q0, int deg, int2* r) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) r[c] = (c * LPN + p < deg) ? t_edge[q0 + c * LPN + p] : int2{0, 0};
  };
  int na, q0a, dega, nb, q0b, degb;
  bounds(0, na, q0a, dega);
  bounds(NPP, nb, q0b, degb);
  int2 reca[MAXC];
  load_recs(q0a, dega, reca);
  uint4 gva = uint4{0, 0, 0, 0};
  if (MODE == 1 && na >= 0) gva = gsrc[na * LPN + p];
  for (int base = 0; base < N; base += NPP) {
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
      if (valid) oseq[n * LPN + p] = pack8(o8);
    }
    na = nb; q0a = q0b; dega = degb; gva = gvb; nb = nc; q0b = q0c; degb = degc;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) reca[c] = recb[c];
  }
  if (MODE == 1 && hu) {
    // ---- user layout H[f][n]: the rows just written are read back (this workgroup's own stores, ordered by the barrier) into a
    // transposed LDS image over the dead z / sc region, then stored along n in 16-byte pieces
    constexpr int IST_PAD = 64;
    const int IST = N + IST_PAD;
    uint16_t* img = reinterpret_cast<uint16_t*>(smem);
    __syncthreads();
    const int total = N * LPN;
    for (int i0 = tid; i0 < total; i0 += 4 * ETHREADS) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * ETHREADS;
        v[u] = idx < total ? oseq[idx] : uint4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * ETHREADS;
        if (idx < total) {
          const int n = idx / LPN;
          const uint32_t w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int j = 0; j < 8; ++j)
            img[(p * 8 + j) * IST + 8 * p + n] = (uint16_t)(j & 1 ? w4[j >> 1] >> 16 : w4[j >> 1] & 0xffffu);
        }
      }
    }
    __syncthreads();
    const int ppr = N / 8;                                                 // 16-byte pieces per feature row
    for (int idx = tid; idx < F * ppr; idx += ETHREADS) {
      const int f = idx / ppr, pc = idx - f * ppr;
      *reinterpret_cast<uint4*>(hu + (int64_t)f * N + pc * 8) = *reinterpret_cast<const uint4*>(img + f * IST + 8 * (f >> 3) + pc * 8);
    }
  }
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

}  // namespace

extern "C" int gcrnn_fused_edge_attention_supported(int64_t N, int64_t F) {
  if (N <= 0 || N % 8 || (F != 32 && F != 64)) return 0;
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
