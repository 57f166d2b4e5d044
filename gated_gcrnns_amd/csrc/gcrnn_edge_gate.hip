// Edge gates of the fused path (reference Utils/graphML.py:2409-2416): the edge-gated cell is
//     h_t = tanh( gi att_in(A(S)x_t + b) + gf att_f(B(S)h_{t-1} + b) ),    att = GraphAttentional(F, F, 1 head) (graphML.py:1999-2128)
//     att(y)[n] = relu( sum_m z_m (S+I)[m][n] alpha[m][n] ),   z = W y,   alpha[m][.] = softmax over the support row m of
//                 LeakyReLU(a1.z_n + a2.z_m)                               (graphAttention, graphML.py:521-627)
// The mixing matrix W acts on features and the shift on nodes, so z = W(sum_k S^k u C_k + b) = sum_k S^k u (C_k W^T) + W b is a
// filter output with composite taps: the step kernel's filter-output pass (gcrnn_fused_filter_output_bf16) produces z directly and
// this file only holds what is left -- the attention itself, one workgroup per item (t, b):
//   load    z [N][F] bf16 -> LDS, scores s1[n] = a1.z_n, s2[n] = a2.z_n on the way (8 lanes per row, xor-shuffle reduction)
//   phase A per support row m: running max and sum of exp of e[m][n] (online softmax, F/8 lanes per row)
//   phase B per node n: o[n] = sum over the in-edges (m -> n) of v alpha z_m -- F/8 lanes per node, each owning 8 features; the
//           edge records {m, v} of a node are fetched by its lanes together (one coalesced load) and broadcast with shuffles;
//           alpha is recomputed from s1, s2 and the row statistics (nothing of size nnz is stored in inference)
//   epilogue  pre-pass: relu(o) (bf16, sequence-major);  step: h = tanh(gi gx + gf relu(o)) stored sequence-major for the next
//           step's filter pass and, through a transposed LDS tile, in the user layout H[b][t][f][:].
// Every sum is a gather in a fixed order: results are deterministic.
#include "gcrnn_common.h"

namespace {

constexpr int ETHREADS = 512;

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

// LDS carve-up shared by the forward and the backward kernel: rows of 8-feature pieces, then the per-node scalars
//   sc[n] = {s1, s2, row max, 1 / row sum}
template <int F> struct EdgeLds {
  static constexpr int LPN = F / 8;                  // lanes per node = 16-byte pieces of a bf16 row
  static constexpr int NPP = ETHREADS / LPN;         // nodes per pass
  static constexpr int TSTR = NPP + 8;               // row stride (elements) of the transposed output tile: 16-byte aligned rows
  static size_t fwd_bytes(int N) { return (size_t)N * F * 2 + (size_t)N * 16 + (size_t)F * TSTR * 2; }
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
    uint16_t* __restrict__ out_seq,          // [items][NPad][F] bf16 (rows >= N are written as zeros)
    uint16_t* __restrict__ r_out,            // MODE 1, training: relu(att(z)) [items][NPad][F] bf16, or null
    uint16_t* __restrict__ Huser,            // MODE 1: user-layout output of item 0, element (f, n) at f * N + n; or null
    int64_t hu_stride,                       // elements between the user-layout blocks of consecutive items
    int N, int NPad, float slope) {
  using L = EdgeLds<F>;
  constexpr int LPN = L::LPN, NPP = L::NPP, TSTR = L::TSTR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* zl = reinterpret_cast<uint4*>(smem);
  float4* sc = reinterpret_cast<float4*>(smem + (size_t)N * F * 2);
  uint16_t* tile = reinterpret_cast<uint16_t*>(sc + N);

  const int tid = threadIdx.x;
  const int p = tid % LPN, nl = tid / LPN;
  const int64_t item = blockIdx.x;
  const uint4* zsrc = reinterpret_cast<const uint4*>(z + item * NPad * F);

  // ---- load z, scores -------------------------------------------------------------------------------------------------
  {
    float a1r[8], a2r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a1r[j] = a12[p * 8 + j]; a2r[j] = a12[F + p * 8 + j]; }
    for (int idx = tid; idx < N * LPN; idx += ETHREADS) {
      const uint4 v = zsrc[idx];
      zl[idx] = v;
      float f8[8];
      unpack8(v, f8);
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { d1 += a1r[j] * f8[j]; d2 += a2r[j] * f8[j]; }
#pragma unroll
      for (int off = 1; off < LPN; off <<= 1) { d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64); }
      if (p == 0) sc[idx / LPN] = float4{d1, d2, 0.f, 0.f};
    }
  }
  __syncthreads();
  // ---- phase A: softmax statistics of every support row ------------------------------------------------------------------
  for (int base = 0; base < N; base += NPP) {
    const int m = base + nl;
    const bool valid = m < N;
    const int j0 = valid ? rowptr[m] : 0, j1 = valid ? rowptr[m + 1] : 0;
    const float s2m = valid ? sc[m].y : 0.f;
    float mx = -INFINITY, sum = 0.f;
    for (int j = j0 + p; j < j1; j += LPN) {
      const int n = r_edge[j].x;
      float e = sc[n].x + s2m;
      e = e > 0.f ? e : slope * e;
      if (e > mx) { sum = sum * eexp(mx - e) + 1.f; mx = e; }
      else sum += eexp(e - mx);
    }
#pragma unroll
    for (int off = 1; off < LPN; off <<= 1) {
      const float omx = __shfl_xor(mx, off, 64), osum = __shfl_xor(sum, off, 64);
      const float nm = fmaxf(mx, omx);
      if (nm > -INFINITY) sum = sum * eexp(mx - nm) + osum * eexp(omx - nm);
      mx = nm;
    }
    if (valid && p == 0) { sc[m].z = mx; sc[m].w = sum > 0.f ? 1.f / sum : 0.f; }
  }
  __syncthreads();
  // ---- phase B: aggregation over the in-edges, epilogue ------------------------------------------------------------------
  float giv = 1.f, gfv = 1.f;
  if (MODE == 1 && gi) { giv = gi[item]; gfv = gf[item]; }
  uint4* oseq = reinterpret_cast<uint4*>(out_seq + item * NPad * F);
  const uint4* gsrc = MODE == 1 ? reinterpret_cast<const uint4*>(gx + item * NPad * F) : nullptr;
  uint4* rdst = (MODE == 1 && r_out) ? reinterpret_cast<uint4*>(r_out + item * NPad * F) : nullptr;
  uint16_t* hu = (MODE == 1 && Huser) ? Huser + item * hu_stride : nullptr;
  for (int base = 0; base < N; base += NPP) {
    const int n = base + nl;
    const bool valid = n < N;
    const int q0 = valid ? t_rowptr[n] : 0, deg = valid ? t_rowptr[n + 1] - q0 : 0;
    const float s1n = valid ? sc[n].x : 0.f;
    uint4 gv = uint4{0, 0, 0, 0};
    if (MODE == 1 && valid) gv = gsrc[n * LPN + p];
    int dmax = deg;
#pragma unroll
    for (int off = LPN; off < 64; off <<= 1) dmax = max(dmax, __shfl_xor(dmax, off, 64));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int2 nxt = (p < deg) ? t_edge[q0 + p] : int2{0, 0};
    for (int e0 = 0; e0 < dmax; e0 += LPN) {
      const int2 my = nxt;
      nxt = (e0 + LPN + p < deg) ? t_edge[q0 + e0 + LPN + p] : int2{0, 0};
#pragma unroll
      for (int i = 0; i < LPN; ++i) {
        const int m = __shfl(my.x, i, LPN);
        const float v = __int_as_float(__shfl(my.y, i, LPN));
        const float4 s = sc[m];
        const uint4 zr = zl[m * LPN + p];
        float e = s1n + s.y;
        e = e > 0.f ? e : slope * e;
        const float c = (v != 0.f) ? v * eexp(e - s.z) * s.w : 0.f;
        float f8[8];
        unpack8(zr, f8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += c * f8[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);                 // the attention layer's ReLU (graphML.py:2101)
    if (MODE == 0) {
      if (valid) oseq[n * LPN + p] = pack8(acc);
    } else {
      if (valid && rdst) rdst[n * LPN + p] = pack8(acc);
      float g8[8];
      unpack8(gv, g8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = etanh(giv * g8[j] + gfv * acc[j]);
      const uint4 hv = pack8(acc);
      if (valid) oseq[n * LPN + p] = hv;
      if (hu) {
        const uint32_t w4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
          tile[(p * 8 + j) * TSTR + nl] = (uint16_t)(j & 1 ? w4[j >> 1] >> 16 : w4[j >> 1] & 0xffffu);
        __syncthreads();
        constexpr int PPR = NPP / 8;                                         // 16-byte pieces per feature row of the tile
        const int f = tid / PPR, pc = tid % PPR;
        if (base + pc * 8 < N)
          *reinterpret_cast<uint4*>(hu + (int64_t)f * N + base + pc * 8) = *reinterpret_cast<const uint4*>(tile + f * TSTR + pc * 8);
        __syncthreads();
      }
    }
  }
  for (int idx = N * LPN + tid; idx < NPad * LPN; idx += ETHREADS) {       // padding rows of the sequence-major output: zeros
    oseq[idx] = uint4{0, 0, 0, 0};
    if (rdst) rdst[idx] = uint4{0, 0, 0, 0};
  }
}

template <int F>
static int edge_att_fwd_t(const void* z, const float* a12, const void* gx, const float* gi, const float* gf, const int32_t* rowptr,
                          const void* r_edge, const int32_t* t_rowptr, const void* t_edge, void* out_seq, void* r_out, void* Huser,
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
                                               (const int2*)t_edge, (uint16_t*)out_seq, (uint16_t*)r_out, (uint16_t*)Huser, hu_stride,
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
                                               void* out_seq, void* r_out, void* Huser, int64_t huser_item_stride, int64_t items,
                                               int64_t N, int64_t NPad, int64_t F, double negative_slope, void* stream) {
  if (!z || !a12 || !rowptr || !r_edge || !t_rowptr || !t_edge || !out_seq) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (!gx && (gi || r_out || Huser)) return GCRNN_ERR_BAD_SHAPE;
  if (items <= 0 || items > 2147483647LL || N <= 0 || N > NPad || N % 8) return GCRNN_ERR_BAD_SHAPE;
  if (Huser && (reinterpret_cast<uintptr_t>(Huser) & 15 || huser_item_stride % 8)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
  if (F == 64) return edge_att_fwd_t<64>(z, a12, gx, gi, gf, rowptr, r_edge, t_rowptr, t_edge, out_seq, r_out, Huser, huser_item_stride, items, N, NPad, (float)negative_slope, st);
  if (F == 32) return edge_att_fwd_t<32>(z, a12, gx, gi, gf, rowptr, r_edge, t_rowptr, t_edge, out_seq, r_out, Huser, huser_item_stride, items, N, NPad, (float)negative_slope, st);
  return GCRNN_ERR_UNSUPPORTED;
}
