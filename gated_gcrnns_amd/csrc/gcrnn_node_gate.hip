// Node gates of the fused path (reference Utils/graphML.py:2379-2399): ni_t = sigmoid(GraphFilter_{F -> 1, K taps}(d_t)), d_t the
// state of an un-gated gate cell on (x_t, h0). The F -> 1 filter is evaluated taps-first, like every filter of the fused path:
//     s_k[item][n] = sum_f d[item][n][f] w_k[f]     (this file: one pass over d, K scalars per node)
//     logit = sum_k P^k s_k + b                      (Horner on ONE-channel signals: K-1 accumulate-SpMMs over [N][items],
//                                                     gcrnn_spmm_ex on the node-major transpose -- a 16th of a state hop each)
// so the gate costs one read of d instead of K-1 hops over F channels. The backward kernel turns the gradient of the s_k into the
// gate cell's pre-activation gradient (in place over d, as the time gates' read-out backward does) and per-item partial sums of dw.
#include "gcrnn_common.h"

namespace {

__device__ __forceinline__ float bf2f_(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f2bf_(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

template <int CTRL> __device__ __forceinline__ float dpp_f_(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int LPN> __device__ __forceinline__ float group_sum_(float v) {      // sum over the LPN (4 or 8) aligned lanes of a node
  v += dpp_f_<0xB1>(v);                 // quad_perm [1,0,3,2]
  v += dpp_f_<0x4E>(v);                 // quad_perm [2,3,0,1]
  if (LPN == 8) v += dpp_f_<0x141>(v);  // row_half_mirror: the other quad's total
  return v;
}

// s[item][k][n] = sum_f d[item][n][f] w[k][f].  F/8 lanes per node, each owning one 16-byte piece of the row: a wave reads whole
// 128-byte lines (the first version's thread-per-node walk touched 64 lines per load and staged the weights through LDS reads:
// 0.95 ms per pass over 1 GB), the K partial dots are summed over the node's lanes by DPP, the weights live in registers.
template <int F, int KMAX>
__global__ __launch_bounds__(256) void node_gate_dot_kernel(const uint16_t* __restrict__ d, const float* __restrict__ w, float* __restrict__ s,
                                                            int N, int NPad, int K) {
  constexpr int LPN = F / 8, NPP = 256 / LPN;
  const int tid = threadIdx.x, p = tid % LPN, nl = tid / LPN;
  const int64_t item = blockIdx.x;
  float wr[KMAX][8];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[k][j] = k < K ? w[k * F + p * 8 + j] : 0.f;
  const uint4* rows = reinterpret_cast<const uint4*>(d + item * NPad * F);
  for (int base = 0; base < N; base += 2 * NPP) {                      // two independent rows per lane and trip
    const int n0 = base + nl, n1 = base + NPP + nl;
    const uint4 v0 = n0 < N ? rows[n0 * LPN + p] : uint4{0, 0, 0, 0};
    const uint4 v1 = n1 < N ? rows[n1 * LPN + p] : uint4{0, 0, 0, 0};
    const uint32_t q0[4] = {v0.x, v0.y, v0.z, v0.w}, q1[4] = {v1.x, v1.y, v1.z, v1.w};
    float f0[8], f1[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f0[2 * e] = __uint_as_float(q0[e] << 16); f0[2 * e + 1] = __uint_as_float(q0[e] & 0xffff0000u);
      f1[2 * e] = __uint_as_float(q1[e] << 16); f1[2 * e + 1] = __uint_as_float(q1[e] & 0xffff0000u);
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {                                                        // (wave-uniform)
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a0 += wr[k][j] * f0[j]; a1 += wr[k][j] * f1[j]; }
        a0 = group_sum_<LPN>(a0);                                         // every lane of the node's group now holds the full dot
        a1 = group_sum_<LPN>(a1);
        if ((k % LPN) == p) {                                             // lane k mod LPN of the group stores tap k
          if (n0 < N) s[(item * K + k) * N + n0] = a0;
          if (n1 < N) s[(item * K + k) * N + n1] = a1;
        }
      }
    }
  }
}

// One workgroup per item: dpre_g[n][f] = (sum_k ds[k][n] w[k][f]) (1 - d[n][f]^2) overwrites d (bf16);
// dw_part[item][k][f] = sum_n ds[k][n] d[n][f] (the caller adds the items in a fixed order).
// F/8 lanes per node: a lane keeps its 8 features of the row, the K x 8 weights and its K x 8 partial sums of dw in registers (the
// first version staged 128-row tiles in LDS and read every operand of every multiply-add from there: 0.8 ms per pass over 1 GB);
// the partial sums are folded over the wave's nodes by xor shuffles, over the waves through 5 KB of LDS, in a fixed order.
template <int F, int KMAX>
__global__ __launch_bounds__(256) void node_gate_dot_bwd_kernel(uint16_t* __restrict__ d, const float* __restrict__ ds, const float* __restrict__ w,
                                                                float* __restrict__ dw_part, int N, int NPad, int K) {
  constexpr int LPN = F / 8, NPP = 256 / LPN;
  __shared__ float red[4][LPN][KMAX * 8];
  const int tid = threadIdx.x, p = tid % LPN, nl = tid / LPN, wave = tid >> 6, lane = tid & 63;
  const int64_t item = blockIdx.x;
  float wr[KMAX][8], dacc[KMAX][8];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) { wr[k][j] = k < K ? w[k * F + p * 8 + j] : 0.f; dacc[k][j] = 0.f; }
  uint4* rows = reinterpret_cast<uint4*>(d + item * NPad * F);
  const float* dsi = ds + item * K * N;
  constexpr int U = 4;                                // rows per lane and trip: all their loads are issued before the first store (the
  for (int n0 = nl; n0 < N; n0 += U * NPP) {          // in-place update would otherwise serialise one memory latency per row)
    uint4 v[U];
    float dv[U][KMAX];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = n0 + u * NPP;
      v[u] = n < N ? rows[n * LPN + p] : uint4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < KMAX; ++k) dv[u][k] = (k < K && n < N) ? dsi[k * N + n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = n0 + u * NPP;
      const uint32_t q4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      float c[8], g[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { c[2 * e] = __uint_as_float(q4[e] << 16); c[2 * e + 1] = __uint_as_float(q4[e] & 0xffff0000u); }
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { g[j] += dv[u][k] * wr[k][j]; dacc[k][j] += dv[u][k] * c[j]; }
        }
      }
      uint32_t o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o4[e] = (uint32_t)f2bf_(g[2 * e] * (1.f - c[2 * e] * c[2 * e])) | ((uint32_t)f2bf_(g[2 * e + 1] * (1.f - c[2 * e + 1] * c[2 * e + 1])) << 16);
      if (n < N) rows[n * LPN + p] = uint4{o4[0], o4[1], o4[2], o4[3]};
    }
  }
  // fold over the nodes of a wave (lanes with equal p: xor 8, 16, 32 for LPN = 8; xor 4 .. 32 for LPN = 4), then over the waves
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = dacc[k][j];
#pragma unroll
        for (int off = LPN; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if (lane < LPN) red[wave][p][k * 8 + j] = v;
      }
    }
  }
  __syncthreads();
  for (int o = tid; o < K * F; o += 256) {
    const int k = o / F, f = o - k * F, pp = f / 8, j = f % 8;
    dw_part[item * (K * F) + o] = red[0][pp][k * 8 + j] + red[1][pp][k * 8 + j] + red[2][pp][k * 8 + j] + red[3][pp][k * 8 + j];
  }
}

// Gate gradients of the node-gated cell for all items at once (one workgroup per item, F/8 lanes per node, coalesced 16-byte pieces):
//   a[n] = sum_f dpre[n][f] Yx[n][f],  c[n] = sum_f dpre[n][f] Yh[n][f]
//   d ni[n] = gi a[n],  d nf[n] = gf c[n],  d gi += ni[n] a[n],  d gf += nf[n] c[n],  dYx[n][:] = gi ni[n] dpre[n][:]
template <int F>
__global__ __launch_bounds__(256) void node_cell_bwd_kernel(const uint16_t* __restrict__ dpre, const uint16_t* __restrict__ yx,
                                                            const uint16_t* __restrict__ yh, const float* __restrict__ ngates,
                                                            const float* __restrict__ gi, const float* __restrict__ gf,
                                                            uint16_t* __restrict__ dyx, float* __restrict__ dni, float* __restrict__ dnf,
                                                            float* __restrict__ dgi, float* __restrict__ dgf, int B, int N, int NPad) {
  constexpr int LPN = F / 8, NPP = 256 / LPN;
  __shared__ float red[2][4];
  const int tid = threadIdx.x, p = tid % LPN, nl = tid / LPN;
  const int64_t item = blockIdx.x;
  const int t = (int)(item / B), b = (int)(item - (int64_t)t * B);
  const float gin = gi ? gi[item] : 1.f, gfo = gf ? gf[item] : 1.f;
  const float* ni = ngates + ((int64_t)(t * 2 + 0) * B + b) * N;
  const float* nf = ngates + ((int64_t)(t * 2 + 1) * B + b) * N;
  const uint4* dr = reinterpret_cast<const uint4*>(dpre + item * NPad * F);
  const uint4* xr = reinterpret_cast<const uint4*>(yx + item * NPad * F);
  const uint4* hr = reinterpret_cast<const uint4*>(yh + item * NPad * F);
  uint4* orow = reinterpret_cast<uint4*>(dyx + item * NPad * F);
  float pgi = 0.f, pgf = 0.f;
  for (int n = nl; n < NPad; n += NPP) {
    if (n >= N) { orow[n * LPN + p] = uint4{0u, 0u, 0u, 0u}; continue; }      // padding rows of dYx stay zero (whole lane groups)
    const uint4 dv = dr[n * LPN + p], xv = xr[n * LPN + p], hv = hr[n * LPN + p];
    const float nin = ni[n], nfn = nf[n];
    const float sc = gin * nin;
    const uint32_t dp[4] = {dv.x, dv.y, dv.z, dv.w}, xp[4] = {xv.x, xv.y, xv.z, xv.w}, hp[4] = {hv.x, hv.y, hv.z, hv.w};
    uint32_t op[4];
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = __uint_as_float(dp[e] << 16), d1 = __uint_as_float(dp[e] & 0xffff0000u);
      a += d0 * __uint_as_float(xp[e] << 16) + d1 * __uint_as_float(xp[e] & 0xffff0000u);
      c += d0 * __uint_as_float(hp[e] << 16) + d1 * __uint_as_float(hp[e] & 0xffff0000u);
      op[e] = (uint32_t)f2bf_(sc * d0) | ((uint32_t)f2bf_(sc * d1) << 16);
    }
    orow[n * LPN + p] = uint4{op[0], op[1], op[2], op[3]};
    a = group_sum_<LPN>(a);
    c = group_sum_<LPN>(c);
    if (p == 0) {
      dni[item * N + n] = gin * a;
      dnf[item * N + n] = gfo * c;
      pgi += nin * a;
      pgf += nfn * c;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { pgi += __shfl_down(pgi, o, 64); pgf += __shfl_down(pgf, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = pgi; red[1][threadIdx.x >> 6] = pgf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (dgi) dgi[item] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    if (dgf) dgf[item] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

}  // namespace

// Gate gradients and the x-part pre-activation gradient of the node-gated cell (graphML.py:2402-2405 under autograd), all T*B items:
// dpre / yx / yh / dyx [T*B][NPad][F] bf16 (yx = A(S)x_t + b, yh = B(S)h_{t-1} + b from the forward; dyx out = gi ni . dpre), ngates
// fp32 [T][2][B][N], gi / gf fp32 [T*B] or NULL (= 1); out: dni, dnf fp32 [T*B][N]; dgi, dgf fp32 [T*B] (may be NULL).
extern "C" int gcrnn_node_cell_backward(const void* dpre, const void* yx, const void* yh, const float* ngates, const float* gi,
                                        const float* gf, void* dyx, float* dni, float* dnf, float* dgi, float* dgf, int64_t B, int64_t T,
                                        int64_t N, int64_t NPad, int64_t F, void* stream) {
  if (!dpre || !yx || !yh || !ngates || !dyx || !dni || !dnf) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || B * T > 2147483647LL || N <= 0 || NPad < N || (F != 32 && F != 64)) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  if (F == 64)
    node_cell_bwd_kernel<64><<<(unsigned)(B * T), 256, 0, as_stream(stream)>>>((const uint16_t*)dpre, (const uint16_t*)yx, (const uint16_t*)yh, ngates,
                                                                             gi, gf, (uint16_t*)dyx, dni, dnf, dgi, dgf, (int)B, (int)N, (int)NPad);
  else
    node_cell_bwd_kernel<32><<<(unsigned)(B * T), 256, 0, as_stream(stream)>>>((const uint16_t*)dpre, (const uint16_t*)yx, (const uint16_t*)yh, ngates,
                                                                             gi, gf, (uint16_t*)dyx, dni, dnf, dgi, dgf, (int)B, (int)N, (int)NPad);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// s[item][k][n] (fp32) = sum_f d[item][n][f] w[k][f]:  d [items][NPad][F] bf16 sequence-major gate-cell states, w [K][F] fp32 (the
// reference's GraphFilter weight 1 x 1 x K x F, graphML.py:2303), K <= 8, F in {32, 64}.
extern "C" int gcrnn_node_gate_dot(const void* d, const float* w, float* s, int64_t items, int64_t N, int64_t NPad, int64_t F, int64_t K,
                                   void* stream) {
  if (!d || !w || !s) return GCRNN_ERR_NULL_POINTER;
  if (items <= 0 || items > 65535 * 64 || N <= 0 || NPad < N || K <= 0 || K > 8 || (F != 32 && F != 64)) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  if (F == 64) node_gate_dot_kernel<64, 8><<<(unsigned)items, 256, 0, as_stream(stream)>>>((const uint16_t*)d, w, s, (int)N, (int)NPad, (int)K);
  else node_gate_dot_kernel<32, 8><<<(unsigned)items, 256, 0, as_stream(stream)>>>((const uint16_t*)d, w, s, (int)N, (int)NPad, (int)K);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// Backward of gcrnn_node_gate_dot through the gate cell's tanh: ds [items][K][N] fp32 = d loss / d s; d is overwritten by the gate
// cell's pre-activation gradient (sum_k ds_k w_k) (1 - d^2) (bf16, rows >= N untouched = zero); dw_part [items][K][F] fp32 partial
// sums of d loss / d w (added by the caller in a fixed order).
extern "C" int gcrnn_node_gate_dot_backward(void* d, const float* ds, const float* w, float* dw_part, int64_t items, int64_t N, int64_t NPad,
                                            int64_t F, int64_t K, void* stream) {
  if (!d || !ds || !w || !dw_part) return GCRNN_ERR_NULL_POINTER;
  if (items <= 0 || items > 2147483647LL || N <= 0 || NPad < N || K <= 0 || K > 8 || (F != 32 && F != 64) || K * F > 512) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  if (K * F > 256) {      // the final fold uses one thread per (k, f): K <= 4 for F = 64 goes through the 8-tap instantiation's 256 threads twice
    if (F == 64) node_gate_dot_bwd_kernel<64, 8><<<(unsigned)items, 256, 0, as_stream(stream)>>>((uint16_t*)d, ds, w, dw_part, (int)N, (int)NPad, (int)K);
    else node_gate_dot_bwd_kernel<32, 8><<<(unsigned)items, 256, 0, as_stream(stream)>>>((uint16_t*)d, ds, w, dw_part, (int)N, (int)NPad, (int)K);
  } else {
    if (F == 64) node_gate_dot_bwd_kernel<64, 4><<<(unsigned)items, 256, 0, as_stream(stream)>>>((uint16_t*)d, ds, w, dw_part, (int)N, (int)NPad, (int)K);
    else node_gate_dot_bwd_kernel<32, 8><<<(unsigned)items, 256, 0, as_stream(stream)>>>((uint16_t*)d, ds, w, dw_part, (int)N, (int)NPad, (int)K);
  }
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}
