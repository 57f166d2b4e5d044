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

// s[item][k][n] = sum_f d[item][n][f] w[k][f];  thread = (item = blockIdx.y, node)
template <int F>
__global__ __launch_bounds__(256) void node_gate_dot_kernel(const uint16_t* __restrict__ d, const float* __restrict__ w, float* __restrict__ s,
                                                            int N, int NPad, int K) {
  __shared__ float ws[8 * F];
  for (int i = threadIdx.x; i < K * F; i += 256) ws[i] = w[i];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int64_t item = blockIdx.y;
  const uint4* row = reinterpret_cast<const uint4*>(d + (item * NPad + n) * F);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
  for (int j = 0; j < F / 8; ++j) {
    const uint4 v = row[j];
    const uint32_t p[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = __uint_as_float(p[e] << 16), hi = __uint_as_float(p[e] & 0xffff0000u);
      for (int k = 0; k < K; ++k) acc[k] += lo * ws[k * F + 8 * j + 2 * e] + hi * ws[k * F + 8 * j + 2 * e + 1];
    }
  }
  for (int k = 0; k < K; ++k) s[(item * K + k) * N + n] = acc[k];
}

// One workgroup per item: dpre_g[n][f] = (sum_k ds[k][n] w[k][f]) (1 - d[n][f]^2) overwrites d (bf16);
// dw_part[item][k][f] = sum_n ds[k][n] d[n][f] (the caller adds the items in a fixed order).
template <int F>
__global__ __launch_bounds__(512) void node_gate_dot_bwd_kernel(uint16_t* __restrict__ d, const float* __restrict__ ds, const float* __restrict__ w,
                                                                float* __restrict__ dw_part, int N, int NPad, int K) {
  constexpr int RT = 128;                       // rows (nodes) per tile
  __shared__ float ws[8 * F];
  __shared__ float dss[8][RT];
  __shared__ uint16_t ct[RT][F + 2];            // +2: odd word stride against bank conflicts of the column reads
  const int tid = threadIdx.x;
  const int64_t item = blockIdx.x;
  for (int i = tid; i < K * F; i += 512) ws[i] = w[i];
  const int kf = tid;                           // phase 2 role: (k, f) = (tid / F, tid % F) for tid < K * F
  float dacc = 0.f;
  for (int n0 = 0; n0 < N; n0 += RT) {
    __syncthreads();
    // stage the tile: ds[k][n0 .. n0+RT) and the original d rows
    for (int i = tid; i < K * RT; i += 512) {
      const int k = i / RT, rr = i - k * RT;
      dss[k][rr] = (n0 + rr < N) ? ds[(item * K + k) * N + n0 + rr] : 0.f;
    }
    for (int i = tid; i < RT * (F / 2); i += 512) {
      const int rr = i / (F / 2), c2 = i - rr * (F / 2);
      uint32_t v = 0;
      if (n0 + rr < N) v = *reinterpret_cast<const uint32_t*>(d + (item * NPad + n0 + rr) * F + 2 * c2);
      ct[rr][2 * c2] = (uint16_t)(v & 0xffffu);
      ct[rr][2 * c2 + 1] = (uint16_t)(v >> 16);
    }
    __syncthreads();
    // phase 1: the gate cell's pre-activation gradient, in place (thread = (row, feature pair))
    for (int i = tid; i < RT * (F / 2); i += 512) {
      const int rr = i / (F / 2), c2 = i - rr * (F / 2);
      if (n0 + rr < N) {
        float g0 = 0.f, g1 = 0.f;
        for (int k = 0; k < K; ++k) { g0 += dss[k][rr] * ws[k * F + 2 * c2]; g1 += dss[k][rr] * ws[k * F + 2 * c2 + 1]; }
        const float c0 = bf2f_(ct[rr][2 * c2]), c1 = bf2f_(ct[rr][2 * c2 + 1]);
        *reinterpret_cast<uint32_t*>(d + (item * NPad + n0 + rr) * F + 2 * c2) =
            (uint32_t)f2bf_(g0 * (1.f - c0 * c0)) | ((uint32_t)f2bf_(g1 * (1.f - c1 * c1)) << 16);
      }
    }
    // phase 2: dw[k][f] += sum_rows ds[k][row] d[row][f] (rows past N were staged as zeros)
    if (kf < K * F) {
      const int k = kf / F, f = kf - k * F;
#pragma unroll 8
      for (int rr = 0; rr < RT; ++rr) dacc += dss[k][rr] * bf2f_(ct[rr][f]);
    }
  }
  if (kf < K * F) dw_part[item * (K * F) + kf] = dacc;
}

}  // namespace

// s[item][k][n] (fp32) = sum_f d[item][n][f] w[k][f]:  d [items][NPad][F] bf16 sequence-major gate-cell states, w [K][F] fp32 (the
// reference's GraphFilter weight 1 x 1 x K x F, graphML.py:2303), K <= 8, F in {32, 64}.
extern "C" int gcrnn_node_gate_dot(const void* d, const float* w, float* s, int64_t items, int64_t N, int64_t NPad, int64_t F, int64_t K,
                                   void* stream) {
  if (!d || !w || !s) return GCRNN_ERR_NULL_POINTER;
  if (items <= 0 || items > 65535 * 64 || N <= 0 || NPad < N || K <= 0 || K > 8 || (F != 32 && F != 64)) return GCRNN_ERR_BAD_SHAPE;
  if (items > 65535) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  const dim3 grid((unsigned)cdiv(N, 256), (unsigned)items);
  if (F == 64) node_gate_dot_kernel<64><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)d, w, s, (int)N, (int)NPad, (int)K);
  else node_gate_dot_kernel<32><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)d, w, s, (int)N, (int)NPad, (int)K);
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
  if (F == 64) node_gate_dot_bwd_kernel<64><<<(unsigned)items, 512, 0, as_stream(stream)>>>((uint16_t*)d, ds, w, dw_part, (int)N, (int)NPad, (int)K);
  else node_gate_dot_bwd_kernel<32><<<(unsigned)items, 512, 0, as_stream(stream)>>>((uint16_t*)d, ds, w, dw_part, (int)N, (int)NPad, (int)K);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}
