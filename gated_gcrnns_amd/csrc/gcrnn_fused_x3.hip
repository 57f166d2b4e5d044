// fp32-accurate fused GCRNN step on the bf16 matrix cores ("x3"): the 1e-5 parity mode of the north_star at fused-kernel speed.
//
// Same decomposition as fused_step_kernel (gcrnn_fused_step.h): one workgroup = (sequence, 16-feature chunk), taps on the matrix
// cores, K-1 Horner hops on an fp32 LDS state image, tanh epilogue -- the reference's h_t = tanh(A(S)x_t + b + B(S)h_{t-1} + b)
// (Utils/graphML.py:2420-2423, un-gated cell). What changes is the operand format: every fp32 operand v is carried as THREE bf16
// planes  v = v1 + v2 + v3,  v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2)  (3 x 8 significant bits = fp32's 24), for the
// state, the input AND the taps. A product z.w is evaluated as the six partial products whose weight is at least 2^-16 of the
// leading one -- z1w1, z1w2, z1w3, z2w1, z2w2, z3w1 -- on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: bf16 x bf16 products
// are exact in fp32, so the result is an fp32 dot product to within a few ulp, at 6/16 of the cost of the fp32 MFMA
// (v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate). Hops, bias, tanh are fp32 as before; the state goes back to HBM as
// three planes (6 bytes per element) and, for the caller, as fp32 in the user layout H[b][t][f][:].
// Uniform-weight graphs only (the weight planes take 3 x the LDS of the bf16 kernel: 64 KiB state + 60 KiB taps + 32 B x entries).
#include "gcrnn_fused_step.h"

namespace {

__device__ __forceinline__ void split3(float v, uint16_t& a, uint16_t& b, uint16_t& c) {
  a = f2bf(v);
  const float r1 = v - bf2f(a);           // exact
  b = f2bf(r1);
  const float r2 = r1 - bf2f(b);          // exact
  c = f2bf(r2);
}

// user fp32 [B][T][C][N]  ->  three bf16 planes, sequence-major [T][3][B][NPad][C], rows N..NPad-1 zero
// Optional pre-operation, fused so that the scaled / derived tensor never exists in fp32: v' = item_scale[t][b] * rowmul[c][n] * (OMS ? 1 - v^2 : v)
// (each factor optional) -- the time-gated cell's scaled operands gi x_t, gf h_{t-1} and its gate cells' upstream gradient
// d logit[t][b] * w[c][n] * (1 - c^2) (graphML.py:2362-2374 under autograd).
__global__ __launch_bounds__(256) void seq_pack_x3_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int B, int Tn, int C,
                                                          int N, int NPad, const float* __restrict__ item_scale = nullptr,
                                                          const float* __restrict__ rowmul = nullptr, int oms = 0, int64_t sstride = 0) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int bt = blockIdx.z, b = bt / Tn, t = bt - b * Tn;
  const int64_t ubase = sstride ? (int64_t)b * sstride + (int64_t)t * C * N : ((int64_t)(b * Tn + t) * C) * N;      // (sstride: elements between the sequences of src)
  const int n = n0 + tx;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    float v = (c < C && n < N) ? src[ubase + (int64_t)c * N + n] : 0.f;
    if (oms) v = 1.f - v * v;
    if (rowmul && c < C && n < N) v = rowmul[(int64_t)c * N + n] * v;
    if (item_scale) v = item_scale[t * B + b] * v;
    tile[ty + 8 * i][tx] = (c < C && n < N) ? v : 0.f;
  }
  __syncthreads();
  const int cp = threadIdx.x & 15, nr = threadIdx.x >> 4;             // 16 column pairs x 16 rows per pass
  const int c = c0 + 2 * cp;
  const int64_t plane = (int64_t)B * NPad * C;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int nl = nr + 16 * i, nn = n0 + nl;
    if (nn < NPad && c < C) {                                          // C is even (host check): whole pairs
      uint16_t a0, a1, a2, b0, b1, b2;
      split3(tile[2 * cp][nl], a0, a1, a2);
      split3(tile[2 * cp + 1][nl], b0, b1, b2);
      uint16_t* d = dst + ((int64_t)t * 3 * B + b) * NPad * C + (int64_t)nn * C + c;
      *reinterpret_cast<uint32_t*>(d) = (uint32_t)a0 | ((uint32_t)b0 << 16);
      *reinterpret_cast<uint32_t*>(d + plane) = (uint32_t)a1 | ((uint32_t)b1 << 16);
      *reinterpret_cast<uint32_t*>(d + 2 * plane) = (uint32_t)a2 | ((uint32_t)b2 << 16);
    }
  }
}

// The same pass on 64 channel x 64 node tiles with 16-byte accesses on both sides (round 5): float4 loads along n, and per node and plane 32 bytes
// (16 channels) per lane -- four lanes write a node's whole 128-byte row. The 32 x 32 tiles above moved 3.5 GB in 1.3 ms (2.7 TB/s: 4-byte
// accesses, half a million workgroups of one tile each). Needs N % 4 == 0, C % 16 == 0 and 16-byte aligned rows; otherwise the kernel above runs.
__global__ __launch_bounds__(256) void seq_pack_x3_wide_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int B, int Tn, int C,
                                                               int N, int NPad, const float* __restrict__ item_scale,
                                                               const float* __restrict__ rowmul, int oms, int64_t sstride) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int bt = blockIdx.z, b = bt / Tn, t = bt - b * Tn;
  const int64_t ubase = sstride ? (int64_t)b * sstride + (int64_t)t * C * N : ((int64_t)(b * Tn + t) * C) * N;
  const float sc = item_scale ? item_scale[t * B + b] : 1.f;
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, c = c0 + (idx >> 4), n = n0 + 4 * (idx & 15);
    v[i] = (c < C && n < N) ? *reinterpret_cast<const float4*>(src + ubase + (int64_t)c * N + n) : float4{0.f, 0.f, 0.f, 0.f};      // (N % 4 == 0: whole quads)
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, cl = idx >> 4, c = c0 + cl, nq = 4 * (idx & 15), n = n0 + nq;
    float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
    if (oms) {
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = 1.f - e[j] * e[j];
    }
    if (rowmul && c < C && n < N) {
      const float4 r = *reinterpret_cast<const float4*>(rowmul + (int64_t)c * N + n);
      e[0] *= r.x; e[1] *= r.y; e[2] *= r.z; e[3] *= r.w;
    }
    const bool in = c < C && n < N;
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[cl][nq + j] = in ? (item_scale ? sc * e[j] : e[j]) : 0.f;
  }
  __syncthreads();
  const int cg = tid & 3, nl = tid >> 2, nn = n0 + nl, c = c0 + 16 * cg;
  if (nn < NPad && c < C) {                                            // C % 16 == 0 (host check): whole groups of 16 channels
    uint32_t w[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint16_t a0, a1, a2, b0, b1, b2;
      split3(tile[16 * cg + 2 * j][nl], a0, a1, a2);
      split3(tile[16 * cg + 2 * j + 1][nl], b0, b1, b2);
      w[0][j] = (uint32_t)a0 | ((uint32_t)b0 << 16);
      w[1][j] = (uint32_t)a1 | ((uint32_t)b1 << 16);
      w[2][j] = (uint32_t)a2 | ((uint32_t)b2 << 16);
    }
    const int64_t plane = (int64_t)B * NPad * C;
    uint16_t* d = dst + ((int64_t)t * 3 * B + b) * NPad * C + (int64_t)nn * C + c;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      uint4* dp = reinterpret_cast<uint4*>(d + pl * plane);
      dp[0] = uint4{w[pl][0], w[pl][1], w[pl][2], w[pl][3]};
      dp[1] = uint4{w[pl][4], w[pl][5], w[pl][6], w[pl][7]};
    }
  }
}

static int seq_pack_x3_launch(const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N, int64_t NPad, const float* item_scale,
                              const float* rowmul, int oms, int64_t sstride, hipStream_t st) {
  const bool wide = N % 4 == 0 && C % 16 == 0 && !(reinterpret_cast<uintptr_t>(src) & 15) && !(reinterpret_cast<uintptr_t>(dst) & 15) &&
                    !(reinterpret_cast<uintptr_t>(rowmul) & 15) && sstride % 4 == 0 && cdiv(C, 64) <= 65535;
  GCRNN_PRE_LAUNCH();
  if (wide)
    seq_pack_x3_wide_kernel<<<dim3((unsigned)cdiv(NPad, 64), (unsigned)cdiv(C, 64), (unsigned)(B * T)), 256, 0, st>>>(
        (const float*)src, (uint16_t*)dst, (int)B, (int)T, (int)C, (int)N, (int)NPad, item_scale, rowmul, oms, sstride);
  else
    seq_pack_x3_kernel<<<dim3((unsigned)cdiv(NPad, 32), (unsigned)cdiv(C, 32), (unsigned)(B * T)), 256, 0, st>>>(
        (const float*)src, (uint16_t*)dst, (int)B, (int)T, (int)C, (int)N, (int)NPad, item_scale, rowmul, oms, sstride);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// One step of the NODE-gated cell at fp32 accuracy behind its two x3 filter passes (Utils/graphML.py:2402-2407, 2420-2423), in one pass:
//   h[b][n][f] = tanh( ni[b][n] (ya[b][n][f] + bias[f]) + nf[b][n] (yb[b][n][f] + bias[f]) ),   ya / yb re-assembled from their three planes,
// stored as three planes again (the next step's operand: no separate plane cut of h) and, transposed through an LDS tile, as fp32 in the user
// layout H[b][.][f][n]. Replaces nine torch kernels + a plane cut per step (0.35 ms of a step's 0.9 at the bench's size).
// One workgroup = 64 nodes x F features of one sequence; F in {32, 64}; rows n >= N are written as zeros.
template <int F>
__global__ __launch_bounds__(256) void x3_node_step_kernel(const uint16_t* __restrict__ ya3, const uint16_t* __restrict__ yb3,
                                                           const float* __restrict__ ni, const float* __restrict__ nf,
                                                           const float* __restrict__ bias, uint16_t* __restrict__ h3,
                                                           float* __restrict__ Huser, int64_t hu_stride, int B, int N, int NPad) {
  constexpr int LPN = F / 8, NPP = 256 / LPN;      // lanes per node (8 features each), nodes per pass
  __shared__ float tile[F][65];
  const int tid = threadIdx.x, p = tid % LPN, nl = tid / LPN;
  const int b = blockIdx.y, n0 = blockIdx.x * 64;
  const int64_t plane = (int64_t)B * NPad * F;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[p * 8 + j] : 0.f;
#pragma unroll
  for (int ps = 0; ps < 64 / NPP; ++ps) {
    const int nn = ps * NPP + nl, n = n0 + nn;
    if (n < NPad) {
      const int64_t o = ((int64_t)b * NPad + n) * F + p * 8;
      float a[8], c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = 0.f; c[j] = 0.f; }
#pragma unroll
      for (int pl = 2; pl >= 0; --pl) {              // small to large: exact for planes cut from one fp32 value
        const uint4 va = *reinterpret_cast<const uint4*>(ya3 + pl * plane + o), vc = *reinterpret_cast<const uint4*>(yb3 + pl * plane + o);
        const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wc[4] = {vc.x, vc.y, vc.z, vc.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[2 * j] += __uint_as_float(wa[j] << 16); a[2 * j + 1] += __uint_as_float(wa[j] & 0xffff0000u);
          c[2 * j] += __uint_as_float(wc[j] << 16); c[2 * j + 1] += __uint_as_float(wc[j] & 0xffff0000u);
        }
      }
      const bool in = n < N;
      const float gi = in ? ni[(int64_t)b * NPad + n] : 0.f, gf = in ? nf[(int64_t)b * NPad + n] : 0.f;
      uint32_t w[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float h2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float pre = gi * (a[2 * j + e] + bv[2 * j + e]) + gf * (c[2 * j + e] + bv[2 * j + e]);
          h2[e] = in ? tanhf(pre) : 0.f;
          tile[p * 8 + 2 * j + e][nn] = h2[e];
        }
        uint16_t a0, a1, a2, b0, b1, b2;
        split3(h2[0], a0, a1, a2);
        split3(h2[1], b0, b1, b2);
        w[0][j] = (uint32_t)a0 | ((uint32_t)b0 << 16);
        w[1][j] = (uint32_t)a1 | ((uint32_t)b1 << 16);
        w[2][j] = (uint32_t)a2 | ((uint32_t)b2 << 16);
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(h3 + pl * plane + o) = uint4{w[pl][0], w[pl][1], w[pl][2], w[pl][3]};
    }
  }
  if (!Huser) return;
  __syncthreads();
  // user layout: row f of the tile along n, 16 bytes (4 nodes) per lane (N % 4 == 0: whole quads)
  for (int idx = tid; idx < F * 16; idx += 256) {
    const int f = idx >> 4, nq = 4 * (idx & 15), n = n0 + nq;
    if (n < N) *reinterpret_cast<float4*>(Huser + (int64_t)b * hu_stride + (int64_t)f * N + n) = float4{tile[f][nq], tile[f][nq + 1], tile[f][nq + 2], tile[f][nq + 3]};
  }
}

// taps -> three planes of per-lane MFMA A fragments: out[p][chunk][tap][kstep][lane][8] (layout of pack_weights_kernel per plane)
__global__ void pack_weights_x3_kernel(const float* __restrict__ wA, const float* __restrict__ wB, uint16_t* __restrict__ out, int F, int G,
                                       int Kin, int Kst, int K) {
  const int KS = (F + G) / 32;
  const int64_t total = (int64_t)(F / FC) * K * KS * 64 * 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 7, lane = (idx >> 3) & 63;
  int64_t rest = idx >> 9;
  const int s = rest % KS; rest /= KS;
  const int tap = rest % K;
  const int chunk = rest / K;
  const int f = chunk * FC + (lane & 15);
  const int feat = 32 * s + 8 * (lane >> 4) + j;
  float v = 0.f;
  if (feat < F) { if (tap < Kst) v = wB[((int64_t)f * Kst + tap) * F + feat]; }
  else          { if (tap < Kin) v = wA[((int64_t)f * Kin + tap) * G + (feat - F)]; }
  uint16_t a, b, c;
  split3(v, a, b, c);
  out[idx] = a; out[idx + total] = b; out[idx + 2 * total] = c;
}

// per item (t, b): <a, b> over the item's NPad x C elements (fp32 re-assembled from the planes) and <a, 1 vec^T> (vec [C] fp32: the bias part of
// a gate's gradient); one workgroup per item, fixed summation order
__global__ __launch_bounds__(256) void x3_item_dots_kernel(const uint16_t* __restrict__ a3, const uint16_t* __restrict__ b3, const float* __restrict__ vec,
                                                           float* __restrict__ out_ab, float* __restrict__ out_av, int B, int NPad, int C) {
  const int it = blockIdx.x, t = it / B, b = it - t * B;
  const int64_t plane = (int64_t)B * NPad * C, base = ((int64_t)t * 3 * B + b) * NPad * C;
  const int per = NPad * C / 2;                       // pairs (C even)
  float sab = 0.f, sav = 0.f;
  for (int i = threadIdx.x; i < per; i += 256) {
    const int c = (2 * i) % C;
    float av[2] = {0.f, 0.f}, bv[2] = {0.f, 0.f};
#pragma unroll
    for (int p = 2; p >= 0; --p) {
      const uint32_t wa = *reinterpret_cast<const uint32_t*>(a3 + base + p * plane + 2 * (int64_t)i);
      av[0] += bf2f((uint16_t)(wa & 0xffffu)); av[1] += bf2f((uint16_t)(wa >> 16));
      if (b3) {
        const uint32_t wb = *reinterpret_cast<const uint32_t*>(b3 + base + p * plane + 2 * (int64_t)i);
        bv[0] += bf2f((uint16_t)(wb & 0xffffu)); bv[1] += bf2f((uint16_t)(wb >> 16));
      }
    }
    sab = __builtin_fmaf(av[1], bv[1], __builtin_fmaf(av[0], bv[0], sab));
    if (vec) sav = __builtin_fmaf(av[1], vec[c + 1], __builtin_fmaf(av[0], vec[c], sav));
  }
  __shared__ float red[2][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sab += __shfl_down(sab, off, 64); sav += __shfl_down(sav, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sab; red[1][threadIdx.x >> 6] = sav; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (out_ab) out_ab[it] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    if (out_av) out_av[it] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// EPI 0: forward step (bias, tanh).  EPI 2 (XS = 0): BPTT data-gradient step in the same arithmetic -- operand = the planes of
// dpre_t, taps = the transposed state taps, adjoint graph; epilogue dpre_{t-1} = (acc + dH_{t-1}) (1 - h_{t-1}^2) in fp32 with dH and h
// re-assembled from their planes (aux0 / aux1, both may be null: the raw state gradient d h0), stored as three planes again.
// SZ (EPI 0): the state operand is all zeros (gate cells from a zero h0, train_rnn.py:256): its fragments are neither loaded nor multiplied
// (exact: the skipped products are zeros) -- a kernel of its own, so that the general one keeps its registers.
// R1: rank-1-weighted graph (the r1 table) -- compile-time as well, so that uniform graphs keep their code and registers.
template <int K, int HS, int XS, int EPI = 0, bool SZ = false, bool R1 = false>
__global__ __launch_bounds__(STHREADS) void fused_step_x3_kernel(
    const uint16_t* __restrict__ xt3,        // [3][B][NP][G]  planes of x_t
    const uint16_t* __restrict__ hp3,        // [3][B][NP][F]  planes of h_{t-1}
    uint16_t* __restrict__ ho3,              // [3][B][NP][F]  planes of h_t
    const uint4* __restrict__ wpack3,        // [3][F/16][K][KS][64] x 16 B
    const float* __restrict__ bias,          // [F] or null
    const int32_t* __restrict__ tile_nodes, const int32_t* __restrict__ tile_off, const uint2* __restrict__ ell_col4,
    float* __restrict__ Huser,               // user-layout fp32 output block of this step, H[.][t][F][N] (or null)
    int64_t ubstride,                        // elements between consecutive sequences of Huser
    int entries, int B, int N, float uni_w,
    const uint16_t* __restrict__ aux0_3,               // EPI 2: planes of the upstream gradient dH_{t-1} [3][B][NP][F] (or null)
    const uint16_t* __restrict__ aux1_3,               // EPI 2: planes of the state h_{t-1} [3][B][NP][F] (or null)
    const float* __restrict__ bscale,                  // EPI 0 (or null): per-sequence weight of the bias [B] instead of 2 (time-gated cell: gi + gf);
                                                       // EPI 2 (or null): per-sequence forget gate [B] of the step back-propagated (scales the hops' output)
    float* __restrict__ gpart,                         // EPI 2 (or null; needs aux1): [B][F/16 * 8] partials of <h_{t-1}, adjoint chain of dpre_t> = d loss / d gf_t
    const float* __restrict__ r1) {                    // rank-1-weighted graph S[m][n] = a[m] b[n] on the plan of its 0/1 pattern (or null): [4][NP] fp32 = a | a b | 1 / b | b
                                                       // (source factor, its product with the destination factor, destination factor and inverse; 1 where b = 0).
                                                       // Horner in t' = t / b: t'_j = u_j / b + sum_{m in N(n)} (a b t'_{j+1})[m], t_0 = b t'_0 -- the stream still adds in place.
  static_assert(!SZ || (EPI == 0 && XS > 0), "zero state: forward cells with an input operand");
  static_assert(GCRNN_HOP_ASM && GCRNN_STEP_WAVES == 8, "x3 runs on the one-block asm hop stream");
  constexpr int KS = HS + XS, F = 32 * HS, G = 32 * XS, NCH = F / FC, HT = STILES;
  constexpr int WPL = K * KS * 64;            // uint4 fragments per weight plane and chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* state = reinterpret_cast<float*>(smem);
  uint4* wl = reinterpret_cast<uint4*>(smem + NP * FC * 4);                       // [3][K][KS][64]
  uint2* lcol4 = reinterpret_cast<uint2*>(smem + NP * FC * 4 + 3 * WPL * 16);

  const int L = blockIdx.x;
  const int grp = L / (8 * NCH), rem = L - grp * (8 * NCH);
  const int chunk = rem >> 3, b0 = grp * 8 + (rem & 7);
  const int seq_slots = (gridDim.x / (8 * NCH)) * 8;
  if (b0 >= B) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

#pragma unroll
  for (int p = 0; p < 3; ++p)
    for (int i = tid; i < WPL; i += STHREADS) wl[p * WPL + i] = wpack3[((int64_t)p * NCH + chunk) * WPL + i];
  {
    const int n = (entries >> 2) * 16;
    for (int i = tid; i < n; i += STHREADS) lcol4[i] = ell_col4[i];
  }
  int tbeg[STILES], tend[STILES], woff[STILES];
#pragma unroll
  for (int i = 0; i < STILES; ++i) {
    tbeg[i] = tile_off[wave * STILES + i];
    tend[i] = tile_off[wave * STILES + i + 1];
    woff[i] = tile_nodes[(wave * STILES + i) * 16 + r] ^ (q << 4);            // node << 16 | row << 6 | swz << 4, this lane's quad
  }
  float bvec[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) bvec[c] = bias[chunk * FC + q * 4 + c];        // the one bias enters through both filters (graphML.py:2420-2421): weight 2, or gi + gf
  }
  __syncthreads();
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  if (lds0 != 0) __builtin_trap();
  const uint32_t qx = (uint32_t)(q * 16);
  const uint32_t lds_col = lds0 + NP * FC * 4 + 3 * WPL * 16;
  const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(hp3), 0, 3 * B * (NP * F * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xt3), 0, 3 * B * (NP * G * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(ho3, 0, 3 * B * (NP * F * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(aux0_3), 0, (EPI == 2 && aux0_3) ? 3 * B * (NP * F * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(aux1_3), 0, (EPI == 2 && aux1_3) ? 3 * B * (NP * F * 2) : 0, 0x00020000);

  for (int b = b0; b < B; b += seq_slots) {
    // ---- phase 1: taps, six partial products per operand pair, two tiles per weight fragment ------------------------------
    // 12 (tile pair, plane) stages; the fragments of stage g+1 are requested before the MFMAs of stage g (two register sets),
    // so one L2 latency is exposed per sequence instead of one per stage.
    f32x4 u[STILES][K - 1];
    bf16x8 fr[2][2][KS];                     // [set][tile of the pair][k-step]
    auto load_stage = [&](int g, int set) {
      const int i = 2 * (g / 3), p = g % 3;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int w = woff[i + j];
        asm volatile("" : "+v"(w));
        const int roh = (w >> 16) * (F * 2) + 16 * q, rox = (w >> 16) * (G * 2) + 16 * q;
        if constexpr (!SZ) {
#pragma unroll
          for (int s = 0; s < HS; ++s)
            fr[set][j][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, roh + 64 * s, (p * B + b) * (NP * F * 2), 0));
        }
#pragma unroll
        for (int s = 0; s < XS; ++s)
          fr[set][j][HS + s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, rox + 64 * s, (p * B + b) * (NP * G * 2), 0));
      }
    };
    load_stage(0, 0);
    f32x4 acc[K][2];
#pragma unroll
    for (int g = 0; g < 3 * (STILES / 2); ++g) {
      const int i = 2 * (g / 3), p = g % 3, set = g & 1;
      if (p == 0) {
#pragma unroll
        for (int tap = 0; tap < K; ++tap) { acc[tap][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[tap][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
      if (g + 1 < 3 * (STILES / 2)) load_stage(g + 1, set ^ 1);
      __builtin_amdgcn_sched_barrier(0);       // keep the next stage's loads ahead of this stage's MFMAs
#pragma unroll
      for (int tap = 0; tap < K; ++tap)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (SZ && s < HS) continue;
#pragma unroll
          for (int wp = 0; wp < 3 - p; ++wp) {             // operand plane p meets weight planes 0 .. 2-p
            const bf16x8 a = __builtin_bit_cast(bf16x8, wl[wp * WPL + (tap * KS + s) * 64 + lane]);
            acc[tap][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fr[set][0][s], acc[tap][0], 0, 0, 0);
            acc[tap][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fr[set][1][s], acc[tap][1], 0, 0, 0);
          }
        }
      if (p == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int wv = woff[i + j];
          asm volatile("" : "+v"(wv));
          f32x4 top = acc[K - 1][j];
          if constexpr (R1) {
            const float binv = r1[2 * NP + (wv >> 16)], a1 = r1[wv >> 16];
#pragma unroll
            for (int tap = 0; tap < K - 1; ++tap) acc[tap][j] *= binv;
            top *= a1;
          }
#pragma unroll
          for (int tap = 0; tap < K - 1; ++tap) u[i + j][tap] = acc[tap][j];
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(state) + (wv & 0xffff)) = top;
        }
      }
    }
    lds_barrier();
    // ---- phase 2: Horner hops on the fp32 state image (uniform-weight asm stream) -------------------------------------------
#pragma unroll
    for (int j = 1; j < K; ++j) {
#define GCRNN_X3_INIT(i) u[i][K - 1 - j]
#define GCRNN_X3_STORE(i, a) u[i][K - 1 - j] = a
      GCRNN_HOP_ASM_UNI_STREAM(GCRNN_X3_INIT, GCRNN_X3_STORE);
#undef GCRNN_X3_INIT
#undef GCRNN_X3_STORE
      if (j < K - 1) {
        lds_barrier();
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          f32x4 im = u[i][K - 1 - j];
          if constexpr (R1) im *= r1[NP + (wv >> 16)];
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(state) + (wv & 0xffff)) = im;
        }
        lds_barrier();
      }
    }
    if constexpr (R1) {
#pragma unroll
      for (int i = 0; i < STILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        u[i][0] *= r1[3 * NP + (wv >> 16)];
      }
    }
    // ---- epilogue: bias, tanh, three planes of h_t; fp32 user-layout copy through an LDS transpose in two node halves ------
    constexpr int LASTU = (K > 1) ? 0 : 0;
    [[maybe_unused]] float part = 0.f;
#pragma unroll
    for (int i = 0; i < STILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      const int node = wv >> 16;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      const int eoff = node * (F * 2) + (chunk * FC + q * 4) * 2;
      if constexpr (EPI == 2) {
        // three planes -> fp32: v = v1 + v2 + v3 (exact: the planes were cut from one fp32 value); zero-length descriptors give 0
        // time-gated cell: the hops' output sum_k S^k (dpre_t B_k^T) is the forget gate's gradient when dotted with h_{t-1} (adjoint identity:
        // <B(S) h_{t-1}, dpre_t>), and enters d h_{t-1} scaled by gf_t (graphML.py:2420-2423 under autograd)
        auto gather3 = [&](const __amdgpu_buffer_rsrc_t& rs) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int p = 2; p >= 0; --p) {
            const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(rs, eoff, (p * B + b) * (NP * F * 2), 0);
            v[0] += bf2f((uint16_t)(w2[0] & 0xffffu)); v[1] += bf2f((uint16_t)(w2[0] >> 16));
            v[2] += bf2f((uint16_t)(w2[1] & 0xffffu)); v[3] += bf2f((uint16_t)(w2[1] >> 16));
          }
          return v;
        };
        if (node < N) {
          o = u[i][LASTU];
          f32x4 hv = {0.f, 0.f, 0.f, 0.f};
          if (aux1_3) hv = gather3(rsrc_a1);
          if (gpart) part = __builtin_fmaf(o[3], hv[3], __builtin_fmaf(o[2], hv[2], __builtin_fmaf(o[1], hv[1], __builtin_fmaf(o[0], hv[0], part))));
          if (bscale) {
            const float gsc = bscale[b];
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] *= gsc;
          }
          if (aux0_3) {
            const f32x4 g = gather3(rsrc_a0);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (o[c] + g[c]) * (1.f - hv[c] * hv[c]);
          }
        }
      } else if (node < N) {
        const f32x4 a = u[i][LASTU];
        const float bs = bscale ? bscale[b] : 2.f;          // (2 b is exact: the un-gated results keep their bits)
        o = f32x4{fast_tanh(a[0] + bs * bvec[0]), fast_tanh(a[1] + bs * bvec[1]), fast_tanh(a[2] + bs * bvec[2]), fast_tanh(a[3] + bs * bvec[3])};
      }
      uint16_t p0[4], p1[4], p2[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) split3(o[c], p0[c], p1[c], p2[c]);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)p0[0] | ((uint32_t)p0[1] << 16), (uint32_t)p0[2] | ((uint32_t)p0[3] << 16)}, rsrc_o, eoff, (0 * B + b) * (NP * F * 2), 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)p1[0] | ((uint32_t)p1[1] << 16), (uint32_t)p1[2] | ((uint32_t)p1[3] << 16)}, rsrc_o, eoff, (1 * B + b) * (NP * F * 2), 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)p2[0] | ((uint32_t)p2[1] << 16), (uint32_t)p2[2] | ((uint32_t)p2[3] << 16)}, rsrc_o, eoff, (2 * B + b) * (NP * F * 2), 0);
      u[i][LASTU] = o;
    }
    if constexpr (EPI == 2) {
      if (gpart) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if (lane == 0) gpart[((int64_t)b * NCH + chunk) * 8 + wave] = part;
      }
    }
    if (Huser) {
      constexpr int RS = 512 * 4 + 16;                    // row stride of the transposed half tile [16 f][512 nodes] fp32
      char* tst = reinterpret_cast<char*>(state);
      float* ub = Huser + (int64_t)b * ubstride + (int64_t)(chunk * FC) * N;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        lds_barrier();                                  // the last hop's reads (hf = 0) / the previous half's row reads are done
#pragma unroll
        for (int i = 0; i < STILES; ++i) {
          int wv = woff[i];
          asm volatile("" : "+v"(wv));
          const int node = wv >> 16;
          if ((node >> 9) == hf) {
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<float*>(tst + (q * 4 + c) * RS + (node & 511) * 4) = u[i][LASTU][c];
          }
        }
        lds_barrier();
        const int nhalf = (N - 512 * hf) < 512 ? (N - 512 * hf) : 512;      // valid nodes of this half (N % 4 == 0: whole 16-byte segments)
        const int segs = nhalf > 0 ? (nhalf >> 2) : 0;
        for (int idx = tid; idx < FC * segs; idx += STHREADS) {
          const int f = idx / segs, sg = idx - f * segs;
          *reinterpret_cast<float4*>(ub + (int64_t)f * N + 512 * hf + sg * 4) = *reinterpret_cast<const float4*>(tst + f * RS + sg * 16);
        }
      }
    }
    lds_barrier();      // `state` is free again before the next sequence's taps land in it
  }
}

template <int K, int HS, int XS>
int x3_launch(const void* xs3, const void* h03, void* hs3, const void* wpack3, const float* bias, const int32_t* tile_nodes,
              const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, float uni_w, float* Huser,
              int last_only, hipStream_t st, const float* r1, const float* bscale = nullptr, int64_t hu_stride = 0, int fixed_state = 0, int state_zero = 0) {
  constexpr int F = 32 * HS, G = 32 * XS, KS = HS + XS, NCH = F / FC;
  const size_t lds = (size_t)NP * FC * 4 + (size_t)3 * K * KS * 1024 + (size_t)entries * 32;
  if (lds > 160 * 1024) return GCRNN_ERR_UNSUPPORTED;
  auto kern = r1 ? ((fixed_state && state_zero) ? fused_step_x3_kernel<K, HS, XS, 0, true, true> : fused_step_x3_kernel<K, HS, XS, 0, false, true>)
                 : ((fixed_state && state_zero) ? fused_step_x3_kernel<K, HS, XS, 0, true> : fused_step_x3_kernel<K, HS, XS>);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  int64_t slots = cdiv(B, 8) * 8;
  const int64_t max_slots = (256 / NCH) / 8 * 8 > 0 ? (256 / NCH) / 8 * 8 : 8;
  if (slots > max_slots) slots = max_slots;
  const int64_t xstep = 3 * B * NP * G, hstep = 3 * B * NP * F;
  const uint16_t* x = (const uint16_t*)xs3;
  uint16_t* h = (uint16_t*)hs3;
  GCRNN_PRE_LAUNCH();
  for (int64_t t = 0; t < T; ++t) {
    // (fixed_state: every step reads h03 and writes its planes over the same block -- T independent one-step cells, the time gates' sub-cells)
    const uint16_t* hp = (t == 0 || fixed_state) ? (const uint16_t*)h03 : h + (t - 1) * hstep;
    float* hu = !Huser ? nullptr : (!last_only ? Huser + t * F * N : (t == T - 1 ? Huser : nullptr));
    kern<<<(unsigned)(slots * NCH), STHREADS, lds, st>>>(x + t * xstep, hp, fixed_state ? h : h + t * hstep, (const uint4*)wpack3, bias, tile_nodes, tile_off,
                                                        (const uint2*)ell_col4, hu, hu_stride ? hu_stride : (int64_t)(last_only ? 1 : T) * F * N, (int)entries, (int)B,
                                                        (int)N, uni_w, nullptr, nullptr, bscale ? bscale + t * B : nullptr, nullptr, r1);
  }
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}


// dpre[i] = dH[i] (1 - h[i]^2) on planes (the seed of the BPTT chain, t = T-1): fp32 from the planes, three planes out
__global__ void bwd_seed_x3_kernel(const uint16_t* __restrict__ dH3, const uint16_t* __restrict__ h3, uint16_t* __restrict__ out3, int64_t plane) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= plane) return;
  const float g = bf2f(dH3[i]) + bf2f(dH3[i + plane]) + bf2f(dH3[i + 2 * plane]);
  const float h = bf2f(h3[i + 2 * plane]) + bf2f(h3[i + plane]) + bf2f(h3[i]);
  uint16_t a, b, c;
  split3(g * (1.f - h * h), a, b, c);
  out3[i] = a; out3[i + plane] = b; out3[i + 2 * plane] = c;
}

template <int K, int HS>
int x3_backward_launch(const void* dHs3, const void* hs3, void* dpre3, void* dh03, const void* wpack3T, const int32_t* tile_nodes,
                       const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, float uni_w,
                       hipStream_t st, const float* r1, const float* gf = nullptr, const void* h03 = nullptr, float* gparts = nullptr) {
  constexpr int F = 32 * HS, NCH = F / FC;
  const size_t lds = (size_t)NP * FC * 4 + (size_t)3 * K * HS * 1024 + (size_t)entries * 32;
  if (lds > 160 * 1024) return GCRNN_ERR_UNSUPPORTED;
  auto kern = r1 ? fused_step_x3_kernel<K, HS, 0, 2, false, true> : fused_step_x3_kernel<K, HS, 0, 2>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  int64_t slots = cdiv(B, 8) * 8;
  const int64_t max_slots = (256 / NCH) / 8 * 8 > 0 ? (256 / NCH) / 8 * 8 : 8;
  if (slots > max_slots) slots = max_slots;
  const int64_t hstep = 3 * B * NP * F, plane = B * NP * F;
  const uint16_t* dH = (const uint16_t*)dHs3;
  const uint16_t* hs = (const uint16_t*)hs3;
  uint16_t* dp = (uint16_t*)dpre3;
  GCRNN_PRE_LAUNCH();
  bwd_seed_x3_kernel<<<(unsigned)cdiv(plane, 256), 256, 0, st>>>(dH + (T - 1) * hstep, hs + (T - 1) * hstep, dp + (T - 1) * hstep, plane);
  for (int64_t t = T - 1; t >= 1; --t)
    kern<<<(unsigned)(slots * NCH), STHREADS, lds, st>>>(nullptr, dp + t * hstep, dp + (t - 1) * hstep, (const uint4*)wpack3T, nullptr, tile_nodes,
                                                        tile_off, (const uint2*)ell_col4, nullptr, 0, (int)entries, (int)B, (int)N, uni_w,
                                                        dH + (t - 1) * hstep, hs + (t - 1) * hstep, gf ? gf + t * B : nullptr,
                                                        gparts ? gparts + t * B * (NCH * 8) : nullptr, r1);
  if (dh03)      // the raw gradient of the initial state (scaled by gf_0); with gparts: d loss / d gf_0 against the planes of h0
    kern<<<(unsigned)(slots * NCH), STHREADS, lds, st>>>(nullptr, dp, (uint16_t*)dh03, (const uint4*)wpack3T, nullptr, tile_nodes, tile_off,
                                                        (const uint2*)ell_col4, nullptr, 0, (int)entries, (int)B, (int)N, uni_w, nullptr,
                                                        gparts ? (const uint16_t*)h03 : nullptr, gf, gparts, r1);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// One raw filter pass sum_k S^k (z W_k) on the x3 step kernel (the chain step without an epilogue operand): z3 / out3 [3][B][NP][F] planes
template <int K, int HS>
int x3_filter_launch(const void* z3, void* out3, const void* wpack3, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4,
                     int64_t entries, int64_t B, int64_t N, float uni_w, hipStream_t st, const float* r1) {
  constexpr int F = 32 * HS, NCH = F / FC;
  const size_t lds = (size_t)NP * FC * 4 + (size_t)3 * K * HS * 1024 + (size_t)entries * 32;
  if (lds > 160 * 1024) return GCRNN_ERR_UNSUPPORTED;
  auto kern = r1 ? fused_step_x3_kernel<K, HS, 0, 2, false, true> : fused_step_x3_kernel<K, HS, 0, 2>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  int64_t slots = cdiv(B, 8) * 8;
  const int64_t max_slots = (256 / NCH) / 8 * 8 > 0 ? (256 / NCH) / 8 * 8 : 8;
  if (slots > max_slots) slots = max_slots;
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)(slots * NCH), STHREADS, lds, st>>>(nullptr, (const uint16_t*)z3, (uint16_t*)out3, (const uint4*)wpack3, nullptr, tile_nodes, tile_off,
                                                      (const uint2*)ell_col4, nullptr, 0, (int)entries, (int)B, (int)N, uni_w, nullptr, nullptr, nullptr, nullptr, r1);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// ------------------------------------------------------------------------------------------
// fp32 weight gradient of the cell (adjoint of the taps, reference graphML.py:134-135 under autograd):
//     dW_k[f'][j] = sum_{t,b,n} du_k[t,b][n][f'] z[t,b][n][j],   du_0 = dpre_t,  du_k = S du_{k-1},   z = [h_{t-1} | x_t]
// with EXACT fp32 products: v_mfma_f32_16x16x4_f32 (the fp32 matrix instruction: 1/16 of the bf16 rate, 4.4 ms of matrix time per
// training step at B = 256 -- the bf16 kernel's six-plane alternative would need three transposed images of du_k in LDS).
// One workgroup = (item, 16-feature chunk of dpre); wave w owns input-feature tile w of z. Node index = contraction dimension:
//   B operand: lane (j, kq) loads z[j][16 m + 4 kq .. + 3] = 16 bytes of the fp32 USER layout (node-contiguous rows) per 4 MFMAs;
//   A operand: du_k transposed in LDS, fp32 [16 f'][1028]: lane (f', kq) reads the same four nodes with one ds_read_b128 (the row
//   stride of 1028 words puts the 16 lanes of a read group on 16 different bank quads).
// du_k is re-assembled from the three planes of dpre (k = 0) or comes from the adjoint hop on the fp32 state image (uniform stream).
// Accumulators persist over the workgroup's items; per-slot partial sums, added by the caller in a fixed order (deterministic).
// LDS: state image 64 KiB | column words 32 B x entries | transposed du_k 16 x 4112 B.
// ------------------------------------------------------------------------------------------
constexpr int DUT_STRIDE = 4112;

template <int K, int HS, int XS>
__global__ __launch_bounds__(512) void fused_wgrad_f32_kernel(
    const uint16_t* __restrict__ dpre3,      // [T][3][B][NP][F] bf16 planes, sequence-major
    const float* __restrict__ Xuser,         // [B][T][G][N] fp32
    const float* __restrict__ Huser,         // [B][T][F][N] fp32 (forward output)
    const float* __restrict__ h0user,        // [B][F][N] fp32
    float* __restrict__ dW,                  // [slots][F][K][F+G] fp32 partial sums (plain stores)
    float* __restrict__ dbsum,               // [slots][F] fp32 partials of 2 sum_{t,b,n} dpre (or null)
    const int32_t* __restrict__ tile_nodes, const int32_t* __restrict__ tile_off, const uint2* __restrict__ ell_col4,
    int entries, int B, int Tn, int N, float uni_w,
    const float* __restrict__ gi, const float* __restrict__ gf,        // time-gated cell (or null): item (t, b) enters the input / state columns with weight gi / gf [T][B]
    int h_is_h0,                                                       // != 0: every item's state operand is h0 (the time gates' sub-cells); a null state pointer = zeros: its tiles are skipped
    const float* __restrict__ r1) {                                    // rank-1-weighted graph (or null): the [4][NP] table of fused_step_x3_kernel; du_k = b (.) sum (a (.) du_{k-1})
  static_assert(GCRNN_HOP_ASM && TILES == 8, "uniform asm hop stream");
  constexpr int F = 32 * HS, G = 32 * XS, C = F + G, NCH = F / FC, JT = C / 16, HT = TILES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* state = reinterpret_cast<float*>(smem);
  uint2* lcol4 = reinterpret_cast<uint2*>(smem + NP * FC * 4);
  char* dut = smem + NP * FC * 4 + entries * 32;
  float* lred = reinterpret_cast<float*>(dut + 16 * DUT_STRIDE);      // [WAVES][16] bias partials

  const int L = blockIdx.x;
  const int grp = L / (8 * NCH), rem = L - grp * (8 * NCH);
  const int chunk = rem >> 3, it0 = grp * 8 + (rem & 7);
  const int seq_slots = (gridDim.x / (8 * NCH)) * 8;
  const int items = B * Tn;
  if (it0 >= items) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  {
    const int n = (entries >> 2) * 16;
    for (int i = tid; i < n; i += 512) lcol4[i] = ell_col4[i];
  }
  int tbeg[TILES], tend[TILES], woff[TILES];
#pragma unroll
  for (int i = 0; i < TILES; ++i) {
    tbeg[i] = tile_off[wave * TILES + i];
    tend[i] = tile_off[wave * TILES + i + 1];
    woff[i] = tile_nodes[(wave * TILES + i) * 16 + r] ^ (q << 4);
  }
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  if (lds0 != 0) __builtin_trap();
  const uint32_t qx = (uint32_t)(q * 16);
  const uint32_t lds_col = lds0 + NP * FC * 4;
  f32x4 accD[K];
#pragma unroll
  for (int k = 0; k < K; ++k) accD[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
  const bool has_tile = wave < JT;
  const bool is_x = wave >= F / 16;
  const int jrow = (is_x ? wave * 16 - F : wave * 16) + r;
  __syncthreads();

  for (int it = it0; it < items; it += seq_slots) {
    const int t = it / B, b = it - t * B;
    // (one descriptor per time step: the three planes of a step span 3 B NP F 2 bytes, all T of them would overflow 32-bit offsets)
    const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dpre3) + (int64_t)t * 3 * B * (NP * F), 0,
                                                                            3 * B * (NP * F * 2), 0x00020000);
    const float* zsrc;
    int zrows;
    if (is_x) { zsrc = Xuser + ((int64_t)b * Tn + t) * G * N; zrows = G; }
    else if (t > 0 && !h_is_h0) { zsrc = Huser + ((int64_t)b * Tn + (t - 1)) * F * N; zrows = F; }
    else { zsrc = h0user ? h0user + (int64_t)b * F * N : nullptr; zrows = F; }
    const bool tile_on = has_tile && zsrc != nullptr;      // (wave-uniform: a zero state operand contributes nothing)
    const __amdgpu_buffer_rsrc_t rsrc_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zsrc), 0, tile_on ? zrows * N * 4 : 0, 0x00020000);
    // gi (A(S) x_t + b) + gf (B(S) h_{t-1} + b): the operand columns scaled in fp32 (exactly the forward's scaled operands), the one bias weighted gi + gf
    const float zsc = gi ? (is_x ? gi[it] : gf[it]) : 1.f;
    const float bsc = gi ? gi[it] + gf[it] : 2.f;
    // du_0 = dpre chunk of this item, fp32 from its planes
    f32x4 cur[TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      int wv = woff[i];
      asm volatile("" : "+v"(wv));
      const int eoff = (wv >> 16) * (F * 2) + (chunk * FC + q * 4) * 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 2; p >= 0; --p) {
        const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc_d, eoff, (p * B + b) * (NP * F * 2), 0);
        v[0] += bf2f((uint16_t)(w2[0] & 0xffffu)); v[1] += bf2f((uint16_t)(w2[0] >> 16));
        v[2] += bf2f((uint16_t)(w2[1] & 0xffffu)); v[3] += bf2f((uint16_t)(w2[1] >> 16));
      }
      cur[i] = v;
      bacc += bsc * v;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      // du_k -> the swizzled state image (for the next hop) and the transposed image (A operand)
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        int wv = woff[i];
        asm volatile("" : "+v"(wv));
        if (k < K - 1) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(state) + (wv & 0xffff)) = r1 ? cur[i] * r1[wv >> 16] : cur[i];
        const int node = wv >> 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<float*>(dut + (q * 4 + c) * DUT_STRIDE + node * 4) = cur[i][c];
      }
      lds_barrier();
      if (tile_on) {
        // D_k += du_k^T z: 64 groups of 16 nodes, 4 exact-fp32 MFMAs each. The z fragments of the NEXT eight groups are requested
        // before the 32 MFMAs of the current eight (two register sets, ping-pong): the L2 latency hides under 1024 matrix cycles.
        auto load_z = [&](int m0, f32x4 (&zz)[8]) {
#pragma unroll
          for (int mm = 0; mm < 8; ++mm) {
            const int n4 = 16 * (m0 + mm) + 4 * q;
            const int zo = (n4 < N) ? (jrow * N + n4) * 4 : 0x7ffffff0;        // nodes >= N: out of the descriptor's range -> zeros
            zz[mm] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_z, zo, 0, 0));
          }
        };
        auto scale_z = [&](f32x4 (&zz)[8]) {
          if (gi) {
#pragma unroll
            for (int mm = 0; mm < 8; ++mm) zz[mm] *= zsc;
          }
        };
        auto mma8 = [&](int m0, const f32x4 (&zz)[8]) {
          f32x4 da[8];
#pragma unroll
          for (int mm = 0; mm < 8; ++mm) da[mm] = *reinterpret_cast<const f32x4*>(dut + r * DUT_STRIDE + (16 * (m0 + mm) + 4 * q) * 4);
#pragma unroll
          for (int mm = 0; mm < 8; ++mm)
#pragma unroll
            for (int e = 0; e < 4; ++e) accD[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[mm][e], zz[mm][e], accD[k], 0, 0, 0);
        };
        f32x4 z0[8], z1[8];
        load_z(0, z0);
#pragma unroll 1
        for (int m0 = 0; m0 < NP / 16; m0 += 16) {
          load_z(m0 + 8, z1);
          scale_z(z0);
          mma8(m0, z0);
          if (m0 + 16 < NP / 16) load_z(m0 + 16, z0);
          scale_z(z1);
          mma8(m0 + 8, z1);
        }
      }
      if (k < K - 1) {
        LGKM_WAIT(0);
#define GCRNN_WF_INIT(i) f32x4{0.f, 0.f, 0.f, 0.f}
#define GCRNN_WF_STORE(i, a) cur[i] = a
        GCRNN_HOP_ASM_UNI_STREAM(GCRNN_WF_INIT, GCRNN_WF_STORE);
#undef GCRNN_WF_INIT
#undef GCRNN_WF_STORE
        if (r1) {
#pragma unroll
          for (int i = 0; i < TILES; ++i) {
            int wv = woff[i];
            asm volatile("" : "+v"(wv));
            cur[i] *= r1[3 * NP + (wv >> 16)];
          }
        }
      }
      lds_barrier();
    }
  }
  if (has_tile) {
    float* dWs = dW + (int64_t)it0 * (F * K * C);
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        dWs[((int64_t)(chunk * FC + q * 4 + c) * K + k) * C + wave * 16 + r] = accD[k][c];
  }
  if (dbsum) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = bacc[c];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
      if (r == 0) lred[wave * FC + q * 4 + c] = v;                    // (weight 2 -- or gi + gf -- already in: the one bias enters both filters)
    }
    __syncthreads();
    if (tid < FC) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += lred[w * FC + tid];
      dbsum[(int64_t)it0 * F + chunk * FC + tid] = v;
    }
  }
}

template <int K, int HS, int XS>
int x3_wgrad_launch(const void* dpre3, const void* Xuser, const void* Huser, const void* h0user, float* dW, float* dbsum,
                    const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T,
                    int64_t N, float uni_w, hipStream_t st, const float* r1, const float* gi = nullptr, const float* gf = nullptr, int h_is_h0 = 0) {
  constexpr int F = 32 * HS, NCH = F / FC;
  const size_t lds = (size_t)NP * FC * 4 + (size_t)entries * 32 + 16 * DUT_STRIDE + WAVES * FC * 4;
  if (lds > 160 * 1024) return GCRNN_ERR_UNSUPPORTED;
  auto kern = fused_wgrad_f32_kernel<K, HS, XS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  const int64_t slots = gcrnn_fused_wgrad_slots(B * T, F);
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)(slots * NCH), 512, lds, st>>>((const uint16_t*)dpre3, (const float*)Xuser, (const float*)Huser, (const float*)h0user, dW, dbsum,
                                                   tile_nodes, tile_off, (const uint2*)ell_col4, (int)entries, (int)B, (int)T, (int)N, uni_w, gi, gf, h_is_h0, r1);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

}  // namespace

// Shapes the fp32-accurate fused path is built for: F = G = 64 / 32 and F = 64 with G <= 32 (padded to 32), K in 2..5,
// N <= 1024 with N % 4 == 0, a uniform-weight graph image of `entries` ELL entries that fits next to the state and three tap planes.
extern "C" int gcrnn_fused_x3_supported(int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries) {
  if (N <= 0 || N > NP || N % 4 || entries < 0 || entries % 4 || K < 2 || K > 5) return 0;
  if (!((F == 64 && (G == 64 || G == 32)) || (F == 32 && G == 32))) return 0;
  return (int64_t)NP * FC * 4 + 3 * K * ((F + G) / 32) * 1024 + entries * 32 <= 160 * 1024;
}

// user fp32 [B][T][C][N] -> bf16 planes [T][3][B][NPad][C] (C even)
extern "C" int gcrnn_pack_seq_major_x3(const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N, int64_t NPad, void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || C <= 0 || C % 2 || N <= 0 || NPad < N || B * T > 65535 || cdiv(C, 32) > 65535) return GCRNN_ERR_BAD_SHAPE;
  return seq_pack_x3_launch(src, dst, B, T, C, N, NPad, nullptr, nullptr, 0, 0, as_stream(stream));
}

// gcrnn_pack_seq_major_x3 of a derived tensor: v' = item_scale[t][b] * rowmul[c][n] * (one_minus_square ? 1 - v^2 : v), every factor optional
// (item_scale [T][B], rowmul [C][N] fp32 or NULL); src_seq_stride: elements between consecutive sequences of src (0 = T C N: contiguous; one
// step of a [B][T'][C][N] tensor is T = 1 with stride T' C N). The time-gated cell at fp32 accuracy packs its scaled operands gi x_t / gf h_{t-1} and its
// gate cells' upstream gradient d logit * w * (1 - c^2) with it (Utils/graphML.py:2362-2374, 2420-2423 and their autograd).
extern "C" int gcrnn_pack_seq_major_x3_ex(const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N, int64_t NPad,
                                          const float* item_scale, const float* rowmul, int one_minus_square, int64_t src_seq_stride,
                                          void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || C <= 0 || C % 2 || N <= 0 || NPad < N || B * T > 65535 || cdiv(C, 32) > 65535) return GCRNN_ERR_BAD_SHAPE;
  if (src_seq_stride < 0 || (src_seq_stride && src_seq_stride < T * C * N)) return GCRNN_ERR_BAD_SHAPE;
  return seq_pack_x3_launch(src, dst, B, T, C, N, NPad, item_scale, rowmul, one_minus_square, src_seq_stride, as_stream(stream));
}

// Per item (t, b) of plane arrays a3, b3 [T][3][B][NPad][C]: out_ab[t][b] = <a, b> (b3 / out_ab may be NULL), out_av[t][b] = sum_{n,c} a[n][c]
// vec[c] (vec [C] fp32; vec / out_av may be NULL), fp32 values re-assembled from the planes, fixed summation order. The time-gated BPTT
// reads d loss / d gi_t = <A(S) x_t, dpre_t> + <b, sum_n dpre_t> off it.
extern "C" int gcrnn_x3_item_dots(const void* a3, const void* b3, const float* vec, float* out_ab, float* out_av, int64_t B, int64_t T,
                                  int64_t NPad, int64_t C, void* stream) {
  if (!a3 || (b3 && !out_ab) || (vec && !out_av)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || NPad <= 0 || C <= 0 || C % 2 || B * T > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  x3_item_dots_kernel<<<(unsigned)(B * T), 256, 0, as_stream(stream)>>>((const uint16_t*)a3, (const uint16_t*)b3, vec, b3 ? out_ab : nullptr,
                                                                        vec ? out_av : nullptr, (int)B, (int)NPad, (int)C);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// One step of the node-gated cell behind its two x3 filter passes (x3_node_step_kernel): ya3 / yb3 / h3 [3][B][NPad][F] bf16 planes, ni / nf fp32
// [B][NPad] (this step's node gates, time gate folded in by the caller; rows >= N are not read), bias fp32 [F] or NULL, Huser fp32 or NULL:
// element (b, f, n) at b huser_seq_stride + f N + n (one step of H [B][T][F][N]: stride T F N). F in {32, 64}, N % 4 == 0, NPad % 64 == 0.
extern "C" int gcrnn_x3_node_gate_step(const void* ya3, const void* yb3, const float* ni, const float* nf, const float* bias, void* h3, void* Huser,
                                       int64_t huser_seq_stride, int64_t B, int64_t N, int64_t NPad, int64_t F, void* stream) {
  if (!ya3 || !yb3 || !ni || !nf || !h3) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || B > 65535 || N <= 0 || N > NPad || N % 4 || NPad % 64 || (F != 32 && F != 64) || huser_seq_stride < 0 || huser_seq_stride % 4) return GCRNN_ERR_BAD_SHAPE;
  if (Huser && (reinterpret_cast<uintptr_t>(Huser) & 15)) return GCRNN_ERR_BAD_SHAPE;
  const dim3 grid((unsigned)(NPad / 64), (unsigned)B);
  GCRNN_PRE_LAUNCH();
  if (F == 64)
    x3_node_step_kernel<64><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)ya3, (const uint16_t*)yb3, ni, nf, bias, (uint16_t*)h3, (float*)Huser,
                                                                 huser_seq_stride, (int)B, (int)N, (int)NPad);
  else
    x3_node_step_kernel<32><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)ya3, (const uint16_t*)yb3, ni, nf, bias, (uint16_t*)h3, (float*)Huser,
                                                                 huser_seq_stride, (int)B, (int)N, (int)NPad);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// fp32 taps wA [F][Kin][G], wB [F][Kst][F] -> wpack3 [3][F/16][K][(F+G)/32][64][8] bf16 planes (K = max(Kin, Kst))
extern "C" int gcrnn_fused_pack_weights_x3(const void* wA, const void* wB, void* wpack3, int64_t F, int64_t G, int64_t Kin, int64_t Kst,
                                           void* stream) {
  if (!wA || !wB || !wpack3) return GCRNN_ERR_NULL_POINTER;
  if (F <= 0 || G < 0 || F % FC || (F + G) % 32 || Kin <= 0 || Kst <= 0) return GCRNN_ERR_BAD_SHAPE;      // (G = 0: a state-only operand, the BPTT chain's transposed taps)
  const int K = (int)(Kin > Kst ? Kin : Kst);
  const int64_t total = (F / FC) * K * ((F + G) / 32) * 64 * 8;
  GCRNN_PRE_LAUNCH();
  pack_weights_x3_kernel<<<(unsigned)cdiv(total, 256), 256, 0, as_stream(stream)>>>((const float*)wA, (const float*)wB, (uint16_t*)wpack3,
                                                                                     (int)F, (int)G, (int)Kin, (int)Kst, K);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

// T launches of the x3 step kernel: xs3 [T][3][B][NPad][G], h03 [3][B][NPad][F], hs3 [T][3][B][NPad][F] (all bf16 planes),
// wpack3 from gcrnn_fused_pack_weights_x3, bias fp32 [F] or NULL, graph arrays of a UNIFORM plan (gcrnn_ell_fill_z; uniform_w = the
// one weight), Huser fp32 [B][T][F][N] (or [B][1][F][N] with last_only) or NULL.
extern "C" int gcrnn_fused_forward_x3(const void* xs3, const void* h03, void* hs3, const void* wpack3, const float* bias,
                                      const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                      int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, double uniform_w, void* Huser,
                                      int last_only, const float* rank1, void* stream) {
  if (!xs3 || !h03 || !hs3 || !wpack3 || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || uniform_w == 0.0 || !gcrnn_fused_x3_supported(N, F, G, K, entries)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * (F > G ? F : G) * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;         // 32-bit buffer offsets
  if (Huser && (reinterpret_cast<uintptr_t>(Huser) & 15)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_X3_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) \
    return x3_launch<KK, HH, XX>(xs3, h03, hs3, wpack3, bias, tile_nodes, tile_off, ell_col4, entries, B, T, N, (float)uniform_w, (float*)Huser, last_only, st, rank1);
  GCRNN_X3_CASE(5, 2, 2) GCRNN_X3_CASE(4, 2, 2) GCRNN_X3_CASE(3, 2, 2) GCRNN_X3_CASE(2, 2, 2)
  GCRNN_X3_CASE(5, 2, 1) GCRNN_X3_CASE(4, 2, 1) GCRNN_X3_CASE(3, 2, 1) GCRNN_X3_CASE(2, 2, 1)
  GCRNN_X3_CASE(5, 1, 1) GCRNN_X3_CASE(4, 1, 1) GCRNN_X3_CASE(3, 1, 1) GCRNN_X3_CASE(2, 1, 1)
#undef GCRNN_X3_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// The same forward with a per-(t, b) weight of the bias instead of 2 (bias_scale [T][B] fp32 or NULL) and an explicit distance between
// consecutive sequences of Huser (huser_seq_stride elements, 0 = T F N): what the time-gated cell at fp32 accuracy is composed from --
// gi (A(S) x_t + b) + gf (B(S) h_{t-1} + b) = A(S)(gi x_t) + B(S)(gf h_{t-1}) + (gi + gf) b (graphML.py:2420-2423): the caller scales the
// operands (exact in fp32, before they are cut into planes) and passes gi + gf here.
extern "C" int gcrnn_fused_forward_x3_scaled(const void* xs3, const void* h03, void* hs3, const void* wpack3, const float* bias,
                                             const float* bias_scale, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4,
                                             int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K,
                                             double uniform_w, void* Huser, int64_t huser_seq_stride, const float* rank1, void* stream) {
  if (!xs3 || !h03 || !hs3 || !wpack3 || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || uniform_w == 0.0 || !gcrnn_fused_x3_supported(N, F, G, K, entries)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * (F > G ? F : G) * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;         // 32-bit buffer offsets
  if (Huser && (reinterpret_cast<uintptr_t>(Huser) & 15)) return GCRNN_ERR_BAD_SHAPE;
  if (huser_seq_stride < 0 || (huser_seq_stride % 4)) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_X3_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) \
    return x3_launch<KK, HH, XX>(xs3, h03, hs3, wpack3, bias, tile_nodes, tile_off, ell_col4, entries, B, T, N, (float)uniform_w, (float*)Huser, 0, st, rank1, \
                                 bias_scale, huser_seq_stride);
  GCRNN_X3_CASE(5, 2, 2) GCRNN_X3_CASE(4, 2, 2) GCRNN_X3_CASE(3, 2, 2) GCRNN_X3_CASE(2, 2, 2)
  GCRNN_X3_CASE(5, 2, 1) GCRNN_X3_CASE(4, 2, 1) GCRNN_X3_CASE(3, 2, 1) GCRNN_X3_CASE(2, 2, 1)
  GCRNN_X3_CASE(5, 1, 1) GCRNN_X3_CASE(4, 1, 1) GCRNN_X3_CASE(3, 1, 1) GCRNN_X3_CASE(2, 1, 1)
#undef GCRNN_X3_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// The time gates' sub-cells at fp32 accuracy (Utils/graphML.py:2362-2366): T x B independent ONE-step cells c[t][b] = tanh(A_g(S) x_t + B_g(S) h0
// + 2 b_g), every one from the same initial state: gcrnn_fused_forward_x3 whose every step reads h03 (xs3 [T][3][B][NPad][G] the planes of X as
// the recurrence packs them -- no per-item copies of X or h0; state_zero != 0: h0 is all zeros, train_rnn.py:256, and its half of the tap
// products is skipped -- exact). scratch3 [3][B][NPad][F] receives (and re-receives) a step's planes; Cuser fp32
// [B][T][F][N] the gate states, user layout.
extern "C" int gcrnn_fused_gate_cells_x3(const void* xs3, const void* h03, void* scratch3, const void* wpack3, const float* bias,
                                         const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                         int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, double uniform_w, void* Cuser,
                                         int state_zero, const float* rank1, void* stream) {
  if (!xs3 || !h03 || !scratch3 || !wpack3 || !tile_nodes || !tile_off || !ell_col4 || !Cuser) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || uniform_w == 0.0 || !gcrnn_fused_x3_supported(N, F, G, K, entries)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * (F > G ? F : G) * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (reinterpret_cast<uintptr_t>(Cuser) & 15) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_X3_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) \
    return x3_launch<KK, HH, XX>(xs3, h03, scratch3, wpack3, bias, tile_nodes, tile_off, ell_col4, entries, B, T, N, (float)uniform_w, (float*)Cuser, 0, st, rank1, \
                                 nullptr, 0, 1, state_zero);
  GCRNN_X3_CASE(5, 2, 2) GCRNN_X3_CASE(4, 2, 2) GCRNN_X3_CASE(3, 2, 2) GCRNN_X3_CASE(2, 2, 2)
  GCRNN_X3_CASE(5, 2, 1) GCRNN_X3_CASE(4, 2, 1) GCRNN_X3_CASE(3, 2, 1) GCRNN_X3_CASE(2, 2, 1)
  GCRNN_X3_CASE(5, 1, 1) GCRNN_X3_CASE(4, 1, 1) GCRNN_X3_CASE(3, 1, 1) GCRNN_X3_CASE(2, 1, 1)
#undef GCRNN_X3_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// BPTT data chain at fp32 accuracy (x3): dHs3 / hs3 [T][3][B][NPad][F] planes of the upstream gradients and of the states (the forward's
// hs3), dpre3 [T][3][B][NPad][F] (out), dh03 [3][B][NPad][F] or NULL (out: the raw gradient of the initial state), wpack3T = the
// TRANSPOSED state taps packed as a state-only operand (gcrnn_fused_pack_weights_x3 with G = 0), graph arrays of the ADJOINT uniform
// plan. One seed launch + T - 1 (+ 1) launches of the x3 step kernel with the chain epilogue. Reference: autograd of
// Utils/graphML.py:2420-2423 in the drivers' precision (kStepPredGRNNs.py:44).
extern "C" int gcrnn_fused_backward_data_x3(const void* dHs3, const void* hs3, void* dpre3, void* dh03, const void* wpack3T,
                                            const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                            int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, double uniform_w, const float* rank1, void* stream) {
  if (!dHs3 || !hs3 || !dpre3 || !wpack3T || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || uniform_w == 0.0 || !gcrnn_fused_x3_supported(N, F, F, K, entries)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * F * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_X3B_CASE(KK, HH) \
  if (K == KK && F == 32 * HH) return x3_backward_launch<KK, HH>(dHs3, hs3, dpre3, dh03, wpack3T, tile_nodes, tile_off, ell_col4, entries, B, T, N, (float)uniform_w, st, rank1);
  GCRNN_X3B_CASE(5, 2) GCRNN_X3B_CASE(4, 2) GCRNN_X3B_CASE(3, 2) GCRNN_X3B_CASE(2, 2)
  GCRNN_X3B_CASE(5, 1) GCRNN_X3B_CASE(4, 1) GCRNN_X3B_CASE(3, 1) GCRNN_X3B_CASE(2, 1)
#undef GCRNN_X3B_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// The same chain for the TIME-GATED cell (graphML.py:2357-2374, 2420-2423 under autograd): dpre_{t-1} = (gf_t sum_k S^k (dpre_t B_k^T) + dH_{t-1})
// (1 - h_{t-1}^2), gf [T][B] fp32; dh03 (required) receives gf_0 times the raw gradient of the initial state. dgf_parts (or NULL; needs h03 =
// the planes of h0 [3][B][NPad][F]): [T][B][F/16 * 8] fp32 partials of <h_{t-1}, sum_k S^k (dpre_t B_k^T)> = <B(S) h_{t-1}, dpre_t>, the
// filter part of d loss / d gf_t (the caller adds them in a fixed order and the bias part).
extern "C" int gcrnn_fused_backward_data_x3_gated(const void* dHs3, const void* hs3, void* dpre3, void* dh03, const void* wpack3T,
                                                  const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4, int64_t entries,
                                                  int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, double uniform_w, const float* gf,
                                                  const void* h03, float* dgf_parts, const float* rank1, void* stream) {
  if (!dHs3 || !hs3 || !dpre3 || !dh03 || !wpack3T || !tile_nodes || !tile_off || !ell_col4 || !gf) return GCRNN_ERR_NULL_POINTER;
  if (dgf_parts && !h03) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || uniform_w == 0.0 || !gcrnn_fused_x3_supported(N, F, F, K, entries)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * F * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_X3B_CASE(KK, HH) \
  if (K == KK && F == 32 * HH) return x3_backward_launch<KK, HH>(dHs3, hs3, dpre3, dh03, wpack3T, tile_nodes, tile_off, ell_col4, entries, B, T, N, (float)uniform_w, st, rank1, gf, h03, dgf_parts);
  GCRNN_X3B_CASE(5, 2) GCRNN_X3B_CASE(4, 2) GCRNN_X3B_CASE(3, 2) GCRNN_X3B_CASE(2, 2)
  GCRNN_X3B_CASE(5, 1) GCRNN_X3B_CASE(4, 1) GCRNN_X3B_CASE(3, 1) GCRNN_X3B_CASE(2, 1)
#undef GCRNN_X3B_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// One graph filter without bias or nonlinearity at fp32 accuracy: out = sum_k S^k (z W_k) (Utils/graphML.py:47-140 LSIGF), z3 / out3
// [3][B][NPad][F] bf16 planes (F features in and out), wpack3 = the filter's taps packed as a state-only operand
// (gcrnn_fused_pack_weights_x3 with G = 0), graph arrays of a uniform plan. The time-gated BPTT reads d loss / d gi_t = <A(S) x_t + b, dpre_t>
// off it (items = all (t, b)).
extern "C" int gcrnn_fused_filter_x3(const void* z3, void* out3, const void* wpack3, const int32_t* tile_nodes, const int32_t* tile_off,
                                     const void* ell_col4, int64_t entries, int64_t B, int64_t N, int64_t F, int64_t K, double uniform_w,
                                     const float* rank1, void* stream) {
  if (!z3 || !out3 || !wpack3 || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || uniform_w == 0.0 || !gcrnn_fused_x3_supported(N, F, F, K, entries)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * F * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_X3F_CASE(KK, HH) \
  if (K == KK && F == 32 * HH) return x3_filter_launch<KK, HH>(z3, out3, wpack3, tile_nodes, tile_off, ell_col4, entries, B, N, (float)uniform_w, st, rank1);
  GCRNN_X3F_CASE(5, 2) GCRNN_X3F_CASE(4, 2) GCRNN_X3F_CASE(3, 2) GCRNN_X3F_CASE(2, 2)
  GCRNN_X3F_CASE(5, 1) GCRNN_X3F_CASE(4, 1) GCRNN_X3F_CASE(3, 1) GCRNN_X3F_CASE(2, 1)
#undef GCRNN_X3F_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// fp32 weight gradient of the fused cell on exact-fp32 matrix instructions: dW [slots][F][K][F+G] / dbsum [slots][F] (or NULL) per-slot
// partial sums (slots = gcrnn_fused_wgrad_slots(B T, F); the caller adds them in a fixed order), dpre3 from
// gcrnn_fused_backward_data_x3, X / H / h0 fp32 in the USER layout (H = the forward's output), adjoint uniform plan.
extern "C" int gcrnn_fused_backward_weight_f32(const void* dpre3, const void* Xuser, const void* Huser, const void* h0user, float* dW,
                                               float* dbsum, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4,
                                               int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K,
                                               double uniform_w, const float* rank1, void* stream) {
  if (!dpre3 || !Xuser || !Huser || !h0user || !dW || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || N % 4 || entries < 0 || entries % 4 || uniform_w == 0.0 || B * T > (1 << 24)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * F * 2) > 2147483647LL || (int64_t)(F > G ? F : G) * N * 4 > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;      // 32-bit buffer offsets (per time step)
  if ((reinterpret_cast<uintptr_t>(Xuser) | reinterpret_cast<uintptr_t>(Huser) | reinterpret_cast<uintptr_t>(h0user)) & 15) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_WF_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) return x3_wgrad_launch<KK, HH, XX>(dpre3, Xuser, Huser, h0user, dW, dbsum, tile_nodes, tile_off, ell_col4, entries, B, T, N, (float)uniform_w, st, rank1);
  GCRNN_WF_CASE(5, 2, 2) GCRNN_WF_CASE(4, 2, 2) GCRNN_WF_CASE(3, 2, 2) GCRNN_WF_CASE(2, 2, 2)
  GCRNN_WF_CASE(5, 2, 1) GCRNN_WF_CASE(4, 2, 1) GCRNN_WF_CASE(3, 2, 1) GCRNN_WF_CASE(2, 2, 1)
  GCRNN_WF_CASE(5, 1, 1) GCRNN_WF_CASE(4, 1, 1) GCRNN_WF_CASE(3, 1, 1) GCRNN_WF_CASE(2, 1, 1)
#undef GCRNN_WF_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// The same weight gradient for the TIME-GATED cell: item (t, b) enters the input-filter columns with weight gi[t][b] and the state-filter
// columns with gf[t][b] (operands scaled in fp32 as the forward scaled them), dbsum = per-slot partials of sum (gi + gf) sum_n dpre
// (gi = gf = NULL: weights 1 and 2). h_is_h0 != 0: every item's state operand is h0 (the time gates' sub-cells read (x_t, h0),
// graphML.py:2362-2374; Huser unused); h0user = NULL then means a zero initial state (train_rnn.py:256): the state columns stay zero.
extern "C" int gcrnn_fused_backward_weight_f32_gated(const void* dpre3, const void* Xuser, const void* Huser, const void* h0user, float* dW,
                                                     float* dbsum, const int32_t* tile_nodes, const int32_t* tile_off, const void* ell_col4,
                                                     int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K,
                                                     double uniform_w, const float* gi, const float* gf, int h_is_h0, const float* rank1, void* stream) {
  if (!dpre3 || !Xuser || !dW || !tile_nodes || !tile_off || !ell_col4 || (!gi) != (!gf)) return GCRNN_ERR_NULL_POINTER;
  if (!h_is_h0 && (!Huser || !h0user)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || N % 4 || entries < 0 || entries % 4 || uniform_w == 0.0 || B * T > (1 << 24)) return GCRNN_ERR_BAD_SHAPE;
  if (3 * B * (NP * F * 2) > 2147483647LL || (int64_t)(F > G ? F : G) * N * 4 > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(Xuser) | reinterpret_cast<uintptr_t>(Huser) | reinterpret_cast<uintptr_t>(h0user)) & 15) return GCRNN_ERR_BAD_SHAPE;
  hipStream_t st = as_stream(stream);
#define GCRNN_WF_CASE(KK, HH, XX) \
  if (K == KK && F == 32 * HH && G == 32 * XX) return x3_wgrad_launch<KK, HH, XX>(dpre3, Xuser, Huser, h0user, dW, dbsum, tile_nodes, tile_off, ell_col4, entries, B, T, N, (float)uniform_w, st, rank1, gi, gf, h_is_h0);
  GCRNN_WF_CASE(5, 2, 2) GCRNN_WF_CASE(4, 2, 2) GCRNN_WF_CASE(3, 2, 2) GCRNN_WF_CASE(2, 2, 2)
  GCRNN_WF_CASE(5, 2, 1) GCRNN_WF_CASE(4, 2, 1) GCRNN_WF_CASE(3, 2, 1) GCRNN_WF_CASE(2, 2, 1)
  GCRNN_WF_CASE(5, 1, 1) GCRNN_WF_CASE(4, 1, 1) GCRNN_WF_CASE(3, 1, 1) GCRNN_WF_CASE(2, 1, 1)
#undef GCRNN_WF_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// Can the fp32-accurate BPTT (gcrnn_fused_backward_data_x3 + gcrnn_fused_backward_weight_f32) take this cell? The forward's shapes,
// and an ADJOINT uniform graph image of `entries_adj` ELL entries next to the fp32 state image and the transposed fp32 image of du_k.
extern "C" int gcrnn_fused_x3_training_supported(int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries_adj) {
  if (!gcrnn_fused_x3_supported(N, F, G, K, entries_adj) || !gcrnn_fused_x3_supported(N, F, F, K, entries_adj)) return 0;
  return (int64_t)NP * FC * 4 + entries_adj * 32 + 16 * DUT_STRIDE + WAVES * FC * 4 <= 160 * 1024;
}
