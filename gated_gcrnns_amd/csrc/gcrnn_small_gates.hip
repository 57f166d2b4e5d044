// Time gates of the small-graph regime on the matrix cores (reference GGCRNNCell time gating, graphML.py:2357-2374):
//   c_t     = tanh( A_g(S) x_t + b_g + B_g(S) h0 + b_g )          one GCRNN cell evaluation on (x_t, h0) -- h0, not h_{t-1}
//   gate_t  = sigmoid( lw . vec_{F,N}(c_t) + lb )                   scalar per sequence and step
// for the input gate and the forget gate (two parameter sets). Since the state operand is always h0,
//   Yb = B_g(S) h0 + 2 b_g   is computed ONCE per sequence (K-1 dense hops of h0), and a time step costs only the hops of the
// G input channels, a K*G-deep tap product, F*N tanh and one block reduction. The composed path materialises
// [T][N][B][F] tensors for this (5 ms of the 9 ms time-gated cfg4 training step); here nothing leaves the CU.
// One workgroup per (sequence, gate): grid (B, 2). Everything in LDS / registers, layouts and strides as in
// gcrnn_small_mfma.hip. The backward kernel recomputes c_t, accumulates the parameter gradients of its sequence in registers
// (tile-fragment layout) and emits per-sequence partial sums (added by the caller in a fixed order).
#include "gcrnn_common.h"
#include "gcrnn_small_mfma.h"

namespace {

template <typename T> __device__ __forceinline__ T sigmoid_t(T v) { return T(1) / (T(1) + exp(-v)); }
template <> __device__ __forceinline__ float sigmoid_t<float>(float v) { return 1.f / (1.f + expf(-v)); }

constexpr int GMAXT = 2;      // output tiles (F x N) per wave: tilesF * tilesN <= 32

// Shared prologue / per-step evaluation, used by both kernels.
template <typename T>
struct GateCtx {
  typedef typename Mf<T>::acc acc_t;
  int N, G, F, Kin, Kst, Ns, N4, G4, KG4, KGs, F4, F16, tilesN, tilesF;
  T *S, *Zx, *Zh0, *Zh1, *WA, *zrow, *red;
  int tid, lane, wave, li, lk;
};

// Sets up S, the tap matrix of the input filter and Yb (in tile-fragment registers); leaves h0's hop levels consumed.
template <typename T>
__device__ __forceinline__ void gate_prologue(GateCtx<T>& c, char* smem, const T* Sd, const T* h0b, const T* wA, const T* wB,
                                              const T* bias, typename Mf<T>::acc (&yb)[GMAXT]) {
  typedef typename Mf<T>::acc acc_t;
  const int N = c.N, G = c.G, F = c.F, Kin = c.Kin, Kst = c.Kst;
  c.Ns = lds_stride<T>(N); c.N4 = (N + 3) & ~3; c.G4 = (G + 3) & ~3; c.F4 = (F + 3) & ~3; c.F16 = (F + 15) & ~15;
  c.KG4 = Kin * c.G4; c.KGs = lds_stride<T>(c.KG4);
  c.tilesN = (N + 15) >> 4; c.tilesF = c.F16 >> 4;
  const int Ns = c.Ns;
  c.S = reinterpret_cast<T*>(smem);                 // [N4][Ns]
  c.Zx = c.S + (size_t)c.N4 * Ns;                   // [Kin][G4][Ns] hop levels of x_t
  c.Zh0 = c.Zx + (size_t)c.KG4 * Ns;                // [F4][Ns] hop levels of h0 (prologue / epilogue), ping
  c.Zh1 = c.Zh0 + (size_t)c.F4 * Ns;                // pong
  c.WA = c.Zh1 + (size_t)c.F4 * Ns;                 // [F16][KGs]: WA[f][k G4 + g] = wA[f][k][g]
  c.zrow = c.WA + (size_t)c.F16 * c.KGs;            // [Ns] zeros
  c.red = c.zrow + Ns;                              // [64]
  const int tid = c.tid;
  for (int i = tid; i < c.N4 * Ns; i += 1024) {
    const int m = i / Ns, n = i - m * Ns;
    c.S[i] = (m < N && n < N) ? Sd[(size_t)m * N + n] : T(0);
  }
  for (int i = tid; i < c.KG4 * Ns + 2 * c.F4 * Ns; i += 1024) c.Zx[i] = T(0);           // Zx, Zh0, Zh1 are contiguous
  for (int i = tid; i < c.F16 * c.KGs; i += 1024) {
    const int f = i / c.KGs, kg = i - f * c.KGs;
    const int k = kg / c.G4, g = kg - k * c.G4;
    c.WA[i] = (f < F && kg < c.KG4 && g < G) ? wA[((size_t)f * Kin + k) * G + g] : T(0);
  }
  for (int i = tid; i < Ns + 64; i += 1024) c.zrow[i] = T(0);
  __syncthreads();
  for (int i = tid; i < F * N; i += 1024) {
    const int f = i / N, n = i - f * N;
    c.Zh0[f * Ns + n] = h0b[i];
  }
  __syncthreads();
  // Yb = sum_k B_k (h0 S^k) + 2 b: taps straight from global memory (once per sequence), hops on the matrix cores
#pragma unroll
  for (int q = 0; q < GMAXT; ++q) yb[q] = acc_t{0, 0, 0, 0};
  T* zc = c.Zh0;
  T* zn = c.Zh1;
  for (int k = 0; k < Kst; ++k) {
#pragma unroll
    for (int q = 0; q < GMAXT; ++q) {
      const int tile = c.wave + q * 16;
      if (tile >= c.tilesF * c.tilesN) break;
      const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
      const int f = i0 + c.li;
      acc_t acc = yb[q];
      for (int s0 = 0; s0 < c.F4; s0 += 4) {
        const int f2 = s0 + c.lk;
        const T a = (f < F && f2 < F) ? wB[((size_t)f * Kst + k) * F + f2] : T(0);
        acc = Mf<T>::mma(a, zc[f2 * Ns + j0 + c.li], acc);
      }
      yb[q] = acc;
    }
    if (k + 1 < Kst) {
      for (int tile = c.wave; tile < c.tilesF * c.tilesN; tile += 16) {
        const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
        acc_t acc = {0, 0, 0, 0};
        const T* ap = (i0 + c.li < F) ? zc + (i0 + c.li) * Ns + c.lk : c.zrow + c.lk;
        acc = tile_mac<T>(acc, ap, 4, c.S + c.lk * Ns + j0 + c.li, 4 * Ns, c.N4 >> 2);
        const int n = j0 + c.li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int fr = i0 + Mf<T>::row(c.lane, r);
          if (fr < F && n < N) zn[fr * Ns + n] = acc[r];
        }
      }
      __syncthreads();
      T* tmp = zc; zc = zn; zn = tmp;
    }
  }
  if (bias) {
#pragma unroll
    for (int q = 0; q < GMAXT; ++q) {
      const int tile = c.wave + q * 16;
      if (tile >= c.tilesF * c.tilesN) break;
      const int i0 = (tile / c.tilesN) << 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = i0 + Mf<T>::row(c.lane, r);
        if (f < F) yb[q][r] += T(2) * bias[f];
      }
    }
  }
  __syncthreads();
}

// x_t -> hop levels Zx, then c = tanh(WA Zxflat + Yb) in tile fragments. Ends WITHOUT a barrier after the taps.
template <typename T>
__device__ __forceinline__ void gate_step(const GateCtx<T>& c, const T* xt, const typename Mf<T>::acc (&yb)[GMAXT],
                                          typename Mf<T>::acc (&cv)[GMAXT]) {
  typedef typename Mf<T>::acc acc_t;
  const int N = c.N, G = c.G, Ns = c.Ns;
  for (int i = c.tid; i < G * N; i += 1024) {
    const int g = i / N, n = i - g * N;
    c.Zx[g * Ns + n] = xt[i];
  }
  __syncthreads();
  for (int k = 1; k < c.Kin; ++k) {
    const T* zp = c.Zx + (size_t)(k - 1) * c.G4 * Ns;
    T* zn = c.Zx + (size_t)k * c.G4 * Ns;
    const int tilesG = (G + 15) >> 4;
    for (int tile = c.wave; tile < tilesG * c.tilesN; tile += 16) {
      const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
      acc_t acc = {0, 0, 0, 0};
      const T* ap = (i0 + c.li < G) ? zp + (i0 + c.li) * Ns + c.lk : c.zrow + c.lk;
      acc = tile_mac<T>(acc, ap, 4, c.S + c.lk * Ns + j0 + c.li, 4 * Ns, c.N4 >> 2);
      const int n = j0 + c.li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = i0 + Mf<T>::row(c.lane, r);
        if (g < G && n < N) zn[g * Ns + n] = acc[r];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < GMAXT; ++q) {
    const int tile = c.wave + q * 16;
    if (tile >= c.tilesF * c.tilesN) break;
    const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
    acc_t acc = yb[q];
    acc = tile_mac<T>(acc, c.WA + (i0 + c.li) * c.KGs + c.lk, 4, c.Zx + c.lk * Ns + j0 + c.li, 4 * Ns, c.KG4 >> 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Mf<T>::tanh_(acc[r]);
    cv[q] = acc;
  }
}

// block-wide sum of one value per thread; result valid in thread 0 (ends with a barrier)
template <typename T>
__device__ __forceinline__ T block_sum(const GateCtx<T>& c, T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if (c.lane == 0) c.red[c.wave] = v;
  __syncthreads();
  T s = T(0);
  if (c.tid == 0)
    for (int w = 0; w < 16; ++w) s += c.red[w];
  __syncthreads();
  return s;
}

template <typename T>
__global__ __launch_bounds__(1024) void small_gate_fwd_kernel(
    const T* __restrict__ X,        // [B][Tn][G][N]
    const T* __restrict__ h0,       // [B][F][N]
    const T* __restrict__ wA2,      // [2][F][Kin][G]
    const T* __restrict__ wB2,      // [2][F][Kst][F]
    const T* __restrict__ bias2,    // [2][F] or null
    const T* __restrict__ lw2,      // [2][F*N]
    const T* __restrict__ lb2,      // [2] or null
    const T* __restrict__ Sd,
    T* __restrict__ gate,           // [2][Tn][B]
    int Tn, int N, int G, int F, int Kin, int Kst, int B) {
  typedef typename Mf<T>::acc acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_gate[];
  GateCtx<T> c;
  c.N = N; c.G = G; c.F = F; c.Kin = Kin; c.Kst = Kst;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.li = c.lane & 15; c.lk = c.lane >> 4;
  const int b = blockIdx.x, g = blockIdx.y;
  acc_t yb[GMAXT], cv[GMAXT], lwv[GMAXT];
  gate_prologue<T>(c, smem_gate, Sd, h0 + (size_t)b * F * N, wA2 + (size_t)g * F * Kin * G, wB2 + (size_t)g * F * Kst * F,
                   bias2 ? bias2 + (size_t)g * F : nullptr, yb);
  const T* lw = lw2 + (size_t)g * F * N;
#pragma unroll
  for (int q = 0; q < GMAXT; ++q) {
    lwv[q] = acc_t{0, 0, 0, 0};
    const int tile = c.wave + q * 16;
    if (tile >= c.tilesF * c.tilesN) break;
    const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = i0 + Mf<T>::row(c.lane, r), n = j0 + c.li;
      if (f < F && n < N) lwv[q][r] = lw[(size_t)f * N + n];       // zero outside: masks the padded fragment entries
    }
  }
  const T lb = lb2 ? lb2[g] : T(0);
  for (int t = 0; t < Tn; ++t) {
    gate_step<T>(c, X + ((size_t)b * Tn + t) * G * N, yb, cv);
    T part = T(0);
#pragma unroll
    for (int q = 0; q < GMAXT; ++q) {
      const int tile = c.wave + q * 16;
      if (tile >= c.tilesF * c.tilesN) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) part += cv[q][r] * lwv[q][r];
    }
    const T s = block_sum<T>(c, part);
    if (c.tid == 0) gate[((size_t)g * Tn + t) * B + b] = sigmoid_t<T>(s + lb);
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void small_gate_bwd_kernel(
    const T* __restrict__ X, const T* __restrict__ h0, const T* __restrict__ wA2, const T* __restrict__ wB2,
    const T* __restrict__ bias2, const T* __restrict__ lw2, const T* __restrict__ Sd,
    const T* __restrict__ dsum,     // [2][Tn][B]: d loss / d (lw . c + lb)
    T* __restrict__ pA,             // [B][2][F][Kin][G]
    T* __restrict__ pB,             // [B][2][F][Kst][F]
    T* __restrict__ pb,             // [B][2][F]
    T* __restrict__ plw,            // [B][2][F*N]
    T* __restrict__ plb,            // [B][2]
    T* __restrict__ pdh0,           // [B][2][F][N] or null
    int Tn, int N, int G, int F, int Kin, int Kst, int B) {
  typedef typename Mf<T>::acc acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_gate[];
  GateCtx<T> c;
  c.N = N; c.G = G; c.F = F; c.Kin = Kin; c.Kst = Kst;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.li = c.lane & 15; c.lk = c.lane >> 4;
  const int b = blockIdx.x, g = blockIdx.y;
  const T* wA = wA2 + (size_t)g * F * Kin * G;
  const T* wB = wB2 + (size_t)g * F * Kst * F;
  acc_t yb[GMAXT], cv[GMAXT], lwv[GMAXT], dlw[GMAXT], dyb[GMAXT];
  gate_prologue<T>(c, smem_gate, Sd, h0 + (size_t)b * F * N, wA, wB, bias2 ? bias2 + (size_t)g * F : nullptr, yb);
  const int Ns = c.Ns;
  T* dP = c.red + 64;                               // [F4][Ns] d pre-activation of the current step (A operand of dA)
  T* dY = dP + (size_t)c.F4 * Ns;                   // [F4][Ns] sum over t of it (= d Yb), filled at the end
  for (int i = c.tid; i < 2 * c.F4 * Ns; i += 1024) dP[i] = T(0);
  const T* lw = lw2 + (size_t)g * F * N;
#pragma unroll
  for (int q = 0; q < GMAXT; ++q) {
    lwv[q] = dlw[q] = dyb[q] = acc_t{0, 0, 0, 0};
    const int tile = c.wave + q * 16;
    if (tile >= c.tilesF * c.tilesN) break;
    const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = i0 + Mf<T>::row(c.lane, r), n = j0 + c.li;
      if (f < F && n < N) lwv[q][r] = lw[(size_t)f * N + n];
    }
  }
  // dA tiles [F x Kin G4]: tile id = fi * tilesKG + kj, persistent over t on waves 0 .. tilesF * tilesKG - 1 (<= 16 checked)
  const int tilesKG = (c.KG4 + 15) >> 4;
  acc_t dA = {0, 0, 0, 0};
  T dlb = T(0);
  __syncthreads();
  for (int t = 0; t < Tn; ++t) {
    gate_step<T>(c, X + ((size_t)b * Tn + t) * G * N, yb, cv);
    const T ds = dsum[((size_t)g * Tn + t) * B + b];
    dlb += ds;
#pragma unroll
    for (int q = 0; q < GMAXT; ++q) {
      const int tile = c.wave + q * 16;
      if (tile >= c.tilesF * c.tilesN) break;
      const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T cc = cv[q][r];
        dlw[q][r] += ds * cc;
        const T dp = ds * lwv[q][r] * (T(1) - cc * cc);            // lwv is zero on padded entries
        dyb[q][r] += dp;
        const int f = i0 + Mf<T>::row(c.lane, r), n = j0 + c.li;
        if (f < F && n < N) dP[f * Ns + n] = dp;
      }
    }
    __syncthreads();
    if (c.wave < c.tilesF * tilesKG) {
      const int i0 = (c.wave / tilesKG) << 4, j0 = (c.wave % tilesKG) << 4;
      // A = dpre (i = f, k = n), B = Zxflat^T (k = n, j = k G4 + g)
      const T* ap = (i0 + c.li < F) ? dP + (i0 + c.li) * Ns + c.lk : c.zrow + c.lk;
      const T* bp = (j0 + c.li < c.KG4) ? c.Zx + (j0 + c.li) * Ns + c.lk : c.zrow + c.lk;
      dA = tile_mac<T>(dA, ap, 4, bp, 4, c.N4 >> 2);
    }
    __syncthreads();                 // Zx and dP are rewritten by the next step
  }
  // ---- per-sequence results
  if (c.tid == 0) plb[(size_t)b * 2 + g] = dlb;
#pragma unroll
  for (int q = 0; q < GMAXT; ++q) {
    const int tile = c.wave + q * 16;
    if (tile >= c.tilesF * c.tilesN) break;
    const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = i0 + Mf<T>::row(c.lane, r), n = j0 + c.li;
      if (f < F && n < N) {
        plw[((size_t)b * 2 + g) * F * N + (size_t)f * N + n] = dlw[q][r];
        dY[f * Ns + n] = dyb[q][r];
      }
    }
  }
  if (c.wave < c.tilesF * tilesKG) {
    const int i0 = (c.wave / tilesKG) << 4, j0 = (c.wave % tilesKG) << 4;
    const int kg = j0 + c.li, k = kg / c.G4, gg = kg - k * c.G4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = i0 + Mf<T>::row(c.lane, r);
      if (f < F && kg < c.KG4 && gg < G) pA[((((size_t)b * 2 + g) * F + f) * Kin + k) * G + gg] = dA[r];
    }
  }
  // h0 again for dB_k = dYb (h0 S^k)^T and d h0 = sum_k (B_k^T dYb)(S^T)^k
  for (int i = c.tid; i < F * N; i += 1024) {
    const int f = i / N, n = i - f * N;
    c.Zh0[f * Ns + n] = h0[(size_t)b * F * N + i];
  }
  __syncthreads();
  if (c.tid < F) {
    const T* dr = dY + c.tid * Ns;
    T s = T(0);
    for (int n = 0; n < N; ++n) s += dr[n];
    pb[((size_t)b * 2 + g) * F + c.tid] = T(2) * s;                // the bias enters both filters of the sub-cell
  }
  T* zc = c.Zh0;
  T* zn = c.Zh1;
  for (int k = 0; k < Kst; ++k) {
    for (int tile = c.wave; tile < c.tilesF * c.tilesF; tile += 16) {
      const int i0 = (tile / c.tilesF) << 4, j0 = (tile % c.tilesF) << 4;
      acc_t acc = {0, 0, 0, 0};
      const T* ap = (i0 + c.li < F) ? dY + (i0 + c.li) * Ns + c.lk : c.zrow + c.lk;
      const T* bp = (j0 + c.li < F) ? zc + (j0 + c.li) * Ns + c.lk : c.zrow + c.lk;
      acc = tile_mac<T>(acc, ap, 4, bp, 4, c.N4 >> 2);
      const int f2 = j0 + c.li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = i0 + Mf<T>::row(c.lane, r);
        if (f < F && f2 < F) pB[((((size_t)b * 2 + g) * F + f) * Kst + k) * F + f2] = acc[r];
      }
    }
    if (k + 1 < Kst) {
      for (int tile = c.wave; tile < c.tilesF * c.tilesN; tile += 16) {
        const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
        acc_t acc = {0, 0, 0, 0};
        const T* ap = (i0 + c.li < F) ? zc + (i0 + c.li) * Ns + c.lk : c.zrow + c.lk;
        acc = tile_mac<T>(acc, ap, 4, c.S + c.lk * Ns + j0 + c.li, 4 * Ns, c.N4 >> 2);
        const int n = j0 + c.li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int fr = i0 + Mf<T>::row(c.lane, r);
          if (fr < F && n < N) zn[fr * Ns + n] = acc[r];
        }
      }
      __syncthreads();
      T* tmp = zc; zc = zn; zn = tmp;
    }
  }
  if (pdh0) {
    __syncthreads();
    T* ac = c.Zh0;
    T* an = c.Zh1;
    for (int k = Kst - 1; k >= 0; --k) {
      for (int tile = c.wave; tile < c.tilesF * c.tilesN; tile += 16) {
        const int i0 = (tile / c.tilesN) << 4, j0 = (tile % c.tilesN) << 4;
        acc_t acc = {0, 0, 0, 0};
        const int f2 = i0 + c.li;
        for (int s0 = 0; s0 < c.F4; s0 += 4) {                       // A = B_k^T (i = f2, k = f) from global, B = dYb
          const int f = s0 + c.lk;
          const T a = (f < F && f2 < F) ? wB[((size_t)f * Kst + k) * F + f2] : T(0);
          acc = Mf<T>::mma(a, dY[f * Ns + j0 + c.li], acc);
        }
        if (k < Kst - 1) {
          const T* ap = (f2 < F) ? ac + f2 * Ns + c.lk : c.zrow + c.lk;
          acc = tile_mac<T>(acc, ap, 4, c.S + (j0 + c.li) * Ns + c.lk, 4, c.N4 >> 2);
        }
        const int n = j0 + c.li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int fr = i0 + Mf<T>::row(c.lane, r);
          if (fr < F && n < N) {
            if (k == 0) pdh0[((size_t)b * 2 + g) * F * N + (size_t)fr * N + n] = acc[r];
            else an[fr * Ns + n] = acc[r];
          }
        }
      }
      __syncthreads();
      T* tmp = ac; ac = an; an = tmp;
    }
  }
}

template <typename T>
size_t gate_lds(int64_t N, int64_t G, int64_t F, int64_t Kin, bool backward) {
  const int Ns = lds_stride<T>((int)N), N4 = ((int)N + 3) & ~3, G4 = ((int)G + 3) & ~3, F4 = ((int)F + 3) & ~3;
  const int F16 = ((int)F + 15) & ~15, KG4 = (int)Kin * G4, KGs = lds_stride<T>(KG4);
  size_t e = (size_t)N4 * Ns + (size_t)KG4 * Ns + 2 * (size_t)F4 * Ns + (size_t)F16 * KGs + Ns + 64;
  if (backward) e += 2 * (size_t)F4 * Ns;
  return sizeof(T) * e + 16;
}

template <typename T>
bool gate_supported(int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, bool backward) {
  if (N <= 0 || G <= 0 || F <= 0 || Kin <= 0 || Kst <= 0 || N > 256 || F > 256 || G > 64 || Kin > 8 || Kst > 8) return false;
  const int64_t tilesN = (N + 15) / 16, tilesF = (F + 15) / 16, tilesKG = (Kin * ((G + 3) / 4 * 4) + 15) / 16;
  if (tilesF * tilesN > 16 * GMAXT || tilesF * tilesKG > 16) return false;
  // dY rows are read as the B operand up to F4 - 1 and j0 + li < 16 tilesN: inside the allocation by construction
  return gate_lds<T>(N, G, F, Kin, backward) <= 160 * 1024;
}

}  // namespace

extern "C" int gcrnn_small_gates_supported(int dtype, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, int backward) {
  if (dtype == GCRNN_F32) return gate_supported<float>(N, G, F, Kin, Kst, backward != 0) ? 1 : 0;
  if (dtype == GCRNN_F64) return gate_supported<double>(N, G, F, Kin, Kst, backward != 0) ? 1 : 0;
  return 0;
}

template <typename T>
static int gates_fwd_launch(const void* X, const void* h0, const void* wA2, const void* wB2, const void* bias2, const void* lw2,
                            const void* lb2, const void* Sd, void* gate, int64_t B, int64_t Tn, int64_t N, int64_t G,
                            int64_t F, int64_t Kin, int64_t Kst, hipStream_t st) {
  const size_t lds = gate_lds<T>(N, G, F, Kin, false);
  auto kern = small_gate_fwd_kernel<T>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  kern<<<dim3((unsigned)B, 2), 1024, lds, st>>>((const T*)X, (const T*)h0, (const T*)wA2, (const T*)wB2, (const T*)bias2,
                                                (const T*)lw2, (const T*)lb2, (const T*)Sd, (T*)gate, (int)Tn, (int)N, (int)G,
                                                (int)F, (int)Kin, (int)Kst, (int)B);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_small_gates_forward(int dtype, const void* X, const void* h0, const void* wA2, const void* wB2,
                                         const void* bias2, const void* lw2, const void* lb2, const void* Sdense, void* gate,
                                         int64_t B, int64_t T, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst,
                                         void* stream) {
  if (!X || !h0 || !wA2 || !wB2 || !lw2 || !Sdense || !gate) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || B > 65535 * 32768LL) return GCRNN_ERR_BAD_SHAPE;
  if (!gcrnn_small_gates_supported(dtype, N, G, F, Kin, Kst, 0)) return GCRNN_ERR_UNSUPPORTED;
  if (dtype == GCRNN_F32)
    return gates_fwd_launch<float>(X, h0, wA2, wB2, bias2, lw2, lb2, Sdense, gate, B, T, N, G, F, Kin, Kst, as_stream(stream));
  return gates_fwd_launch<double>(X, h0, wA2, wB2, bias2, lw2, lb2, Sdense, gate, B, T, N, G, F, Kin, Kst, as_stream(stream));
}

template <typename T>
static int gates_bwd_launch(const void* X, const void* h0, const void* wA2, const void* wB2, const void* bias2, const void* lw2,
                            const void* Sd, const void* dsum, void* pA, void* pB, void* pb, void* plw, void* plb, void* pdh0,
                            int64_t B, int64_t Tn, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, hipStream_t st) {
  const size_t lds = gate_lds<T>(N, G, F, Kin, true);
  auto kern = small_gate_bwd_kernel<T>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  kern<<<dim3((unsigned)B, 2), 1024, lds, st>>>((const T*)X, (const T*)h0, (const T*)wA2, (const T*)wB2, (const T*)bias2,
                                                (const T*)lw2, (const T*)Sd, (const T*)dsum, (T*)pA, (T*)pB, (T*)pb, (T*)plw,
                                                (T*)plb, (T*)pdh0, (int)Tn, (int)N, (int)G, (int)F, (int)Kin, (int)Kst, (int)B);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_small_gates_backward(int dtype, const void* X, const void* h0, const void* wA2, const void* wB2,
                                          const void* bias2, const void* lw2, const void* Sdense, const void* dsum, void* pA,
                                          void* pB, void* pb, void* plw, void* plb, void* pdh0, int64_t B, int64_t T, int64_t N,
                                          int64_t G, int64_t F, int64_t Kin, int64_t Kst, void* stream) {
  if (!X || !h0 || !wA2 || !wB2 || !lw2 || !Sdense || !dsum || !pA || !pB || !pb || !plw || !plb) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || B > 65535 * 32768LL) return GCRNN_ERR_BAD_SHAPE;
  if (!gcrnn_small_gates_supported(dtype, N, G, F, Kin, Kst, 1)) return GCRNN_ERR_UNSUPPORTED;
  if (dtype == GCRNN_F32)
    return gates_bwd_launch<float>(X, h0, wA2, wB2, bias2, lw2, Sdense, dsum, pA, pB, pb, plw, plb, pdh0, B, T, N, G, F, Kin, Kst,
                                   as_stream(stream));
  return gates_bwd_launch<double>(X, h0, wA2, wB2, bias2, lw2, Sdense, dsum, pA, pB, pb, plw, plb, pdh0, B, T, N, G, F, Kin, Kst,
                                  as_stream(stream));
}
