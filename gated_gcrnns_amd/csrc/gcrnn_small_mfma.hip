// Small-graph regime on the fp64 / fp32 matrix cores (BASELINE configs[0], [3]; the drivers' N = 50..80).
//
// With N <= ~128 nodes the GSO fits in LDS as a DENSE N x N matrix, and every piece of a time step is a small GEMM:
//   hop      Z_k   [C x N]  = Z_{k-1} [C x N] * S [N x N]                     (reference x @ S, graphML.py:123)
//   taps     pre   [F x N]  = W [F x K C] * Zflat [K C x N]                   (graphML.py:134-135)
//   dW       [F x C] per tap += dpre [F x N] * Z_k^T [N x C]
//   adjoint  carry [F x N]  = sum_k (B_k^T dpre) (S^T)^k   (Horner: acc <- acc S^T + B_k^T dpre)
// so the whole T-step recurrence (and its BPTT) of one sequence runs in ONE workgroup, ONE launch, on
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 with operands read straight from LDS. The gather-based kernels of
// gcrnn_small.hip spend 12.5 us (forward) / 21 us (backward) per time step on LDS instruction issue; here a step is a few
// hundred MFMAs. Everything stays channel-major [c][n] = the user layout, so x_t / H rows move as contiguous segments.
//
// LDS row stride: every matrix uses a stride whose byte size is an ODD multiple of 16 (mod 256). Then the pattern
// "16 lanes over 16 rows, 4 lane groups over 4 consecutive elements" (A operands, transposed B operands) is conflict-free
// and the pattern "16 lanes over 16 consecutive elements, 4 lane groups over 4 rows" costs 2x the minimum.
//
// Register layouts (probed on gfx950, tools/probes/mfma_layout_probe.hip): A[i][k] in lane i + 16 k, B[k][j] in lane
// j + 16 k; D[i][j]: column j = lane % 16, row i = 4 (lane / 16) + r for fp32 but (lane / 16) + 4 r for fp64.
#include "gcrnn_common.h"
#include "gcrnn_small_mfma.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int MAXT>
__global__ __launch_bounds__(1024) void small_dense_fwd_kernel(
    const T* __restrict__ X,       // [B][Tn][G][N]
    const T* __restrict__ h0,      // [B][F][N]
    const T* __restrict__ wA,      // [F][Kin][G]
    const T* __restrict__ wB,      // [F][Kst][F]
    const T* __restrict__ bias,    // [F] or null
    const T* __restrict__ gi, const T* __restrict__ gf,       // gates of the input / state filter per (b, t, n) through strides, or null
    const T* __restrict__ Sd,      // [N][N] dense S (row m, column n)
    T* __restrict__ H,             // [B][Tn][F][N]
    int Tn, int N, int G, int F, int Kin, int Kst, int B,
    int64_t gsb, int64_t gst, int64_t gsn) {     // gate element (b, t, n) sits at b gsb + t gst + n gsn
  typedef typename Mf<T>::acc acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_dense[];
  const int K = Kin > Kst ? Kin : Kst;
  const int C = G + F, KC = K * C;
  const int Ns = lds_stride<T>(N), KCs = lds_stride<T>(KC);
  const int N4 = (N + 3) & ~3, KC4 = (KC + 3) & ~3;
  T* S = reinterpret_cast<T*>(smem_dense);         // [N4][Ns]
  T* Z = S + (size_t)N4 * Ns;                      // [KC4][Ns]  (level k = rows k C .. k C + C - 1)
  T* W = Z + (size_t)KC4 * Ns;                     // [F16][KCs] combined taps, F16 = F rounded up to 16
  const int F16 = (F + 15) & ~15;
  T* zrow = W + (size_t)F16 * KCs;                 // [Ns] zeros: what masked lanes read
  T* gvec = zrow + Ns;                             // [2][Ns] gates of this step: input-filter | state-filter, per node
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.x;

  for (int i = tid; i < N4 * Ns; i += 1024) {
    const int m = i / Ns, n = i - m * Ns;
    S[i] = (m < N && n < N) ? Sd[(size_t)m * N + n] : T(0);
  }
  for (int i = tid; i < KC4 * Ns; i += 1024) Z[i] = T(0);
  for (int i = tid; i < 3 * Ns; i += 1024) zrow[i] = T(0);
  for (int i = tid; i < F16 * KCs; i += 1024) {
    const int f = i / KCs, kc = i - f * KCs;
    T v = T(0);
    if (f < F && kc < KC) {
      const int k = kc / C, c = kc - k * C;
      if (c < G) { if (k < Kin) v = wA[((size_t)f * Kin + k) * G + c]; }
      else       { if (k < Kst) v = wB[((size_t)f * Kst + k) * F + (c - G)]; }
    }
    W[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < F * N; i += 1024) {
    const int f = i / N, n = i - f * N;
    Z[(G + f) * Ns + n] = h0[(size_t)b * F * N + i];
  }

  const int tilesN = (N + 15) >> 4, tilesC = (C + 15) >> 4, tilesF = F16 >> 4;
  const T* xb = X + (size_t)b * Tn * G * N;
  const int GN = G * N;
  // x_t is fetched one step ahead (the first 1024 values in a register) so that its global-memory latency overlaps
  // the previous step instead of sitting in front of a barrier
  const int xg = tid / N, xn = tid - xg * N;
  T xpre = (tid < GN) ? xb[tid] : T(0);
  for (int t = 0; t < Tn; ++t) {
    const T* xt = xb + (size_t)t * GN;
    if (tid < GN) Z[xg * Ns + xn] = xpre;
    for (int i = tid + 1024; i < GN; i += 1024) {
      const int g = i / N, n = i - g * N;
      Z[g * Ns + n] = xt[i];
    }
    if (t + 1 < Tn && tid < GN) xpre = xt[GN + tid];
    if (gi && tid < 2 * N) {
      const int w = tid >= N, n = tid - w * N;
      gvec[w * Ns + n] = (w ? gf : gi)[b * gsb + t * gst + n * gsn];
    }
    __syncthreads();
    // ---- hops: Z_k = Z_{k-1} S
    for (int k = 1; k < K; ++k) {
      const T* zp = Z + (size_t)(k - 1) * C * Ns;
      T* zn = Z + (size_t)k * C * Ns;
      for (int tile = wave; tile < tilesC * tilesN; tile += 16) {
        const int i0 = (tile / tilesN) << 4, j0 = (tile % tilesN) << 4;
        acc_t acc = {0, 0, 0, 0};
        const T* ap = (i0 + li < C) ? zp + (i0 + li) * Ns + lk : zrow + lk;
        acc = tile_mac<T>(acc, ap, 4, S + lk * Ns + j0 + li, 4 * Ns, N4 >> 2);
        const int n = j0 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = i0 + Mf<T>::row(lane, r);
          if (c < C && n < N) zn[c * Ns + n] = acc[r];
        }
      }
      __syncthreads();
    }
    // ---- taps: pre = W Zflat (time gates scale the x- and h-columns of W)
    acc_t outv[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
      const int tile = wave + q * 16;
      if (tile >= tilesF * tilesN) break;
      const int i0 = (tile / tilesN) << 4, j0 = (tile % tilesN) << 4;
      acc_t acc = {0, 0, 0, 0};
      const T* wr = W + (i0 + li) * KCs + lk;
      acc = tile_mac<T>(acc, wr, 4, Z + lk * Ns + j0 + li, 4 * Ns, KC4 >> 2);          // A(S)x + B(S)h
      T gxn = T(1), ghn = T(1);
      acc_t ax = {0, 0, 0, 0};
      if (gi) {
        // gates multiply the two filters separately: the input filter's share is the (few) x-columns k C + g again
        const int n = j0 + li;
        gxn = gvec[n < N ? n : 0]; ghn = gvec[Ns + (n < N ? n : 0)];
        for (int k = 0; k < Kin; ++k)
          for (int g0 = 0; g0 < G; g0 += 4) {
            const int g = g0 + lk, gc = g < G ? g : G - 1;
            const T a = (W + (i0 + li) * KCs)[k * C + gc] * (g < G ? T(1) : T(0));
            ax = Mf<T>::mma(a, Z[(k * C + gc) * Ns + j0 + li], ax);
          }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = i0 + Mf<T>::row(lane, r);
        const T bb = (bias && f < F) ? bias[f] : T(0);
        const T pre = gi ? gxn * (ax[r] + bb) + ghn * ((acc[r] - ax[r]) + bb) : acc[r] + T(2) * bb;
        acc[r] = Mf<T>::tanh_(pre);
      }
      outv[q] = acc;
    }
    __syncthreads();                    // every wave is done reading Z_0 (the old state) before it is replaced
    T* Hout = H + ((size_t)b * Tn + t) * F * N;
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
      const int tile = wave + q * 16;
      if (tile >= tilesF * tilesN) break;
      const int i0 = (tile / tilesN) << 4, j0 = (tile % tilesN) << 4;
      const int n = j0 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = i0 + Mf<T>::row(lane, r);
        if (f < F && n < N) {
          const T v = outv[q][r];
          Z[(G + f) * Ns + n] = v;
          Hout[(size_t)f * N + n] = v;
        }
      }
    }
    // the x_t copy of the next step touches rows 0..G-1 only; its barrier orders the new state too
  }
}

template <typename T>
size_t dense_fwd_lds(int64_t N, int64_t G, int64_t F, int64_t K) {
  const int C = (int)(G + F), KC = (int)K * C;
  const int Ns = lds_stride<T>((int)N), KCs = lds_stride<T>(KC);
  const int N4 = ((int)N + 3) & ~3, KC4 = (KC + 3) & ~3, F16 = ((int)F + 15) & ~15;
  return sizeof(T) * ((size_t)N4 * Ns + (size_t)KC4 * Ns + (size_t)F16 * KCs + 3 * (size_t)Ns) + 16;
}

// ------------------------------------------------------------------------------------------------------------------
// backward (BPTT); same contract as small_cell_bwd_kernel in gcrnn_small.hip
// ------------------------------------------------------------------------------------------------------------------
template <typename T, bool GATED, int MAXW>
__global__ __launch_bounds__(1024) void small_dense_bwd_kernel(
    const T* __restrict__ X, const T* __restrict__ h0, const T* __restrict__ H, const T* __restrict__ dH,
    const T* __restrict__ wA, const T* __restrict__ wB, const T* __restrict__ bias,
    const T* __restrict__ gi, const T* __restrict__ gf,       // gates per (b, t, n) through strides (GATED)
    const T* __restrict__ Sd,
    T* __restrict__ pA,             // [B][F][Kin][G]
    T* __restrict__ pB,             // [B][F][Kst][F]
    T* __restrict__ pb,             // [B][F]
    T* __restrict__ dgi, T* __restrict__ dgf,                 // [B][Tn][N] gradients of the gates (GATED)
    T* __restrict__ dh0,
    int Tn, int N, int G, int F, int Kin, int Kst, int B,
    int64_t gsb, int64_t gst, int64_t gsn) {     // gate element (b, t, n) sits at b gsb + t gst + n gsn (dgi / dgf are dense)
  typedef typename Mf<T>::acc acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_dense[];
  const int K = Kin > Kst ? Kin : Kst;
  const int C = G + F;
  const int Ns = lds_stride<T>(N), Fs = lds_stride<T>(F + 1), Cs = lds_stride<T>(C);      // Fs > F: column F of WBt is zero
  const int N4 = (N + 3) & ~3, F4 = (F + 3) & ~3, C4 = (C + 3) & ~3, C16 = (C + 15) & ~15, F16 = (F + 15) & ~15;
  T* S = reinterpret_cast<T*>(smem_dense);         // [max(N4, 16 tilesN)][Ns]   dense S; rows also serve as B operand S[n][m]
  const int tilesN = (N + 15) >> 4, tilesC = C16 >> 4, tilesF = F16 >> 4;
  const int SR = tilesN * 16;                      // rows allocated for S (rows >= N are zero)
  T* Z0 = S + (size_t)SR * Ns;                     // [C4][Ns]  ping
  T* Z1 = Z0 + (size_t)C4 * Ns;                    // [C4][Ns]  pong   (both also hold the adjoint accumulators [F][Ns])
  T* dpre = Z1 + (size_t)C4 * Ns;                  // [F4][Ns]  (rows >= F stay zero: k-padding of the B operands)
  T* carry = dpre + (size_t)F4 * Ns;               // [F4][Ns]
  T* WBt = carry + (size_t)F4 * Ns;                // [Kst][F4 (f)][Fs (f2)]   w_B[f][k][f2]
  T* WAl = WBt + (size_t)Kst * F4 * Fs;            // GATED: [F][K][Cs] combined taps
  T* red = WAl + (GATED ? (size_t)F * K * Cs : 0);     // [64]
  T* zrow = red + 64;                                  // [max(Ns, K Cs)] zeros: what masked lanes read
  const int ZR = Ns > K * Cs ? Ns : K * Cs;
  T* gvec = zrow + ZR;                                 // GATED: [2][Ns] gates of this step (input | state filter) per node
  T* gpart = gvec + 2 * Ns;                            // GATED: [2][tilesF][Ns] partial column sums of the gate gradients
  (void)C16; (void)F16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.x;

  for (int i = tid; i < SR * Ns; i += 1024) {
    const int m = i / Ns, n = i - m * Ns;
    S[i] = (m < N && n < N) ? Sd[(size_t)m * N + n] : T(0);
  }
  for (int i = tid; i < 2 * C4 * Ns + 2 * F4 * Ns; i += 1024) Z0[i] = T(0);        // Z0, Z1, dpre, carry are contiguous
  for (int i = tid; i < ZR + (GATED ? 2 * Ns + 2 * tilesF * Ns : 0); i += 1024) zrow[i] = T(0);
  for (int i = tid; i < Kst * F4 * Fs; i += 1024) {
    const int k = i / (F4 * Fs), rem = i - k * (F4 * Fs);
    const int f = rem / Fs, f2 = rem - f * Fs;
    WBt[i] = (f < F && f2 < F) ? wB[((size_t)f * Kst + k) * F + f2] : T(0);
  }
  if (GATED)
    for (int i = tid; i < F * K * Cs; i += 1024) {
      const int f = i / (K * Cs), rem = i - f * (K * Cs);
      const int k = rem / Cs, c = rem - k * Cs;
      T v = T(0);
      if (c < C) {
        if (c < G) { if (k < Kin) v = wA[((size_t)f * Kin + k) * G + c]; }
        else       { if (k < Kst) v = wB[((size_t)f * Kst + k) * F + (c - G)]; }
      }
      WAl[i] = v;
    }
  __syncthreads();

  // weight-gradient accumulator tiles, persistent over t: tile id = (k, fi, cj) -> dW[f][k][c]; wave w owns w, w + 16, ...
  const int wtiles = K * tilesF * tilesC;
  acc_t wacc[MAXW];
#pragma unroll
  for (int q = 0; q < MAXW; ++q) wacc[q] = acc_t{0, 0, 0, 0};
  T bacc = T(0);
  const int FN = F * N, GN = G * N;

  for (int t = Tn - 1; t >= 0; --t) {
    const T* xt = X + ((size_t)b * Tn + t) * GN;
    const T* hp = (t == 0) ? h0 + (size_t)b * FN : H + ((size_t)b * Tn + t - 1) * FN;
    const T* ht = H + ((size_t)b * Tn + t) * FN;
    const T* dht = dH + ((size_t)b * Tn + t) * FN;
    if (GATED && tid < 2 * N) {
      const int w = tid >= N, n = tid - w * N;
      gvec[w * Ns + n] = (w ? gf : gi)[b * gsb + t * gst + n * gsn];
    }
    for (int i = tid; i < (G + F) * N; i += 1024) {
      const int c = i / N, n = i - c * N;
      Z0[c * Ns + n] = (c < G) ? xt[i] : hp[i - GN];
    }
    for (int i = tid; i < FN; i += 1024) {
      const int f = i / N, n = i - f * N;
      const T h = ht[i];
      dpre[f * Ns + n] = (dht[i] + carry[f * Ns + n]) * (T(1) - h * h);
    }
    __syncthreads();
    // ---- levels k = 0 .. K-1: dW_k += dpre Z_k^T  |  (gated) ya / yb += W_k Z_k  |  Z_{k+1} = Z_k S
    acc_t ya[2], yb[2];
    if (GATED) { ya[0] = ya[1] = yb[0] = yb[1] = acc_t{0, 0, 0, 0}; }
    T* zc = Z0;
    T* zn = Z1;
    for (int k = 0; k < K; ++k) {
      // weight gradients of this level
#pragma unroll
      for (int q = 0; q < MAXW; ++q) {
        const int tile = wave + q * 16;
        if (tile < wtiles && tile / (tilesF * tilesC) == k) {
          const int rem = tile - k * (tilesF * tilesC);
          const int i0 = (rem / tilesC) << 4, j0 = (rem % tilesC) << 4;
          // A = dpre (i = f, k = n), B = Z_k^T (k = n, j = c): both "lanes over rows, lane groups over consecutive n"
          const T* ap = (i0 + li < F) ? dpre + (i0 + li) * Ns + lk : zrow + lk;
          const T* bp = (j0 + li < C) ? zc + (j0 + li) * Ns + lk : zrow + lk;
          // A = dpre (i = f, k = n), B = Z_k^T (k = n, j = c) scaled by the gate of (filter of column c, node n)
          if (GATED) wacc[q] = tile_mac_gated<T>(wacc[q], bp, gvec + ((j0 + li < G) ? 0 : Ns) + lk, ap, 4, N4 >> 2, true);
          else       wacc[q] = tile_mac<T>(wacc[q], ap, 4, bp, 4, N4 >> 2);
        }
      }
      if (GATED) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int tile = wave + q * 16;
          if (tile >= tilesF * tilesN) break;
          const int i0 = (tile / tilesN) << 4, j0 = (tile % tilesN) << 4;
          const T* wr = (i0 + li < F) ? WAl + ((size_t)(i0 + li) * K + k) * Cs : zrow;      // columns C .. Cs-1 are zero
          acc_t a = ya[q], bq = yb[q];
          const int G4 = (G + 3) & ~3;
          // x-columns c < G into ya, h-columns into yb (a k-step that straddles G contributes to both, masked by select)
          for (int c0 = 0; c0 < C4; c0 += 4) {
            const int c = c0 + lk;
            const T wv = wr[c];
            const T zv = zc[c * Ns + j0 + li];                       // rows C .. C4-1 are zero
            if (c0 < G4) a = Mf<T>::mma(c < G ? wv : T(0), zv, a);
            if (c0 + 3 >= G) bq = Mf<T>::mma(c >= G ? wv : T(0), zv, bq);
          }
          ya[q] = a; yb[q] = bq;
        }
      }
      if (k + 1 < K) {
        for (int tile = wave; tile < tilesC * tilesN; tile += 16) {
          const int tl = (tile + 5) % (tilesC * tilesN);          // start the hop tiles on other waves than the dW tiles
          const int i0 = (tl / tilesN) << 4, j0 = (tl % tilesN) << 4;
          acc_t acc = {0, 0, 0, 0};
          const T* ap = (i0 + li < C) ? zc + (i0 + li) * Ns + lk : zrow + lk;
          acc = tile_mac<T>(acc, ap, 4, S + lk * Ns + j0 + li, 4 * Ns, N4 >> 2);
          const int n = j0 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = i0 + Mf<T>::row(lane, r);
            if (c < C && n < N) zn[c * Ns + n] = acc[r];
          }
        }
        __syncthreads();
        T* tmp = zc; zc = zn; zn = tmp;
      }
    }
    __syncthreads();                    // the Z buffers are free: they become the adjoint accumulators
    // ---- bias / gate partial sums
    if (tid < F) {
      const T* dr = dpre + tid * Ns;
      T s = T(0);
      if (GATED) { for (int n = 0; n < N; ++n) s += dr[n] * (gvec[n] + gvec[Ns + n]); }
      else       { for (int n = 0; n < N; ++n) s += dr[n]; s *= T(2); }
      bacc += s;
    }
    if (GATED) {
      // d gate[n] = sum_f dpre[f][n] (filter output[f][n] + b[f]): each lane holds 4 rows of its column; the 4 lanes of a
      // column (lane, lane ^ 16, ^ 32, ^ 48) are folded by shuffles, the tilesF row tiles through LDS in a fixed order
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int tile = wave + q * 16;
        if (tile >= tilesF * tilesN) break;
        const int i0 = (tile / tilesN) << 4, j0 = (tile % tilesN) << 4;
        const int n = j0 + li;
        T si = T(0), sf = T(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = i0 + Mf<T>::row(lane, r);
          if (f < F && n < N) {
            const T bb = bias ? bias[f] : T(0);
            const T d = dpre[f * Ns + n];
            si += d * (ya[q][r] + bb);
            sf += d * (yb[q][r] + bb);
          }
        }
        si += __shfl_xor(si, 16, 64); si += __shfl_xor(si, 32, 64);
        sf += __shfl_xor(sf, 16, 64); sf += __shfl_xor(sf, 32, 64);
        if (n < N && lk == 0) {
          gpart[(i0 >> 4) * Ns + n] = si;
          gpart[(tilesF + (i0 >> 4)) * Ns + n] = sf;
        }
      }
      __syncthreads();
      if (tid < 2 * N) {
        const int w = tid >= N, n = tid - w * N;
        T a = T(0);
        for (int p = 0; p < tilesF; ++p) a += gpart[(w * tilesF + p) * Ns + n];
        (w ? dgf : dgi)[((size_t)b * Tn + t) * N + n] = a;
      }
    }
    // ---- carry_{t-1}: acc <- gf B_k^T dpre + acc S^T, k = Kst-1 .. 0
    T* ac = Z0;
    T* an = Z1;
    for (int k = Kst - 1; k >= 0; --k) {
      T* dst = (k == 0) ? carry : an;
      const T* wk = WBt + (size_t)k * F4 * Fs;
      for (int tile = wave; tile < tilesF * tilesN; tile += 16) {
        const int i0 = (tile / tilesN) << 4, j0 = (tile % tilesN) << 4;
        acc_t acc = {0, 0, 0, 0};
        // A = B_k^T (i = f2, k = f) = wk[f][f2], B = dpre (k = f, j = n)
        const bool acol = i0 + li < F;
        const int nb = j0 + li;
        acc = tile_mac<T>(acc, wk + lk * Fs + (acol ? i0 + li : F), 4 * Fs, dpre + lk * Ns + j0 + li, 4 * Ns, F4 >> 2, T(1),
                          GATED ? gvec[Ns + (nb < N ? nb : 0)] : T(1));
        if (k < Kst - 1) {
          // A = acc (i = f2, k = m), B = S^T (k = m, j = n) = S[n][m]
          const T* ap = acol ? ac + (i0 + li) * Ns + lk : zrow + lk;
          acc = tile_mac<T>(acc, ap, 4, S + (j0 + li) * Ns + lk, 4, N4 >> 2);
        }
        const int n = j0 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f2 = i0 + Mf<T>::row(lane, r);
          if (f2 < F && n < N) dst[f2 * Ns + n] = acc[r];
        }
      }
      __syncthreads();
      T* tmp = ac; ac = an; an = tmp;
    }
  }

  // ---- results
  if (dh0)
    for (int i = tid; i < FN; i += 1024) {
      const int f = i / N, n = i - f * N;
      dh0[(size_t)b * FN + i] = carry[f * Ns + n];
    }
  if (tid < F) pb[(size_t)b * F + tid] = bacc;
#pragma unroll
  for (int q = 0; q < MAXW; ++q) {
    const int tile = wave + q * 16;
    if (tile < wtiles) {
      const int k = tile / (tilesF * tilesC), rem = tile - k * (tilesF * tilesC);
      const int i0 = (rem / tilesC) << 4, j0 = (rem % tilesC) << 4;
      const int c = j0 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = i0 + Mf<T>::row(lane, r);
        if (f < F && c < C) {
          if (c < G) { if (k < Kin) pA[(((size_t)b * F + f) * Kin + k) * G + c] = wacc[q][r]; }
          else       { if (k < Kst) pB[(((size_t)b * F + f) * Kst + k) * F + (c - G)] = wacc[q][r]; }
        }
      }
    }
  }
}

template <typename T>
size_t dense_bwd_lds(int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, bool gated) {
  const int K = (int)(Kin > Kst ? Kin : Kst), C = (int)(G + F);
  const int Ns = lds_stride<T>((int)N), Fs = lds_stride<T>((int)F + 1), Cs = lds_stride<T>(C);
  const int F4 = ((int)F + 3) & ~3, C4 = (C + 3) & ~3;
  const int SR = (((int)N + 15) >> 4) * 16;
  return sizeof(T) * ((size_t)SR * Ns + 2 * (size_t)C4 * Ns + 2 * (size_t)F4 * Ns + (size_t)Kst * F4 * Fs +
                      (gated ? (size_t)F * K * Cs : 0) + 64 + (size_t)(Ns > K * Cs ? Ns : K * Cs) +
                      (gated ? (2 + 2 * (size_t)(((int)F + 15) >> 4)) * Ns : 0)) + 16;
}

constexpr size_t DENSE_LDS_MAX = 160 * 1024;

template <typename T>
bool dense_shape_ok(int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst) {
  if (N <= 0 || G <= 0 || F <= 0 || Kin <= 0 || Kst <= 0 || N > 256 || F > 256 || G > 256 || Kin > 8 || Kst > 8) return false;
  return true;
}

}  // namespace

extern "C" int gcrnn_small_dense_supported(int dtype, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst,
                                           int backward, int gated) {
  if (dtype != GCRNN_F32 && dtype != GCRNN_F64) return 0;
  const int64_t K = Kin > Kst ? Kin : Kst;
  const int64_t tilesN = (N + 15) / 16, tilesF = (F + 15) / 16, tilesC = (G + F + 15) / 16;
  if (dtype == GCRNN_F32) {
    if (!dense_shape_ok<float>(N, G, F, Kin, Kst)) return 0;
    if (!backward) return tilesF * tilesN <= 16 * 4 && dense_fwd_lds<float>(N, G, F, K) <= DENSE_LDS_MAX;
    return K * tilesF * tilesC <= 16 * 4 && (!gated || tilesF * tilesN <= 32) &&
           dense_bwd_lds<float>(N, G, F, Kin, Kst, gated != 0) <= DENSE_LDS_MAX;
  }
  if (!dense_shape_ok<double>(N, G, F, Kin, Kst)) return 0;
  if (!backward) return tilesF * tilesN <= 16 * 4 && dense_fwd_lds<double>(N, G, F, K) <= DENSE_LDS_MAX;
  return K * tilesF * tilesC <= 16 * 4 && (!gated || tilesF * tilesN <= 32) &&
         dense_bwd_lds<double>(N, G, F, Kin, Kst, gated != 0) <= DENSE_LDS_MAX;
}

template <typename T>
static int dense_fwd_launch(const void* X, const void* h0, const void* wA, const void* wB, const void* bias, const void* gi,
                            const void* gf, const void* Sd, void* H, int64_t B, int64_t Tn, int64_t N, int64_t G, int64_t F,
                            int64_t Kin, int64_t Kst, int64_t gsb, int64_t gst, int64_t gsn, hipStream_t st) {
  const int64_t K = Kin > Kst ? Kin : Kst;
  const size_t lds = dense_fwd_lds<T>(N, G, F, K);
  const int64_t ot = ((F + 15) / 16) * ((N + 15) / 16);             // output tiles per step, dealt over 16 waves
  auto kern = ot <= 16 ? small_dense_fwd_kernel<T, 1> : (ot <= 32 ? small_dense_fwd_kernel<T, 2> : small_dense_fwd_kernel<T, 4>);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)B, 1024, lds, st>>>((const T*)X, (const T*)h0, (const T*)wA, (const T*)wB, (const T*)bias, (const T*)gi,
                                      (const T*)gf, (const T*)Sd, (T*)H, (int)Tn, (int)N, (int)G, (int)F, (int)Kin, (int)Kst,
                                      (int)B, gsb, gst, gsn);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_small_dense_forward(int dtype, const void* X, const void* h0, const void* wA, const void* wB,
                                         const void* bias, const void* gi, const void* gf, const void* Sdense, void* H,
                                         int64_t B, int64_t T, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst,
                                         int64_t gate_stride_b, int64_t gate_stride_t, int64_t gate_stride_n, void* stream) {
  if (!X || !h0 || !wA || !wB || !Sdense || !H) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || B > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (!gcrnn_small_dense_supported(dtype, N, G, F, Kin, Kst, 0, gi != nullptr)) return GCRNN_ERR_UNSUPPORTED;
  if (dtype == GCRNN_F32)
    return dense_fwd_launch<float>(X, h0, wA, wB, bias, gi, gf, Sdense, H, B, T, N, G, F, Kin, Kst, gate_stride_b,
                                   gate_stride_t, gate_stride_n, as_stream(stream));
  return dense_fwd_launch<double>(X, h0, wA, wB, bias, gi, gf, Sdense, H, B, T, N, G, F, Kin, Kst, gate_stride_b, gate_stride_t,
                                  gate_stride_n, as_stream(stream));
}

template <typename T, bool GATED>
static int dense_bwd_launch(const void* X, const void* h0, const void* H, const void* dH, const void* wA, const void* wB,
                            const void* bias, const void* gi, const void* gf, const void* Sd, void* pA, void* pB, void* pb,
                            void* dgi, void* dgf, void* dh0, int64_t B, int64_t Tn, int64_t N, int64_t G, int64_t F,
                            int64_t Kin, int64_t Kst, int64_t gsb, int64_t gst, int64_t gsn, hipStream_t st) {
  const size_t lds = dense_bwd_lds<T>(N, G, F, Kin, Kst, GATED);
  const int64_t wt = (Kin > Kst ? Kin : Kst) * ((F + 15) / 16) * ((G + F + 15) / 16);   // weight-gradient tiles, persistent
  auto kern = wt <= 16 ? small_dense_bwd_kernel<T, GATED, 1>
                       : (wt <= 32 ? small_dense_bwd_kernel<T, GATED, 2> : small_dense_bwd_kernel<T, GATED, 4>);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)B, 1024, lds, st>>>((const T*)X, (const T*)h0, (const T*)H, (const T*)dH, (const T*)wA, (const T*)wB,
                                      (const T*)bias, (const T*)gi, (const T*)gf, (const T*)Sd, (T*)pA, (T*)pB, (T*)pb,
                                      (T*)dgi, (T*)dgf, (T*)dh0, (int)Tn, (int)N, (int)G, (int)F, (int)Kin, (int)Kst, (int)B, gsb,
                                      gst, gsn);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_small_dense_backward(int dtype, const void* X, const void* h0, const void* H, const void* dH,
                                          const void* wA, const void* wB, const void* bias, const void* gi, const void* gf,
                                          const void* Sdense, void* pA, void* pB, void* pb, void* dgi, void* dgf, void* dh0,
                                          int64_t B, int64_t T, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst,
                                          int64_t gate_stride_b, int64_t gate_stride_t, int64_t gate_stride_n, void* stream) {
  if (!X || !h0 || !H || !dH || !wA || !wB || !Sdense || !pA || !pB || !pb) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (gi && (!dgi || !dgf)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || B > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (!gcrnn_small_dense_supported(dtype, N, G, F, Kin, Kst, 1, gi != nullptr)) return GCRNN_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (dtype == GCRNN_F32)
    return gi ? dense_bwd_launch<float, true>(X, h0, H, dH, wA, wB, bias, gi, gf, Sdense, pA, pB, pb, dgi, dgf, dh0, B, T, N, G, F, Kin, Kst, gate_stride_b, gate_stride_t, gate_stride_n, st)
              : dense_bwd_launch<float, false>(X, h0, H, dH, wA, wB, bias, gi, gf, Sdense, pA, pB, pb, dgi, dgf, dh0, B, T, N, G, F, Kin, Kst, gate_stride_b, gate_stride_t, gate_stride_n, st);
  return gi ? dense_bwd_launch<double, true>(X, h0, H, dH, wA, wB, bias, gi, gf, Sdense, pA, pB, pb, dgi, dgf, dh0, B, T, N, G, F, Kin, Kst, gate_stride_b, gate_stride_t, gate_stride_n, st)
            : dense_bwd_launch<double, false>(X, h0, H, dH, wA, wB, bias, gi, gf, Sdense, pA, pB, pb, dgi, dgf, dh0, B, T, N, G, F, Kin, Kst, gate_stride_b, gate_stride_t, gate_stride_n, st);
}
