// Small-graph regime (N * (G+F) * K values fit in LDS; BASELINE configs[0] and [3]: N = 50..80, T up to 200):
// the whole recurrence of one sequence runs inside ONE workgroup in ONE launch -- no per-step launches, the state
// never leaves the CU. This regime is latency-bound (T dependent steps), not HBM- or MFMA-bound.
//
//   for t:  z_0 = [x_t | h_{t-1}]  (channel-major [c][n], exactly the user layout x[b][t][g][:], H[b][t][f][:])
//           z_k = z_{k-1} S        (K-1 hops; CSR(S^T) held in LDS; one thread per (channel, node))
//           h_t[f][n] = tanh( gi (sum_{k,g} A[f][k][g] z_k[g][n] + b[f]) + gf (sum_{k,f'} B[f][k][f'] z_k[G+f'][n] + b[f]) )
// Reference: GGCRNNCell.forward, Utils/graphML.py:2336-2427 (un-gated and time-gated; gates precomputed per (t, b)).
#include "gcrnn_common.h"

template <typename T>
__device__ __forceinline__ T tanh_t(T v);
template <> __device__ __forceinline__ float tanh_t<float>(float v) { return tanhf(v); }
template <> __device__ __forceinline__ double tanh_t<double>(double v) { return tanh(v); }

template <typename T>
__global__ __launch_bounds__(1024) void small_cell_kernel(
    const T* __restrict__ X,       // [B][Tn][G][N]
    const T* __restrict__ h0,      // [B][F][N]
    const T* __restrict__ wA,      // [F][Kin][G]
    const T* __restrict__ wB,      // [F][Kst][F]
    const T* __restrict__ bias,    // [F] or null
    const T* __restrict__ gi,      // [Tn][B] or null
    const T* __restrict__ gf,      // [Tn][B] or null
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,
    T* __restrict__ H,             // [B][Tn][F][N]
    int Tn, int N, int G, int F, int Kin, int Kst, int nnz, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem_small[];
  const int K = Kin > Kst ? Kin : Kst;
  const int C = G + F;
  T* z = reinterpret_cast<T*>(smem_small);                 // [K][C][N]
  T* wAl = z + (size_t)K * C * N;                          // [F][Kin][G]
  T* wBl = wAl + (size_t)F * Kin * G;                      // [F][Kst][F]
  T* vall = wBl + (size_t)F * Kst * F;                     // [nnz]
  int32_t* rpl = reinterpret_cast<int32_t*>(vall + nnz);   // [N + 1]
  int32_t* coll = rpl + (N + 1);                           // [nnz]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.x;

  for (int i = tid; i < F * Kin * G; i += nt) wAl[i] = wA[i];
  for (int i = tid; i < F * Kst * F; i += nt) wBl[i] = wB[i];
  for (int i = tid; i < nnz; i += nt) { vall[i] = val[i]; coll[i] = col[i]; }
  for (int i = tid; i <= N; i += nt) rpl[i] = rowptr[i];
  for (int i = tid; i < F * N; i += nt) z[(size_t)G * N + i] = h0[(size_t)b * F * N + i];     // z_0[G + f][n] = h0
  __syncthreads();

  // i -> (i / N, i % N) for i = tid, tid + 1024, ...: stepped without divisions (1024 = dq * N + dr)
  const int dq = 1024 / N, dr = 1024 - dq * N;
  const int q0 = tid / N, r0 = tid - q0 * N;
  const int CN = C * N, FN = F * N;

  // x_t is fetched one step ahead (first 1024 values per thread-pass in a register) so that its global-memory
  // latency overlaps the previous step instead of sitting in front of a barrier
  const T* xb = X + (size_t)b * Tn * G * N;
  const int GN = G * N;
  T xpre = (tid < GN) ? xb[tid] : T(0);
  for (int t = 0; t < Tn; ++t) {
    const T* xt = xb + (size_t)t * GN;
    if (tid < GN) z[tid] = xpre;                                                              // z_0[g][n] = x_t
    for (int i = tid + 1024; i < GN; i += 1024) z[i] = xt[i];
    if (t + 1 < Tn && tid < GN) xpre = xt[GN + tid];
    __syncthreads();
    for (int k = 1; k < K; ++k) {                                                            // z_k = z_{k-1} S
      const T* zp = z + (size_t)(k - 1) * C * N;
      T* zn = z + (size_t)k * C * N;
      int cN = q0 * N, n = r0;                       // cN = c * N
      for (int i = tid; i < CN; i += 1024) {
        // channels of a filter with fewer taps than K need no deeper hops, but computing them is harmless
        T acc = T(0);
        const T* zr = zp + cN;
        for (int j = rpl[n]; j < rpl[n + 1]; ++j) acc += vall[j] * zr[coll[j]];
        zn[i] = acc;
        n += dr; cN += dq * N;
        if (n >= N) { n -= N; cN += N; }
      }
      __syncthreads();
    }
    T gin = T(1), gfo = T(1);
    if (gi) { gin = gi[(size_t)t * B + b]; gfo = gf[(size_t)t * B + b]; }
    T* Hout = H + ((size_t)b * Tn + t) * F * N;
    // taps: one output (f, n) per thread pass; all reads of z happen before the barrier, the new state is
    // written to z_0 after it
    T hnew[4];                                      // F * N <= 4 * 1024 is checked on the host
    int f = q0, n = r0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = tid + p * 1024;
      T out = T(0);
      if (i < FN) {
        T ya = T(0), yb = T(0);
        const T* wa = wAl + f * Kin * G;
        const T* wb = wBl + f * Kst * F;
        for (int k = 0; k < Kin; ++k) {
          const T* zc = z + k * CN + n;
          for (int g = 0; g < G; ++g) ya += wa[k * G + g] * zc[g * N];
        }
        for (int k = 0; k < Kst; ++k) {
          const T* zc = z + k * CN + G * N + n;
          const T* wk = wb + k * F;
          int g = 0;
          for (; g + 4 <= F; g += 4)
            yb += wk[g] * zc[g * N] + wk[g + 1] * zc[(g + 1) * N] + wk[g + 2] * zc[(g + 2) * N] + wk[g + 3] * zc[(g + 3) * N];
          for (; g < F; ++g) yb += wk[g] * zc[g * N];
        }
        const T bb = bias ? bias[f] : T(0);
        out = tanh_t<T>(gin * (ya + bb) + gfo * (yb + bb));
        Hout[i] = out;
      }
      hnew[p] = out;
      n += dr; f += dq;
      if (n >= N) { n -= N; f += 1; }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = tid + p * 1024;
      if (i < FN) z[G * N + i] = hnew[p];
    }
    // the x_t copy of the next step touches z_0[0..G), disjoint from what was just written; its barrier orders both
  }
}

static size_t small_lds_bytes(int dtype, int64_t N, int64_t nnz, int64_t G, int64_t F, int64_t Kin, int64_t Kst) {
  const size_t e = dtype == GCRNN_F64 ? 8 : 4;
  const int64_t K = Kin > Kst ? Kin : Kst;
  return e * (size_t)(K * (G + F) * N + F * Kin * G + F * Kst * F + nnz) + 4 * (size_t)(N + 1 + nnz) + 16;
}

extern "C" int gcrnn_small_supported(int dtype, int64_t N, int64_t nnz, int64_t G, int64_t F, int64_t Kin, int64_t Kst) {
  if (dtype != GCRNN_F32 && dtype != GCRNN_F64) return 0;
  if (N <= 0 || G <= 0 || F <= 0 || Kin <= 0 || Kst <= 0 || nnz < 0) return 0;
  if (F * N > 4 * 1024 || N > 1024) return 0;
  return small_lds_bytes(dtype, N, nnz, G, F, Kin, Kst) <= 150 * 1024 ? 1 : 0;
}

template <typename T>
static int small_launch(const void* X, const void* h0, const void* wA, const void* wB, const void* bias, const void* gi,
                        const void* gf, const int32_t* rowptr, const int32_t* col, const void* val, void* H, int64_t B,
                        int64_t Tn, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst, int64_t nnz, size_t lds,
                        hipStream_t st) {
  auto kern = small_cell_kernel<T>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)B, 1024, lds, st>>>((const T*)X, (const T*)h0, (const T*)wA, (const T*)wB, (const T*)bias, (const T*)gi,
                                      (const T*)gf, rowptr, col, (const T*)val, (T*)H, (int)Tn, (int)N, (int)G, (int)F,
                                      (int)Kin, (int)Kst, (int)nnz, (int)B);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_small_forward(int dtype, const void* X, const void* h0, const void* wA, const void* wB,
                                   const void* bias, const void* gi, const void* gf, const int32_t* rowptr,
                                   const int32_t* col, const void* val, void* H, int64_t B, int64_t T, int64_t N,
                                   int64_t G, int64_t F, int64_t Kin, int64_t Kst, int64_t nnz, void* stream) {
  if (!X || !h0 || !wA || !wB || !rowptr || !H || (nnz > 0 && (!col || !val))) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || B > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (!gcrnn_small_supported(dtype, N, nnz, G, F, Kin, Kst)) return GCRNN_ERR_UNSUPPORTED;
  const size_t lds = small_lds_bytes(dtype, N, nnz, G, F, Kin, Kst);
  if (dtype == GCRNN_F32)
    return small_launch<float>(X, h0, wA, wB, bias, gi, gf, rowptr, col, val, H, B, T, N, G, F, Kin, Kst, nnz, lds, as_stream(stream));
  return small_launch<double>(X, h0, wA, wB, bias, gi, gf, rowptr, col, val, H, B, T, N, G, F, Kin, Kst, nnz, lds, as_stream(stream));
}

// ------------------------------------------------------------------------------------------
// BPTT of the small-graph cell: one workgroup per sequence walks t = T-1 .. 0 with everything in LDS -- ONE launch for
// the whole backward pass of the recurrence (the composed path needs ~10 launches per time step).
//   g_t    = dH_t + carry_t                      carry_T-1 = 0
//   dpre_t = g_t (1 - h_t^2)                                                          (tanh)
//   dA[f][k][g]  += gi_t sum_n dpre_t[f][n] (x_t S^k)[g][n]        dB[f][k][f'] += gf_t sum_n dpre_t[f][n] (h_{t-1} S^k)[f'][n]
//   db[f]        += (gi_t + gf_t) sum_n dpre_t[f][n]               (one bias, added by both filters: graphML.py:2420-2421)
//   dgi_t = sum_{f,n} dpre_t (A(S)x_t + b),  dgf_t = sum_{f,n} dpre_t (B(S)h_{t-1} + b)      (time-gated cells)
//   carry_{t-1} = gf_t sum_k (B_k^T dpre_t) (S^T)^k   in Horner form with CSR(S)   (adjoint of graphML.py:118-135)
// The shifted signals z_k are recomputed (K-1 hops) instead of stored. Per-sequence partial sums go to
//   pA [B][2][F][Kin][G], pB [B][2][F][Kst][F], pb [B][F]  (the caller adds them up in a fixed order: deterministic).
// dX is not produced.
// ------------------------------------------------------------------------------------------
template <typename T, int K, bool GATED, int P>      // P = passes of 1024 threads over the 2 F C weight slots and the F N outputs
__global__ __launch_bounds__(1024) void small_cell_bwd_kernel(
    const T* __restrict__ X,        // [B][Tn][G][N]
    const T* __restrict__ h0,       // [B][F][N]
    const T* __restrict__ H,        // [B][Tn][F][N]  forward output
    const T* __restrict__ dH,       // [B][Tn][F][N]
    const T* __restrict__ wA, const T* __restrict__ wB, const T* __restrict__ bias,
    const T* __restrict__ gi, const T* __restrict__ gf,                                     // [Tn][B] (GATED)
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,          // CSR(S^T)
    const int32_t* __restrict__ arowptr, const int32_t* __restrict__ acol, const T* __restrict__ aval,       // CSR(S)
    T* __restrict__ pA, T* __restrict__ pB, T* __restrict__ pb, T* __restrict__ dgi, T* __restrict__ dgf,
    T* __restrict__ dh0,            // [B][F][N] or null
    int Tn, int N, int G, int F, int Kin, int Kst, int nnz, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem_small[];
  const int C = G + F;
  const int Ns = N | 1;                                     // odd row stride: rows of different channels hit different banks
  T* zA = reinterpret_cast<T*>(smem_small);                 // [C][Ns]
  T* zB = zA + (size_t)C * Ns;                              // [C][Ns]
  T* dpre = zB + (size_t)C * Ns;                            // [F][Ns]
  T* carry = dpre + (size_t)F * Ns;                         // [F][Ns]
  T* wAl = carry + (size_t)F * Ns;                          // [F][Kin][G]
  T* wBl = wAl + (size_t)F * Kin * G;                       // [F][Kst][F]
  T* vall = wBl + (size_t)F * Kst * F;                      // [nnz]
  T* avall = vall + nnz;                                    // [nnz]
  T* red = avall + nnz;                                     // [64]
  int32_t* rpl = reinterpret_cast<int32_t*>(red + 64);      // [N + 1]
  int32_t* arpl = rpl + (N + 1);                            // [N + 1]
  int32_t* coll = arpl + (N + 1);                           // [nnz]
  int32_t* acoll = coll + nnz;                              // [nnz]
  const int tid = threadIdx.x;
  const int b = blockIdx.x;

  for (int i = tid; i < F * Kin * G; i += 1024) wAl[i] = wA[i];
  for (int i = tid; i < F * Kst * F; i += 1024) wBl[i] = wB[i];
  for (int i = tid; i < nnz; i += 1024) { vall[i] = val[i]; coll[i] = col[i]; avall[i] = aval[i]; acoll[i] = acol[i]; }
  for (int i = tid; i <= N; i += 1024) { rpl[i] = rowptr[i]; arpl[i] = arowptr[i]; }
  for (int i = tid; i < F * Ns; i += 1024) carry[i] = T(0);
  __syncthreads();

  const int FN = F * N, CN = C * N, GN = G * N;
  // weight-gradient slots: slot s -> (pair = s >> 1 = f * C + c, half = s & 1); up to 4 per thread (checked on the host)
  const int nslots = 2 * F * C;
  const int nh = (N + 1) >> 1;
  T wacc[P][K];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int k = 0; k < K; ++k) wacc[p][k] = T(0);
  T bacc = T(0);                                            // threads f < F: bias partial

  for (int t = Tn - 1; t >= 0; --t) {
    const T* xt = X + ((size_t)b * Tn + t) * GN;
    const T* hp = (t == 0) ? h0 + (size_t)b * FN : H + ((size_t)b * Tn + t - 1) * FN;
    const T* ht = H + ((size_t)b * Tn + t) * FN;
    const T* dht = dH + ((size_t)b * Tn + t) * FN;
    T gin = T(1), gfo = T(1);
    if (GATED) { gin = gi[(size_t)t * B + b]; gfo = gf[(size_t)t * B + b]; }
    // ---- z_0 = [x_t | h_{t-1}], dpre_t
    for (int i = tid; i < CN; i += 1024) {
      const int c = i / N, n = i - c * N;
      zA[c * Ns + n] = (c < G) ? xt[i] : hp[i - GN];
    }
    for (int i = tid; i < FN; i += 1024) {
      const int f = i / N, n = i - f * N;
      const T h = ht[i];
      dpre[f * Ns + n] = (dht[i] + carry[f * Ns + n]) * (T(1) - h * h);
    }
    __syncthreads();
    // ---- taps k = 0 .. K-1: weight-gradient dots against z_k, (gated) A(S)x / B(S)h, next hop
    T ya[P], yb[P];
    if (GATED) {
#pragma unroll
      for (int p = 0; p < P; ++p) { ya[p] = T(0); yb[p] = T(0); }
    }
    T* zc = zA;
    T* zn = zB;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int s = tid + p * 1024;
        if (s < nslots) {
          const int pair = s >> 1, half = s & 1;
          const int f = pair / C, c = pair - f * C;
          if (k < ((c < G) ? Kin : Kst)) {
            const T* dr = dpre + f * Ns;
            const T* zr = zc + c * Ns;
            const int n0 = half * nh, n1 = half ? N : nh;
            T d = T(0);
            for (int n = n0; n < n1; ++n) d += dr[n] * zr[n];
            wacc[p][k] += ((c < G) ? gin : gfo) * d;
          }
        }
      }
      if (GATED) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = tid + p * 1024;
          if (i < FN) {
            const int f = i / N, n = i - f * N;
            if (k < Kin) {
              const T* wa = wAl + (f * Kin + k) * G;
              T a = T(0);
              for (int g = 0; g < G; ++g) a += wa[g] * zc[g * Ns + n];
              ya[p] += a;
            }
            if (k < Kst) {
              const T* wb = wBl + (f * Kst + k) * F;
              T a = T(0);
              for (int g = 0; g < F; ++g) a += wb[g] * zc[(G + g) * Ns + n];
              yb[p] += a;
            }
          }
        }
      }
      if (k + 1 < K) {
        for (int i = tid; i < CN; i += 1024) {
          const int c = i / N, n = i - c * N;
          const T* zr = zc + c * Ns;
          T acc = T(0);
          for (int j = rpl[n]; j < rpl[n + 1]; ++j) acc += vall[j] * zr[coll[j]];
          zn[c * Ns + n] = acc;
        }
        __syncthreads();
        T* tmp = zc; zc = zn; zn = tmp;
      }
    }
    __syncthreads();                        // every read of the z buffers is done: they become the adjoint accumulators
    // ---- bias / gate partial sums
    if (tid < F) {
      const T* dr = dpre + tid * Ns;
      T s = T(0);
      for (int n = 0; n < N; ++n) s += dr[n];
      bacc += (gin + gfo) * s;
    }
    if (GATED) {
      T si = T(0), sf = T(0);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = tid + p * 1024;
        if (i < FN) {
          const int f = i / N, n = i - f * N;
          const T bb = bias ? bias[f] : T(0);
          const T d = dpre[f * Ns + n];
          si += d * (ya[p] + bb);
          sf += d * (yb[p] + bb);
        }
      }
      for (int o = 32; o > 0; o >>= 1) { si += __shfl_down(si, o, 64); sf += __shfl_down(sf, o, 64); }
      if ((tid & 63) == 0) { red[(tid >> 6) * 2] = si; red[(tid >> 6) * 2 + 1] = sf; }
      __syncthreads();
      if (tid == 0) {
        T a = T(0), c2 = T(0);
        for (int w = 0; w < 16; ++w) { a += red[2 * w]; c2 += red[2 * w + 1]; }
        dgi[(size_t)t * B + b] = a;
        dgf[(size_t)t * B + b] = c2;
      }
    }
    // ---- carry_{t-1} = gf sum_k (B_k^T dpre)(S^T)^k : acc <- acc S^T + B_k^T dpre, k = Kst-1 .. 0 (acc in zA / zB, [F][Ns])
    T* ac = zA;
    T* an = zB;
    for (int k = Kst - 1; k >= 0; --k) {
      T* dst = (k == 0) ? carry : an;
      for (int i = tid; i < FN; i += 1024) {
        const int f2 = i / N, n = i - f2 * N;
        T acc = T(0);
        for (int f = 0; f < F; ++f) acc += wBl[(f * Kst + k) * F + f2] * dpre[f * Ns + n];
        acc *= gfo;
        if (k < Kst - 1) {
          const T* ar = ac + f2 * Ns;
          for (int j = arpl[n]; j < arpl[n + 1]; ++j) acc += avall[j] * ar[acoll[j]];
        }
        dst[f2 * Ns + n] = acc;
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
  for (int p = 0; p < P; ++p) {
    const int s = tid + p * 1024;
    if (s < nslots) {
      const int pair = s >> 1, half = s & 1;
      const int f = pair / C, c = pair - f * C;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (c < G) { if (k < Kin) pA[(((size_t)b * 2 + half) * F + f) * Kin * G + k * G + c] = wacc[p][k]; }
        else       { if (k < Kst) pB[(((size_t)b * 2 + half) * F + f) * Kst * F + k * F + (c - G)] = wacc[p][k]; }
      }
    }
  }
}

static size_t small_bwd_lds_bytes(int dtype, int64_t N, int64_t nnz, int64_t G, int64_t F, int64_t Kin, int64_t Kst) {
  const size_t e = dtype == GCRNN_F64 ? 8 : 4;
  const int64_t Ns = N | 1, C = G + F;
  return e * (size_t)(2 * C * Ns + 2 * F * Ns + F * Kin * G + F * Kst * F + 2 * nnz + 64) + 4 * (size_t)(2 * (N + 1) + 2 * nnz) + 16;
}

extern "C" int gcrnn_small_backward_supported(int dtype, int64_t N, int64_t nnz, int64_t G, int64_t F, int64_t Kin,
                                              int64_t Kst) {
  if (dtype != GCRNN_F32 && dtype != GCRNN_F64) return 0;
  if (N <= 0 || G <= 0 || F <= 0 || Kin <= 0 || Kst <= 0 || nnz < 0) return 0;
  const int64_t K = Kin > Kst ? Kin : Kst;
  if (K > 5 || F * N > 4 * 1024 || N > 1024 || 2 * F * (G + F) > 4 * 1024) return 0;
  return small_bwd_lds_bytes(dtype, N, nnz, G, F, Kin, Kst) <= 150 * 1024 ? 1 : 0;
}

template <typename T, int K, bool GATED, int P>
static int small_bwd_launch(const void* X, const void* h0, const void* H, const void* dH, const void* wA, const void* wB,
                            const void* bias, const void* gi, const void* gf, const int32_t* rowptr, const int32_t* col,
                            const void* val, const int32_t* arowptr, const int32_t* acol, const void* aval, void* pA, void* pB,
                            void* pb, void* dgi, void* dgf, void* dh0, int64_t B, int64_t Tn, int64_t N, int64_t G, int64_t F,
                            int64_t Kin, int64_t Kst, int64_t nnz, size_t lds, hipStream_t st) {
  auto kern = small_cell_bwd_kernel<T, K, GATED, P>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  kern<<<(unsigned)B, 1024, lds, st>>>((const T*)X, (const T*)h0, (const T*)H, (const T*)dH, (const T*)wA, (const T*)wB,
                                      (const T*)bias, (const T*)gi, (const T*)gf, rowptr, col, (const T*)val, arowptr, acol,
                                      (const T*)aval, (T*)pA, (T*)pB, (T*)pb, (T*)dgi, (T*)dgf, (T*)dh0, (int)Tn, (int)N,
                                      (int)G, (int)F, (int)Kin, (int)Kst, (int)nnz, (int)B);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

template <typename T>
static int small_bwd_dispatch(bool gated, int64_t K, const void* X, const void* h0, const void* H, const void* dH,
                              const void* wA, const void* wB, const void* bias, const void* gi, const void* gf,
                              const int32_t* rowptr, const int32_t* col, const void* val, const int32_t* arowptr,
                              const int32_t* acol, const void* aval, void* pA, void* pB, void* pb, void* dgi, void* dgf,
                              void* dh0, int64_t B, int64_t Tn, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst,
                              int64_t nnz, size_t lds, hipStream_t st) {
  const bool two = F * N <= 2048 && 2 * F * (G + F) <= 2048;      // the drivers' sizes: fewer accumulators per thread, no spills
#define GCRNN_SMALL_BWD_ARGS X, h0, H, dH, wA, wB, bias, gi, gf, rowptr, col, val, arowptr, acol, aval, pA, pB, pb, dgi, dgf, \
                             dh0, B, Tn, N, G, F, Kin, Kst, nnz, lds, st
#define GCRNN_SMALL_BWD_CASE(KK)                                                                                   \
  if (K == KK) {                                                                                                   \
    if (gated) return two ? small_bwd_launch<T, KK, true, 2>(GCRNN_SMALL_BWD_ARGS)                                 \
                          : small_bwd_launch<T, KK, true, 4>(GCRNN_SMALL_BWD_ARGS);                                \
    return two ? small_bwd_launch<T, KK, false, 2>(GCRNN_SMALL_BWD_ARGS)                                           \
               : small_bwd_launch<T, KK, false, 4>(GCRNN_SMALL_BWD_ARGS);                                          \
  }
  GCRNN_SMALL_BWD_CASE(1)
  GCRNN_SMALL_BWD_CASE(2)
  GCRNN_SMALL_BWD_CASE(3)
  GCRNN_SMALL_BWD_CASE(4)
  GCRNN_SMALL_BWD_CASE(5)
#undef GCRNN_SMALL_BWD_CASE
#undef GCRNN_SMALL_BWD_ARGS
  return GCRNN_ERR_UNSUPPORTED;
}

extern "C" int gcrnn_small_backward(int dtype, const void* X, const void* h0, const void* H, const void* dH, const void* wA,
                                    const void* wB, const void* bias, const void* gi, const void* gf, const int32_t* rowptr,
                                    const int32_t* col, const void* val, const int32_t* arowptr, const int32_t* acol,
                                    const void* aval, void* pA, void* pB, void* pb, void* dgi, void* dgf, void* dh0,
                                    int64_t B, int64_t T, int64_t N, int64_t G, int64_t F, int64_t Kin, int64_t Kst,
                                    int64_t nnz, void* stream) {
  if (!X || !h0 || !H || !dH || !wA || !wB || !rowptr || !arowptr || !pA || !pB || !pb) return GCRNN_ERR_NULL_POINTER;
  if (nnz > 0 && (!col || !val || !acol || !aval)) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (gi && (!dgi || !dgf)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || B > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (!gcrnn_small_backward_supported(dtype, N, nnz, G, F, Kin, Kst)) return GCRNN_ERR_UNSUPPORTED;
  const size_t lds = small_bwd_lds_bytes(dtype, N, nnz, G, F, Kin, Kst);
  const int64_t K = Kin > Kst ? Kin : Kst;
  if (dtype == GCRNN_F32)
    return small_bwd_dispatch<float>(gi != nullptr, K, X, h0, H, dH, wA, wB, bias, gi, gf, rowptr, col, val, arowptr, acol,
                                     aval, pA, pB, pb, dgi, dgf, dh0, B, T, N, G, F, Kin, Kst, nnz, lds, as_stream(stream));
  return small_bwd_dispatch<double>(gi != nullptr, K, X, h0, H, dH, wA, wB, bias, gi, gf, rowptr, col, val, arowptr, acol,
                                    aval, pA, pB, pb, dgi, dgf, dh0, B, T, N, G, F, Kin, Kst, nnz, lds, as_stream(stream));
}
