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
