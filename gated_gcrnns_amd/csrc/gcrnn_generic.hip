// Any-shape kernels of the GCRNN hot path for gfx950: layout pack/unpack, batched CSR row
// SpMM (the graph shift), and the filter-tap GEMMs (forward / data-grad / weight-grad).
// These are the "streaming" regime of DESIGN.md: every hop is one HBM/L2 pass over a
// node-major [N][L] matrix. The fused per-sequence kernels live in gcrnn_small.hip /
// gcrnn_fused.hip.
#include "gcrnn_common.h"

// ------------------------------------------------------------------------------------------
// layout: user [B][T][C][N]  <->  node-major [T][N][B][C]
// For a fixed t this is a transpose of the (Q = B*C) x N matrix whose row q = (b, c) starts at
// ((b*T + t)*C + c)*N.  32x32 LDS tile, 256 threads (32 x 8).
// ------------------------------------------------------------------------------------------
template <typename T, bool PACK>
__global__ __launch_bounds__(256) void layout_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t B,
                                                     int64_t Tn, int64_t C, int64_t N,
                                                     const int32_t* __restrict__ perm, int64_t nsum = 1) {
  // nsum > 1 (PACK): the user side is [B][nsum][T][C][N] and the nsum slices are ADDED on the way (in order: deterministic) -- partial sums
  // that a producer kernel stored per slice (the node gates' tap dots per 32-feature chunk) cost no pass of their own
  __shared__ T tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t n0 = (int64_t)blockIdx.x * 32, q0 = (int64_t)blockIdx.y * 32, t = blockIdx.z;
  const int64_t Q = B * C;
  if (PACK) {
    // read user rows q (coalesced along n), write node-major rows n (coalesced along q)
    const int64_t n = n0 + tx;
    const int64_t nsrc = (n < N) ? (perm ? (int64_t)perm[n] : n) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t q = q0 + ty + 8 * i;
      T v = T(0);
      if (q < Q && n < N) {
        const int64_t b = q / C, c = q - b * C;
        v = src[(((b * nsum) * Tn + t) * C + c) * N + nsrc];
        for (int64_t sl = 1; sl < nsum; ++sl) v += src[(((b * nsum + sl) * Tn + t) * C + c) * N + nsrc];
      }
      tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t nn = n0 + ty + 8 * i, q = q0 + tx;
      if (nn < N && q < Q) dst[(t * N + nn) * Q + q] = tile[tx][ty + 8 * i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t nn = n0 + ty + 8 * i, q = q0 + tx;
      tile[ty + 8 * i][tx] = (nn < N && q < Q) ? src[(t * N + nn) * Q + q] : T(0);
    }
    __syncthreads();
    const int64_t n = n0 + tx;
    const int64_t ndst = (n < N) ? (perm ? (int64_t)perm[n] : n) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t q = q0 + ty + 8 * i;
      if (q < Q && n < N) {
        const int64_t b = q / C, c = q - b * C;
        dst[((b * Tn + t) * C + c) * N + ndst] = tile[tx][ty + 8 * i];
      }
    }
  }
}

template <bool PACK>
static int layout_launch(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C, int64_t N,
                         const int32_t* perm, void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || C <= 0 || N <= 0 || T > 65535) return GCRNN_ERR_BAD_SHAPE;
  const int64_t gy = cdiv(B * C, 32);
  if (gy > 65535) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  dim3 grid((unsigned)cdiv(N, 32), (unsigned)gy, (unsigned)T);
  if (dtype == GCRNN_F32)
    layout_kernel<float, PACK><<<grid, 256, 0, as_stream(stream)>>>((const float*)src, (float*)dst, B, T, C, N, perm);
  else if (dtype == GCRNN_F64)
    layout_kernel<double, PACK><<<grid, 256, 0, as_stream(stream)>>>((const double*)src, (double*)dst, B, T, C, N, perm);
  else if (dtype == GCRNN_BF16)
    layout_kernel<uint16_t, PACK><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)src, (uint16_t*)dst, B, T, C,
                                                                         N, perm);  // bf16 moves as raw 16-bit words
  else
    return GCRNN_ERR_BAD_DTYPE;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_pack_node_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C,
                                     int64_t N, const int32_t* perm, void* stream) {
  return layout_launch<true>(dtype, src, dst, B, T, C, N, perm, stream);
}
// gcrnn_pack_node_major of the SUM over S slices: src fp32 [B][S][T][C][N] -> dst [T][N][B][C] = sum_s src[b][s][t][c][n] (in slice order).
extern "C" int gcrnn_pack_node_major_sum_f32(const void* src, void* dst, int64_t B, int64_t S, int64_t T, int64_t C, int64_t N, void* stream) {
  if (!src || !dst) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || S <= 0 || T <= 0 || C <= 0 || N <= 0 || T > 65535) return GCRNN_ERR_BAD_SHAPE;
  const int64_t gy = cdiv(B * C, 32);
  if (gy > 65535) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  dim3 grid((unsigned)cdiv(N, 32), (unsigned)gy, (unsigned)T);
  layout_kernel<float, true><<<grid, 256, 0, as_stream(stream)>>>((const float*)src, (float*)dst, B, T, C, N, nullptr, S);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}
// ------------------------------------------------------------------------------------------
// second stage of the node gates' F -> 1 GraphFilter (Utils/graphML.py:2387-2399) in ONE pass: per item the K one-channel signals
// u_k = sum_s parts[item][s][k][:] (the per-chunk tap dots the gate pre-pass stored) run the K - 1 Horner hops  a <- u_k + P a, then bias +
// sigmoid, written where the node-gated recurrence reads its gates. HBM-bound on the partials (S K N floats per item in, N out), so:
// persistent workgroups of 512 threads (thread = two nodes) over groups of 4 items, the CSR rows of P copied into LDS once per workgroup (col
// as u16: N <= 1024; no weights at all for a uniform-weight graph), the fetches of u_k issued one hop ahead of their use, the running signal
// of the 4 items in LDS as one float4 per node (one 16-byte gather per CSR entry), <= 64 registers so that up to four workgroups share a CU.
// ------------------------------------------------------------------------------------------
template <int K, bool UNI>
__global__ __launch_bounds__(512, 8) void node_gate_filter_kernel(const float* __restrict__ parts, int S, int N, int64_t items, int groups, int64_t B,
                                                               const int* __restrict__ rowptr, const int* __restrict__ col,
                                                               const float* __restrict__ val, int nnz, float uniform_w,
                                                               const float* __restrict__ bias, int sigmoid, float* __restrict__ out,
                                                               int64_t ngroups) {
  constexpr int I = 4;
  extern __shared__ __align__(16) unsigned char node_gate_lds[];
  float4* acc = reinterpret_cast<float4*>(node_gate_lds);                        // [N] x 4 items
  float* vv = reinterpret_cast<float*>(acc + N);                                  // [nnz] (not for UNI)
  uint16_t* cc = reinterpret_cast<uint16_t*>(vv + (UNI ? 0 : nnz));               // [nnz]
  const int tid = threadIdx.x;
  for (int j = tid; j < nnz; j += 512) {
    if (!UNI) vv[j] = val[j];
    cc[j] = (uint16_t)col[j];
  }
  int j0[2], j1[2], nc[2];
  bool on[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int n = tid + 512 * r;
    on[r] = n < N;
    nc[r] = on[r] ? n : N - 1;
    j0[r] = on[r] ? rowptr[n] : 0;
    j1[r] = on[r] ? rowptr[n + 1] : 0;
  }
  const int64_t kstride = (int64_t)K * N, istride = (int64_t)S * kstride;
  // u_k of group grp for this thread's two rows: S x 8 fetches, no branch around any of them (rows past N and items past the end read a
  // clamped address and are never stored -- a conditional fetch keeps its wait inside the branch and the fetches run one round trip after
  // the other). Fetches run ONE hop ahead of their use (and the next group's first signal during the last hop), so that they are in flight
  // while the rows gather; 16 of them per thread keeps the kernel at <= 64 registers: four workgroups per CU.
  auto fetch = [&](float (&dst)[2][I], int64_t grp, int k) {
    const int64_t i0 = grp * I;
    const float* pi[I];
#pragma unroll
    for (int i = 0; i < I; ++i) pi[i] = parts + (i0 + i < items ? i0 + i : items - 1) * istride + (int64_t)k * N;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < I; ++i) dst[r][i] = pi[i][nc[r]];
    if (S == 2) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < I; ++i) dst[r][i] += pi[i][kstride + nc[r]];
    } else {
      for (int s = 1; s < S; ++s) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < I; ++i) dst[r][i] += pi[i][s * kstride + nc[r]];
      }
    }
  };
  float un[2][I];
  if ((int64_t)blockIdx.x < ngroups) fetch(un, blockIdx.x, K - 1);
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t i0 = grp * I;
    const int64_t nxt = grp + gridDim.x < ngroups ? grp + gridDim.x : grp;      // (the last group prefetches itself: harmless)
    float a[2][I];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < I; ++i) a[r][i] = un[r][i];
    if (K > 1) fetch(un, grp, K - 2);
    else fetch(un, nxt, K - 1);
#pragma unroll
    for (int k = K - 2; k >= 0; --k) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (on[r]) acc[tid + 512 * r] = make_float4(a[r][0], a[r][1], a[r][2], a[r][3]);
      __syncthreads();      // (the first one of a workgroup also covers the CSR copy)
      float uk[2][I];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < I; ++i) uk[r][i] = un[r][i];
      if (k > 0) fetch(un, grp, k - 1);
      else fetch(un, nxt, K - 1);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = j0[r]; j < j1[r]; ++j) {
          const float4 v = acc[cc[j]];
          const float w = UNI ? 1.f : vv[j];
          sum.x = fmaf(w, v.x, sum.x);
          sum.y = fmaf(w, v.y, sum.y);
          sum.z = fmaf(w, v.z, sum.z);
          sum.w = fmaf(w, v.w, sum.w);
        }
        const float sc = UNI ? uniform_w : 1.f;
        a[r][0] = fmaf(sc, sum.x, uk[r][0]);
        a[r][1] = fmaf(sc, sum.y, uk[r][1]);
        a[r][2] = fmaf(sc, sum.z, uk[r][2]);
        a[r][3] = fmaf(sc, sum.w, uk[r][3]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < I; ++i) {
      const int64_t item = i0 + i;
      if (item < items) {
        const int g = (int)(item % groups);
        const int64_t ib = item / groups, t = ib / B, b = ib % B;
        const float bg = bias ? bias[g] : 0.f;
        float* o = out + ((t * groups + g) * B + b) * N;
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (on[r]) {
            const float v = a[r][i] + bg;
            o[tid + 512 * r] = sigmoid ? 1.f / (1.f + __expf(-v)) : v;
          }
      }
    }
  }
}
static inline size_t node_gate_filter_lds(int64_t N, int64_t nnz, bool uni) { return (size_t)N * 16 + (size_t)nnz * (uni ? 2 : 6) + 16; }
template <int K, bool UNI>
static int node_gate_filter_launch(const float* parts, float* out, int64_t items, int S, int N, int groups, int64_t B, const int32_t* rowptr,
                                   const int32_t* col, const float* val, int nnz, float uniform_w, const float* bias, int sigmoid, void* stream) {
  const size_t lds = node_gate_filter_lds(N, nnz, UNI);
  auto kern = node_gate_filter_kernel<K, UNI>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  const int64_t ngroups = cdiv(items, (int64_t)4);
  int64_t per_cu = (int64_t)(160 * 1024 / lds);      // workgroups of 8 waves per CU: LDS, and the register file (<= 64 VGPRs) allows 4
  per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
  const int64_t slots = 256 * per_cu;
  const unsigned grid = (unsigned)(ngroups < slots ? ngroups : slots);
  kern<<<dim3(grid), 512, lds, as_stream(stream)>>>(parts, S, N, items, groups, B, rowptr, col, val, nnz, uniform_w, bias, sigmoid, out, ngroups);
  return GCRNN_OK;
}
// 1 where gcrnn_node_gate_filter_f32 runs this problem (K <= 5 taps, N <= 1024, the CSR rows beside the running signal in LDS), else 0.
extern "C" int gcrnn_node_gate_filter_supported(int64_t K, int64_t N, int64_t nnz, double uniform_w) {
  return (K >= 1 && K <= 5 && N >= 1 && N <= 1024 && nnz >= 0 && nnz < (1 << 24) && node_gate_filter_lds(N, nnz, uniform_w != 0.0) <= 160 * 1024) ? 1 : 0;
}
// parts fp32 [items][S][K][N], item = (t B + b) groups + g  ->  out fp32 [T][groups][B][N] = act(sum_k P^k u_k + bias[g]); P = the CSR rows
// (rowptr int32[N + 1], col int32[nnz], val fp32[nnz]; uniform_w != 0: every stored entry has this weight and val is not read),
// bias fp32[groups] or NULL, act = sigmoid or identity.
extern "C" int gcrnn_node_gate_filter_f32(const void* parts, void* out, int64_t items, int64_t S, int64_t K, int64_t N, int64_t groups, int64_t B,
                                          const int32_t* rowptr, const int32_t* col, const void* val, int64_t nnz, double uniform_w,
                                          const void* bias, int sigmoid, void* stream) {
  const bool uni = uniform_w != 0.0;
  if (!parts || !out || !rowptr || (nnz > 0 && (!col || (!uni && !val)))) return GCRNN_ERR_NULL_POINTER;
  if (items <= 0 || S <= 0 || groups <= 0 || B <= 0 || items % (groups * B) || !gcrnn_node_gate_filter_supported(K, N, nnz, uniform_w))
    return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  int rc = GCRNN_OK;
#define NGF(KK)                                                                                                                             \
  rc = uni ? node_gate_filter_launch<KK, true>((const float*)parts, (float*)out, items, (int)S, (int)N, (int)groups, B, rowptr, col,        \
                                               (const float*)val, (int)nnz, (float)uniform_w, (const float*)bias, sigmoid, stream)         \
           : node_gate_filter_launch<KK, false>((const float*)parts, (float*)out, items, (int)S, (int)N, (int)groups, B, rowptr, col,       \
                                                (const float*)val, (int)nnz, 0.f, (const float*)bias, sigmoid, stream)
  switch ((int)K) {
    case 1: NGF(1); break;
    case 2: NGF(2); break;
    case 3: NGF(3); break;
    case 4: NGF(4); break;
    default: NGF(5); break;
  }
#undef NGF
  if (rc != GCRNN_OK) return rc;
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}
extern "C" int gcrnn_unpack_node_major(int dtype, const void* src, void* dst, int64_t B, int64_t T, int64_t C,
                                       int64_t N, const int32_t* perm, void* stream) {
  return layout_launch<false>(dtype, src, dst, B, T, C, N, perm, stream);
}

// ------------------------------------------------------------------------------------------
// graph shift: Y[i][n][:] = sum_j val[j] * X[i][col[j]][:]
// The vector path (rows that are whole 16-byte vectors, 16-byte aligned) is the streaming kernel of gcrnn_spmm.hip
// (gcrnn_spmm_ex). What stays here is the scalar fallback for odd row lengths: one workgroup per (row n, column block,
// batch i), wave-uniform neighbour loop, one element per lane.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void spmm_scalar_kernel(int64_t N, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                   const T* __restrict__ val, const T* __restrict__ X, T* __restrict__ Y, int64_t L,
                                   int accumulate) {
  const int64_t n = blockIdx.x;
  const int64_t l = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int64_t base = (int64_t)blockIdx.z * N * L;
  const int s = rowptr[n], e = rowptr[n + 1];
  T acc = T(0);
  const T* xb = X + base + l;
  for (int j = s; j < e; ++j) acc += val[j] * xb[(int64_t)col[j] * L];
  T* yp = Y + base + n * L + l;
  yp[0] = accumulate ? (yp[0] + acc) : acc;
}

template <typename T>
static int spmm_scalar_launch(int64_t N, const int32_t* rowptr, const int32_t* col, const void* val, const void* X, void* Y,
                              int64_t L, int64_t nbatch, int accumulate, void* stream) {
  int threads = L >= 256 ? 256 : (int)(cdiv(L, 64) * 64);
  const int64_t gy = cdiv(L, threads);
  if (gy > 65535 || nbatch > 65535) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  dim3 grid((unsigned)N, (unsigned)gy, (unsigned)nbatch);
  spmm_scalar_kernel<T><<<grid, threads, 0, as_stream(stream)>>>(N, rowptr, col, (const T*)val, (const T*)X, (T*)Y, L, accumulate);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_spmm(int dtype, int64_t N, const int32_t* rowptr, const int32_t* col, const void* val,
                          const void* X, void* Y, int64_t L, int64_t nbatch, int accumulate, void* stream) {
  if (!rowptr || !X || !Y) return GCRNN_ERR_NULL_POINTER;
  if (N <= 0 || L <= 0 || nbatch <= 0 || N > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (X == Y) return GCRNN_ERR_UNSUPPORTED;  // a hop cannot run in place
  const int ve = dtype == GCRNN_F32 ? 4 : (dtype == GCRNN_F64 ? 2 : (dtype == GCRNN_BF16 ? 8 : 0));
  if (!ve) return GCRNN_ERR_BAD_DTYPE;
  const bool vec = (L % ve == 0) && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) % 16 == 0);
  if (vec) return gcrnn_spmm_ex(dtype, N, rowptr, col, val, X, Y, L, nbatch, accumulate, nullptr, 0.0, 0, 0, 0, 0, 0, stream);
  if (dtype == GCRNN_F32) return spmm_scalar_launch<float>(N, rowptr, col, val, X, Y, L, nbatch, accumulate, stream);
  if (dtype == GCRNN_F64) return spmm_scalar_launch<double>(N, rowptr, col, val, X, Y, L, nbatch, accumulate, stream);
  return GCRNN_ERR_UNSUPPORTED;                 // bf16 rows must be whole 16-byte vectors
}

// ------------------------------------------------------------------------------------------
// filter taps: three GEMM flavours over one 64x64x16 LDS-tiled kernel with accessor functors.
//   C(i, j) (+)= sum_k A(i, k) * B(k, j),  k in [blockIdx.z * ksplit, ...)
// ------------------------------------------------------------------------------------------
template <typename T>
struct TapOperand {  // element (r, kk) of the concatenated hop matrix [rows][KK*G]
  const T* z0;
  const T* zrest;
  int64_t zstride;
  int G;
  __device__ __forceinline__ const T* ptr(int64_t r, int64_t kk) const {
    const int64_t k = kk / G, g = kk - k * G;
    const T* base = (k == 0) ? z0 : (zrest + (k - 1) * zstride);
    return base + r * G + g;
  }
};

// forward: A(i,k) = Zcat[i][k] (k fast), B(k,j) = w[j][k] (k fast), C -> y[i][j]
template <typename T>
struct FwdA { TapOperand<T> z; static constexpr bool KFAST = true;
  __device__ __forceinline__ T operator()(int64_t i, int64_t k) const { return *z.ptr(i, k); } };
template <typename T>
struct FwdB { const T* w; int64_t Kd; static constexpr bool KFAST = true;
  __device__ __forceinline__ T operator()(int64_t k, int64_t j) const { return w[j * Kd + k]; } };
template <typename T>
struct FwdC { T* y; const T* bias; T bias_scale; int F; int accumulate;
  __device__ __forceinline__ void operator()(int64_t i, int64_t j, T v) const {
    if (bias) v += bias_scale * bias[j];
    T* p = y + i * F + j;
    *p = accumulate ? (*p + v) : v;
  } };

// backward data: A(i,k=f) = dy[i][f] (k fast), B(k=f, j=kk) = w[f][kk] (j fast), C -> dz_k[i][g]
template <typename T>
struct BdA { const T* dy; int F; static constexpr bool KFAST = true;
  __device__ __forceinline__ T operator()(int64_t i, int64_t k) const { return dy[i * F + k]; } };
template <typename T>
struct BdB { const T* w; int64_t Kd; static constexpr bool KFAST = false;
  __device__ __forceinline__ T operator()(int64_t k, int64_t j) const { return w[k * Kd + j]; } };
template <typename T>
struct BdC { TapOperand<T> dz;
  __device__ __forceinline__ void operator()(int64_t i, int64_t j, T v) const { *const_cast<T*>(dz.ptr(i, j)) = v; } };

// backward weight: A(i=f, k=r) = dy[r][f] (i fast), B(k=r, j=kk) = Zcat[r][kk] (j fast), C -> dw_part[split][f][kk]: every row
// split (blockIdx.z) stores its own partial sum, the caller adds the splits in a fixed order (deterministic, no atomics)
template <typename T>
struct BwA { const T* dy; int F; static constexpr bool KFAST = false;
  __device__ __forceinline__ T operator()(int64_t i, int64_t k) const { return dy[k * F + i]; } };
template <typename T>
struct BwB { TapOperand<T> z; static constexpr bool KFAST = false;
  __device__ __forceinline__ T operator()(int64_t k, int64_t j) const { return *z.ptr(k, j); } };
template <typename T>
struct BwC { T* dw; int64_t Kd; int64_t FKd;
  __device__ __forceinline__ void operator()(int64_t i, int64_t j, T v) const { dw[(int64_t)blockIdx.z * FKd + i * Kd + j] = v; } };

template <typename T, typename AL, typename BL, typename CS>
__global__ __launch_bounds__(256) void gemm64_kernel(AL a, BL b, CS c, int64_t M, int64_t Nc, int64_t Kd,
                                                     int64_t ksplit) {
  __shared__ T As[16][64 + 4];
  __shared__ T Bs[16][64 + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 64, n0 = (int64_t)blockIdx.y * 64;
  const int64_t kbeg = (int64_t)blockIdx.z * ksplit;
  const int64_t kend = (kbeg + ksplit < Kd) ? (kbeg + ksplit) : Kd;
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = T(0);

  for (int64_t k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      const int r = AL::KFAST ? (e >> 4) : (e & 63);
      const int kk = AL::KFAST ? (e & 15) : (e >> 6);
      As[kk][r] = (m0 + r < M && k0 + kk < kend) ? a(m0 + r, k0 + kk) : T(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      const int cc = BL::KFAST ? (e >> 4) : (e & 63);
      const int kk = BL::KFAST ? (e & 15) : (e >> 6);
      Bs[kk][cc] = (n0 + cc < Nc && k0 + kk < kend) ? b(k0 + kk, n0 + cc) : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      T av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + tx * 4 + j;
      if (n < Nc) c(m, n, acc[i][j]);
    }
  }
}

// dbias_part[block][f] = scale * sum over the block's rows of dy[r][f].  256 threads = (256 / FP) row lanes x FP column lanes
// (FP = F rounded up to a power of two, <= 256): every thread sums a strided subset of the block's rows, an LDS tree folds
// the row lanes, one plain store per column and block (the caller adds the blocks in a fixed order).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, T* __restrict__ dbias, T scale,
                                                     int64_t rows, int F, int FP, int64_t rows_per_block) {
  __shared__ T part[256];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  const int RL = 256 / FP;                       // row lanes
  const int cf = threadIdx.x % FP, rl = threadIdx.x / FP;
  for (int f0 = 0; f0 < F; f0 += FP) {
    const int f = f0 + cf;
    T s = T(0);
    if (f < F)
      for (int64_t r = r0 + rl; r < r1; r += RL) s += dy[r * F + f];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int step = RL >> 1; step > 0; step >>= 1) {
      if (rl < step) part[threadIdx.x] += part[threadIdx.x + step * FP];
      __syncthreads();
    }
    if (rl == 0 && f < F) dbias[(int64_t)blockIdx.x * F + f] = scale * part[cf];
    __syncthreads();
  }
}

static bool tap_shape_ok(int64_t rows, int64_t KK, int64_t G, int64_t F) {
  return rows > 0 && KK > 0 && G > 0 && F > 0 && G < (1 << 30) && F < (1 << 30) && KK * G < (1LL << 31);
}

template <typename T>
static int taps_fwd(const void* z0, const void* zrest, int64_t zstride, const void* w, const void* bias,
                    double bias_scale, void* y, int64_t rows, int64_t KK, int64_t G, int64_t F, int accumulate,
                    void* stream) {
  const int64_t Kd = KK * G;
  FwdA<T> a{{(const T*)z0, (const T*)zrest, zstride, (int)G}};
  FwdB<T> b{(const T*)w, Kd};
  FwdC<T> c{(T*)y, (const T*)bias, (T)bias_scale, (int)F, accumulate};
  GCRNN_PRE_LAUNCH();
  dim3 grid((unsigned)cdiv(rows, 64), (unsigned)cdiv(F, 64), 1);
  gemm64_kernel<T><<<grid, 256, 0, as_stream(stream)>>>(a, b, c, rows, F, Kd, Kd);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_taps_forward(int dtype, const void* z0, const void* zrest, int64_t zstride, const void* w,
                                  const void* bias, double bias_scale, void* y, int64_t rows, int64_t KK, int64_t G,
                                  int64_t F, int accumulate, void* stream) {
  if (!z0 || !w || !y || (KK > 1 && !zrest)) return GCRNN_ERR_NULL_POINTER;
  if (!tap_shape_ok(rows, KK, G, F) || cdiv(F, 64) > 65535) return GCRNN_ERR_BAD_SHAPE;
  if (dtype == GCRNN_F32) return taps_fwd<float>(z0, zrest, zstride, w, bias, bias_scale, y, rows, KK, G, F, accumulate, stream);
  if (dtype == GCRNN_F64) return taps_fwd<double>(z0, zrest, zstride, w, bias, bias_scale, y, rows, KK, G, F, accumulate, stream);
  return GCRNN_ERR_BAD_DTYPE;
}

template <typename T>
static int taps_bwd_data(const void* dy, const void* w, void* dz0, void* dzrest, int64_t zstride, int64_t rows,
                         int64_t KK, int64_t G, int64_t F, void* stream) {
  const int64_t Kd = KK * G;
  BdA<T> a{(const T*)dy, (int)F};
  BdB<T> b{(const T*)w, Kd};
  BdC<T> c{{(const T*)dz0, (const T*)dzrest, zstride, (int)G}};
  GCRNN_PRE_LAUNCH();
  dim3 grid((unsigned)cdiv(rows, 64), (unsigned)cdiv(Kd, 64), 1);
  gemm64_kernel<T><<<grid, 256, 0, as_stream(stream)>>>(a, b, c, rows, Kd, F, F);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

extern "C" int gcrnn_taps_backward_data(int dtype, const void* dy, const void* w, void* dz0, void* dzrest,
                                        int64_t zstride, int64_t rows, int64_t KK, int64_t G, int64_t F,
                                        void* stream) {
  if (!dy || !w || !dz0 || (KK > 1 && !dzrest)) return GCRNN_ERR_NULL_POINTER;
  if (!tap_shape_ok(rows, KK, G, F) || cdiv(KK * G, 64) > 65535) return GCRNN_ERR_BAD_SHAPE;
  if (dtype == GCRNN_F32) return taps_bwd_data<float>(dy, w, dz0, dzrest, zstride, rows, KK, G, F, stream);
  if (dtype == GCRNN_F64) return taps_bwd_data<double>(dy, w, dz0, dzrest, zstride, rows, KK, G, F, stream);
  return GCRNN_ERR_BAD_DTYPE;
}

// row splits of the weight-gradient GEMM and row blocks of the bias reduction for a (rows, KK, G, F) problem
static void bwd_weight_plan(int64_t rows, int64_t KK, int64_t G, int64_t F, int64_t* splits, int64_t* ksplit, int64_t* bblocks, int64_t* rpb) {
  const int64_t Kd = KK * G;
  // split the (huge) row reduction so that the grid fills the chip: aim at >= 1024 workgroups
  const int64_t tiles = cdiv(F, 64) * cdiv(Kd, 64);
  int64_t sp = cdiv(1024, tiles);
  int64_t ks = cdiv(cdiv(rows, sp), 16) * 16;
  if (ks < 64) ks = 64;
  sp = cdiv(rows, ks);
  if (sp > 65535) { ks = cdiv(cdiv(rows, 65535), 16) * 16; sp = cdiv(rows, ks); }
  *splits = sp; *ksplit = ks;
  int FP = 1;
  while (FP < F && FP < 256) FP <<= 1;
  int64_t r = cdiv(rows, 512);                     // ~512 blocks, at least 256/FP * 8 rows each
  const int64_t min_rpb = (int64_t)(256 / FP) * 8;
  if (r < min_rpb) r = min_rpb;
  *rpb = r; *bblocks = cdiv(rows, r);
}

// Partial-sum counts of gcrnn_taps_backward_weight: dw_part is [splits][F][KK*G], dbias_part [bias_blocks][F].
extern "C" int gcrnn_taps_backward_weight_parts(int64_t rows, int64_t KK, int64_t G, int64_t F, int64_t* splits, int64_t* bias_blocks) {
  if (!splits || !bias_blocks) return GCRNN_ERR_NULL_POINTER;
  if (!tap_shape_ok(rows, KK, G, F)) return GCRNN_ERR_BAD_SHAPE;
  int64_t ks, rpb;
  bwd_weight_plan(rows, KK, G, F, splits, &ks, bias_blocks, &rpb);
  return GCRNN_OK;
}

template <typename T>
static int taps_bwd_weight(const void* dy, const void* z0, const void* zrest, int64_t zstride, void* dw, void* dbias,
                           double bias_scale, int64_t rows, int64_t KK, int64_t G, int64_t F, void* stream) {
  const int64_t Kd = KK * G;
  int64_t splits, ksplit, bblocks, rpb;
  bwd_weight_plan(rows, KK, G, F, &splits, &ksplit, &bblocks, &rpb);
  BwA<T> a{(const T*)dy, (int)F};
  BwB<T> b{{(const T*)z0, (const T*)zrest, zstride, (int)G}};
  BwC<T> c{(T*)dw, Kd, F * Kd};
  GCRNN_PRE_LAUNCH();
  dim3 grid((unsigned)cdiv(F, 64), (unsigned)cdiv(Kd, 64), (unsigned)splits);
  gemm64_kernel<T><<<grid, 256, 0, as_stream(stream)>>>(a, b, c, F, Kd, rows, ksplit);
  GCRNN_CHECK_LAUNCH();
  if (dbias) {
    int FP = 1;
    while (FP < F && FP < 256) FP <<= 1;
    colsum_kernel<T><<<(unsigned)bblocks, 256, 0, as_stream(stream)>>>((const T*)dy, (T*)dbias, (T)bias_scale, rows, (int)F, FP, rpb);
    GCRNN_CHECK_LAUNCH();
  }
  return GCRNN_OK;
}

extern "C" int gcrnn_taps_backward_weight(int dtype, const void* dy, const void* z0, const void* zrest,
                                          int64_t zstride, void* dw, void* dbias, double bias_scale, int64_t rows,
                                          int64_t KK, int64_t G, int64_t F, void* stream) {
  if (!dy || !z0 || !dw || (KK > 1 && !zrest)) return GCRNN_ERR_NULL_POINTER;
  if (!tap_shape_ok(rows, KK, G, F) || cdiv(F, 64) > 2147483647LL || cdiv(KK * G, 64) > 65535) return GCRNN_ERR_BAD_SHAPE;
  if (dtype == GCRNN_F32) return taps_bwd_weight<float>(dy, z0, zrest, zstride, dw, dbias, bias_scale, rows, KK, G, F, stream);
  if (dtype == GCRNN_F64) return taps_bwd_weight<double>(dy, z0, zrest, zstride, dw, dbias, bias_scale, rows, KK, G, F, stream);
  return GCRNN_ERR_BAD_DTYPE;
}
