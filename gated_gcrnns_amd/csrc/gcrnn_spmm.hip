// Graph shift for graphs beyond LDS (BASELINE configs[4]: N = 1e5, nnz = 1e7): batched CSR row SpMM on node-major data
//     Y[i][n][:] = act( (accumulate ? Y[i][n][:] : 0) + sum_{j in row n} val[j] * X[i][col[j]][:]  [+ bias] )
// replacing the reference's dense x @ S (Utils/graphML.py:116-125), which cannot even hold this graph (1e10 entries).
//
// What bounds it is the GATHER: every non-zero pulls one L-wide row of X through the cache hierarchy (5 GB per hop at
// cfg5 in bf16 against a 51 MB operand that lives in the 256 MB Infinity Cache). So the kernel is built around bytes in
// flight and cache footprint, not arithmetic:
//   * a wave owns one destination row at a time; its 64 lanes are NPW = 64 / LP neighbour slots of LP lanes, each lane
//     moving 16 bytes: one wave instruction fetches a PW = 16 LP byte piece of NPW different neighbour rows, U such
//     instructions are issued back to back before the first FMA (NPW * U independent row fetches in flight per wave,
//     16 waves per CU);
//   * the row's column indices and weights are fetched once with coalesced loads and staged in a wave-private LDS
//     tile (128 entries per trip); the gather addresses are formed from LDS reads (broadcast inside a slot);
//   * the NPW partial sums of a row are folded by a wavefront segmented reduction (xor shuffles over the slot bits) in a
//     fixed order: deterministic, no atomics;
//   * the columns are cut into chunks of PW bytes and the workgroup -> (chunk, row block) map is XCD-aware: all
//     workgroups with the same blockIdx % 8 (one XCD under round-robin dispatch) work on the same column chunk, so the
//     slab an XCD's 4 MiB L2 has to hold is N * PW bytes instead of N * L * s (speed only; nothing depends on it);
//   * optional epilogue: + bias_scale * bias[l % F] and tanh, so the last hop of a Horner step writes h_t directly.
#include "gcrnn_common.h"

namespace {

template <typename T> struct El;
template <> struct El<float> { typedef float acc_t; static constexpr int VE = 4; };
template <> struct El<double> { typedef double acc_t; static constexpr int VE = 2; };
template <> struct El<uint16_t> { typedef float acc_t; static constexpr int VE = 8; };      // bf16 storage, fp32 weights / sums

__device__ __forceinline__ void unpack16(const uint4& v, float (&o)[4], float) {
  o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
}
__device__ __forceinline__ void unpack16(const uint4& v, double (&o)[2], double) {
  o[0] = __longlong_as_double(((unsigned long long)v.y << 32) | v.x);
  o[1] = __longlong_as_double(((unsigned long long)v.w << 32) | v.z);
}
__device__ __forceinline__ void unpack16(const uint4& v, float (&o)[8], uint16_t) {
  const uint32_t p[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(p[i] << 16); o[2 * i + 1] = __uint_as_float(p[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack16(const float (&a)[4], float) {
  return uint4{__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])};
}
__device__ __forceinline__ uint4 pack16(const double (&a)[2], double) {
  const unsigned long long u0 = __double_as_longlong(a[0]), u1 = __double_as_longlong(a[1]);
  return uint4{(uint32_t)u0, (uint32_t)(u0 >> 32), (uint32_t)u1, (uint32_t)(u1 >> 32)};
}
__device__ __forceinline__ uint32_t bf16_rn(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f); }   // v_cvt_pk_bf16_f32: RNE, NaN-safe
__device__ __forceinline__ uint4 pack16(const float (&a)[8], uint16_t) {
  return uint4{bf16_rn(a[0]) | (bf16_rn(a[1]) << 16), bf16_rn(a[2]) | (bf16_rn(a[3]) << 16),
               bf16_rn(a[4]) | (bf16_rn(a[5]) << 16), bf16_rn(a[6]) | (bf16_rn(a[7]) << 16)};
}
__device__ __forceinline__ float act_tanh(float x) {      // 1 - 2 / (1 + exp(2x)) on v_exp / v_rcp: abs error < 3e-7, inf-safe
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ double act_tanh(double x) { return tanh(x); }

constexpr int STAGE = 128;       // CSR entries staged per trip and wave
constexpr int WAVES = 4;         // waves per workgroup

// LP: lanes per neighbour piece (4 .. 64, power of two); U: gather instructions in flight per wave.
template <typename T, int LP, int U, bool TANH>
__global__ __launch_bounds__(64 * WAVES) void spmm_stream_kernel(
    int N, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const typename El<T>::acc_t* __restrict__ val,
    const T* __restrict__ X, T* __restrict__ Y, int64_t L, int nchunks, int rows_per_wave, int RB, int bids_per_slice,
    int accumulate, const typename El<T>::acc_t* __restrict__ bias, typename El<T>::acc_t bias_scale, int F) {
  typedef typename El<T>::acc_t A;
  constexpr int VE = El<T>::VE;
  constexpr int NPW = 64 / LP;
  __shared__ int32_t s_col[WAVES][STAGE];
  __shared__ A s_val[WAVES][STAGE];

  // ---- XCD-aware decode of the block id: x = bid % 8 is the XCD under round-robin dispatch -------------------------
  const int bid = blockIdx.x % bids_per_slice, slice = blockIdx.x / bids_per_slice;
  int chunk, rb;
  if (nchunks <= 8 && (8 % nchunks) == 0) {
    const int m = 8 / nchunks, x = bid & 7, slot = bid >> 3;
    chunk = x % nchunks;
    rb = slot * m + x / nchunks;
  } else if ((nchunks & 7) == 0) {
    const int x = bid & 7, slot = bid >> 3;            // XCD x walks its chunks one after the other
    chunk = x + 8 * (slot / RB);
    rb = slot % RB;
  } else {
    chunk = bid % nchunks;
    rb = bid / nchunks;
  }
  if (rb >= RB) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = lane / LP, pl = lane % LP;
  const int64_t coff = (int64_t)chunk * (LP * VE) + pl * VE;          // this lane's 16 bytes inside a row
  const bool colok = coff < L;                                         // L % VE == 0 (host check): whole vectors only
  const T* Xb = X + (int64_t)slice * N * L + coff;
  T* Yb = Y + (int64_t)slice * N * L + coff;
  int32_t* mycol = s_col[wave];
  A* myval = s_val[wave];

  A bvec[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) bvec[e] = A(0);
  if (TANH && bias && colok) {
#pragma unroll
    for (int e = 0; e < VE; ++e) bvec[e] = bias_scale * bias[(coff + e) % F];
  }

  const int row0 = (rb * WAVES + wave) * rows_per_wave;
  for (int i = 0; i < rows_per_wave; ++i) {
    const int n = row0 + i;
    if (n >= N) break;
    const int beg = __builtin_amdgcn_readfirstlane(rowptr[n]), end = __builtin_amdgcn_readfirstlane(rowptr[n + 1]);
    A acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = A(0);
    uint4 yold = uint4{0u, 0u, 0u, 0u};
    if (accumulate && s == 0 && colok) yold = *reinterpret_cast<const uint4*>(Yb + (int64_t)n * L);      // in flight during the gathers
    for (int c0 = beg; c0 < end; c0 += STAGE) {
      const int cnt = (end - c0 < STAGE) ? end - c0 : STAGE;
      // stage this trip's indices / weights (coalesced), wave-private: the wave's own program order is the only fence needed
      {
        const int i0 = lane, i1 = lane + 64;
        int32_t ca = 0, cb = 0;
        A va = A(0), vb = A(0);
        if (i0 < cnt) { ca = col[c0 + i0]; va = val[c0 + i0]; }
        if (i1 < cnt) { cb = col[c0 + i1]; vb = val[c0 + i1]; }
        mycol[i0] = ca; myval[i0] = va;
        mycol[i1] = cb; myval[i1] = vb;
      }
      for (int j = 0; j < cnt; j += NPW * U) {
        uint4 xv[U];
        A w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = j + u * NPW + s;
          const bool ok = idx < cnt;                       // entries past cnt were staged as (col 0, weight 0)
          const int32_t c = mycol[ok ? idx : 0];
          w[u] = ok ? myval[idx] : A(0);
          xv[u] = uint4{0u, 0u, 0u, 0u};
          if (ok && colok) xv[u] = *reinterpret_cast<const uint4*>(Xb + (int64_t)c * L);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          A xe[VE];
          unpack16(xv[u], xe, T());
#pragma unroll
          for (int e = 0; e < VE; ++e) acc[e] += w[u] * xe[e];
        }
      }
    }
    // wavefront segmented reduction: fold the NPW neighbour slots (lanes that differ only in the slot bits), fixed order
#pragma unroll
    for (int off = LP; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += __shfl_xor(acc[e], off, 64);
    }
    if (s == 0 && colok) {
      if (accumulate) {
        A ye[VE];
        unpack16(yold, ye, T());
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += ye[e];
      }
      if (TANH) {
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] = act_tanh(acc[e] + bvec[e]);
      }
      *reinterpret_cast<uint4*>(Yb + (int64_t)n * L) = pack16(acc, T());
    }
  }
}

template <typename T, int LP, int U>
int launch_lp(int N, const int32_t* rowptr, const int32_t* col, const void* val, const void* X, void* Y, int64_t L, int64_t nbatch,
              int accumulate, const void* bias, double bias_scale, int F, int act, int rows_per_wave, hipStream_t st) {
  typedef typename El<T>::acc_t A;
  constexpr int VE = El<T>::VE;
  const int64_t chunk_elems = (int64_t)LP * VE;
  const int nchunks = (int)cdiv(L, chunk_elems);
  const int RB = (int)cdiv(N, (int64_t)WAVES * rows_per_wave);
  int64_t per_slice;
  if (nchunks <= 8 && (8 % nchunks) == 0) per_slice = cdiv(RB, 8 / nchunks) * 8;
  else if ((nchunks & 7) == 0) per_slice = (int64_t)nchunks * RB;
  else per_slice = (int64_t)nchunks * RB;
  const int64_t grid = per_slice * nbatch;
  if (grid <= 0 || grid > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  if (act)
    spmm_stream_kernel<T, LP, U, true><<<(unsigned)grid, 64 * WAVES, 0, st>>>(N, rowptr, col, (const A*)val, (const T*)X, (T*)Y, L, nchunks,
                                                                              rows_per_wave, RB, (int)per_slice, accumulate, (const A*)bias, (A)bias_scale, F);
  else
    spmm_stream_kernel<T, LP, U, false><<<(unsigned)grid, 64 * WAVES, 0, st>>>(N, rowptr, col, (const A*)val, (const T*)X, (T*)Y, L, nchunks,
                                                                               rows_per_wave, RB, (int)per_slice, accumulate, (const A*)bias, (A)bias_scale, F);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

template <typename T>
int launch_t(int N, const int32_t* rowptr, const int32_t* col, const void* val, const void* X, void* Y, int64_t L, int64_t nbatch,
             int accumulate, const void* bias, double bias_scale, int F, int act, int piece_lanes, int unroll, int rows_per_wave,
             hipStream_t st) {
  constexpr int VE = El<T>::VE;
  if (L % VE) return GCRNN_ERR_UNSUPPORTED;
  if (piece_lanes <= 0) {
    // auto: the widest piece that does not overshoot the row (fewest wave instructions per gathered byte); rows wider than
    // one wave instruction (1 KiB) are cut into 1 KiB chunks
    const int64_t lanes = L / VE;
    piece_lanes = 64;
    while (piece_lanes > 4 && piece_lanes / 2 >= lanes) piece_lanes >>= 1;
    // An operand far beyond the XCDs' L2 (4 MiB each) is gathered from the Infinity Cache at a rate set by 128-byte line
    // requests (measured at cfg5, N = 1e5, nnz = 1e7, bf16: 512 / 256 / 128 / 64-byte pieces -> 7.8 / 8.2 / 8.9 / 4.5 TB/s):
    // 128-byte pieces keep whole lines and give an XCD the smallest slab (N x 128 B) to find again in its L2.
    if ((int64_t)N * L * (16 / VE) > (8LL << 20) && piece_lanes > 8) piece_lanes = 8;
  }
  if (unroll <= 0) unroll = (piece_lanes >= 8) ? 8 : 4;
  if (rows_per_wave <= 0) rows_per_wave = 4;
#define GCRNN_SPMM_CASE(LPV, UV)                                                                                             \
  if (piece_lanes == LPV && unroll == UV)                                                                                    \
    return launch_lp<T, LPV, UV>(N, rowptr, col, val, X, Y, L, nbatch, accumulate, bias, bias_scale, F, act, rows_per_wave, st);
  GCRNN_SPMM_CASE(64, 8) GCRNN_SPMM_CASE(64, 4)
  GCRNN_SPMM_CASE(32, 8) GCRNN_SPMM_CASE(32, 4)
  GCRNN_SPMM_CASE(16, 8) GCRNN_SPMM_CASE(16, 4)
  GCRNN_SPMM_CASE(8, 8) GCRNN_SPMM_CASE(8, 4) GCRNN_SPMM_CASE(8, 2)
  GCRNN_SPMM_CASE(4, 8) GCRNN_SPMM_CASE(4, 4) GCRNN_SPMM_CASE(4, 2)
#undef GCRNN_SPMM_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int gcrnn_spmm_ex(int dtype, int64_t N, const int32_t* rowptr, const int32_t* col, const void* val, const void* X,
                             void* Y, int64_t L, int64_t nbatch, int accumulate, const void* bias, double bias_scale, int64_t F,
                             int act, int piece_lanes, int unroll, int rows_per_wave, void* stream) {
  if (!rowptr || !X || !Y) return GCRNN_ERR_NULL_POINTER;
  if (N <= 0 || L <= 0 || nbatch <= 0 || N > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (X == Y) return GCRNN_ERR_UNSUPPORTED;                  // a hop cannot run in place
  if (act != 0 && act != 1) return GCRNN_ERR_UNSUPPORTED;    // 0: none, 1: tanh(. + bias_scale * bias[l % F])
  if (act && bias && (F <= 0 || L % F)) return GCRNN_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) % 16) return GCRNN_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (dtype == GCRNN_F32)
    return launch_t<float>((int)N, rowptr, col, val, X, Y, L, nbatch, accumulate, bias, bias_scale, (int)F, act, piece_lanes, unroll, rows_per_wave, st);
  if (dtype == GCRNN_F64)
    return launch_t<double>((int)N, rowptr, col, val, X, Y, L, nbatch, accumulate, bias, bias_scale, (int)F, act, piece_lanes, unroll, rows_per_wave, st);
  if (dtype == GCRNN_BF16)
    return launch_t<uint16_t>((int)N, rowptr, col, val, X, Y, L, nbatch, accumulate, bias, bias_scale, (int)F, act, piece_lanes, unroll, rows_per_wave, st);
  return GCRNN_ERR_BAD_DTYPE;
}
