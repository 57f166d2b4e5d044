// Micro-benchmark behind DESIGN 4.1 (r2, "x from the user layout"): how fast can ONE workgroup per CU pull the x operand of an item
// (128 KB) from L2 / HBM  (A) as MFMA B fragments from the sequence-major layout [n][64] (16 B per lane, 128-byte rows, the
// pattern of the step kernel's phase 1), (B) as coalesced 16-byte loads of the user layout [64][N] written to LDS,
// (C) the same through LDS-DMA (global_load_lds_dwordx4)?   Build: hipcc --offload-arch=gfx950 -O3 -o xload_probe xload_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NP = 1024, G = 64, THREADS = 512, ITEM_BYTES = NP * G * 2;

template <int MODE>
__global__ __launch_bounds__(THREADS) void probe(const uint4* __restrict__ x, float* __restrict__ sink, int items, int slots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc = 0.f;
  for (int it = blockIdx.x; it < items; it += slots) {
    const uint4* src = x + (size_t)it * (ITEM_BYTES / 16);
    if (MODE == 0) {
      // fragments: wave owns 8 tiles of 16 nodes; lane (r = node in tile, q = 16-byte piece) reads 2 k-steps (pieces q, q + 4) of its node's 128-byte row
      uint4 v[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int node = (wave * 8 + i) * 16 + (lane & 15);
#pragma unroll
        for (int s = 0; s < 2; ++s) v[2 * i + s] = src[node * 8 + (lane >> 4) + 4 * s];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += __uint_as_float(v[i].x ^ v[i].y ^ v[i].z ^ v[i].w);
    } else if (MODE == 1) {
      uint4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = src[i * THREADS + tid];
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[i * THREADS + tid] = v[i];
      __syncthreads();
      acc += __uint_as_float(lds[(tid * 37) & 8191].x);
      __syncthreads();
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * THREADS + wave * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(lds + i * THREADS + wave * 64), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      acc += __uint_as_float(lds[(tid * 37) & 8191].x);
      __syncthreads();
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const int items = 2048, slots = 256;            // 256 MB working set: mostly HBM; the second run of each mode re-reads a 64-item set from L2 / MALL
  uint4* x; float* sink;
  CK(hipMalloc(&x, (size_t)items * ITEM_BYTES)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(x, 1, (size_t)items * ITEM_BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = 150 * 1024;                  // one workgroup per CU, as in the step kernel
  auto run = [&](int mode, int n_items, const char* what) -> int {
    void (*k)(const uint4*, float*, int, int) = mode == 0 ? probe<0> : mode == 1 ? probe<1> : probe<2>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; ++rep) k<<<slots, THREADS, lds>>>(x, sink, n_items, slots);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) k<<<slots, THREADS, lds>>>(x, sink, n_items, slots);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_item = 1e3 * ms / reps / (n_items / (double)slots);
    printf("%-52s %4d items: %7.2f us per item-slot, %6.1f GB/s per CU, %5.2f TB/s chip\n", what, n_items, us_item,
           ITEM_BYTES / us_item * 1e-3, ITEM_BYTES / us_item * 1e-6 * slots);
    return 0;
  };
  for (int n : {2048, 256}) {
    if (run(0, n, "A fragments from [n][64] (16 B / lane, 128-B rows)")) return 1;
    if (run(1, n, "B coalesced 16-B loads -> registers -> LDS")) return 1;
    if (run(2, n, "C LDS-DMA global_load_lds_dwordx4")) return 1;
  }
  return 0;
}
