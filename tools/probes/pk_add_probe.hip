// Issue cost of v_pk_add_f32 on gfx950 at one and two waves per SIMD, chains as in the hop stream (two accumulator pairs, four
// dependent adds each per trip) and as a tree. Measured: 8.5 cycles per instruction for a wave alone, 4.5 per SIMD with two waves; the
// shape of the dependence chains does not matter.   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_add_probe tools/probes/pk_add_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  f32x2 a0 = {1.f, 2.f}, a1 = {3.f, 4.f};
  f32x2 r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = f32x2{(float)threadIdx.x + i, (float)i * 0.5f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {        // 8 v_pk_add_f32: two chains of four
      asm volatile(
          "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
          "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\t"
          "v_pk_add_f32 %0, %0, %6\n\tv_pk_add_f32 %1, %1, %7\n\t"
          "v_pk_add_f32 %0, %0, %8\n\tv_pk_add_f32 %1, %1, %9\n\t"
          : "+v"(a0), "+v"(a1) : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
    } else {                // 8 v_pk_add_f32 as a tree: four independent, two, one, one (same count, shorter chains)
      f32x2 s0, s1, s2, s3;
      asm volatile(
          "v_pk_add_f32 %2, %6, %8\n\tv_pk_add_f32 %3, %7, %9\n\t"
          "v_pk_add_f32 %4, %10, %12\n\tv_pk_add_f32 %5, %11, %13\n\t"
          "v_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %5\n\t"
          "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
          : "+v"(a0), "+v"(a1), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
          : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y + a1.x + a1.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  const char* names[2] = {"8 v_pk_add_f32 (2 chains of 4)", "8 v_pk_add_f32 (tree)"};
  for (int threads : {256, 512}) {
    for (int mode = 0; mode < 2; ++mode) {
      long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) probe<0><<<256, threads>>>(out, cyc, iters);
        if (mode == 1) probe<2><<<256, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
      }
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%d waves/SIMD  %-34s %.1f s_memtime ticks per trip per wave\n", threads / 256, names[mode], (double)h / iters);
    }
  }
  return 0;
}
