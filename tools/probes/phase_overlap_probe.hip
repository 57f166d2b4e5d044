// Does a second, independent workgroup on the CU overlap the step kernel's phases? (budget of the two-workgroups-per-CU design)
// Synthetic item = phase A (pull `a_kb` KiB of 128-byte rows from an L2-resident buffer into registers, as the operand fragments are),
// phase B (`trips` trips of the hop stream's shape: 4 dependent-address ds_read_b128 gathers + 8 v_pk_add_f32 per trip, per wave),
// phase C (store `c_kb` KiB in 32-byte pieces of separate lines, as the sequence-major state is). Variant 1: one 512-thread workgroup
// per CU (150 KiB of LDS) doing `items` items; variant 2: two 256-thread workgroups per CU (75 KiB each), each doing `items` items with
// the same A, B with the same trips per wave on a half-size image (so the CU does twice the A work -- the 8x operand redundancy -- and
// the same B work), half of C.   hipcc --offload-arch=gfx950 -O3 -o /tmp/pop tools/probes/phase_overlap_probe.hip && /tmp/pop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const u32x4* __restrict__ src, u32x4* __restrict__ dst, float* __restrict__ out,
                                                 int items, int a_loads, int trips, int c_stores, int lds_rows, int phases) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* img = reinterpret_cast<f32x4*>(smem);                 // [lds_rows][4] x 16 B = 64-byte rows
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < lds_rows * 4; i += THREADS) img[i] = f32x4{1.f, 2.f, 3.f, (float)i};
  __syncthreads();
  f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
  unsigned sum = 0;
  unsigned idx = (unsigned)(tid * 2654435761u);
  const size_t wg = blockIdx.x;
  for (int it = 0; it < items; ++it) {
    if (phases & 1) {                                          // A: a_loads x 16 B per lane, 4 lanes per 64-byte piece of a 128-byte row
      const u32x4* base = src + ((wg * 131 + it * 17) % 8) * 32768;      // eight 512-KiB windows shared by every workgroup: L2-resident
      u32x4 v[16];
      for (int l0 = 0; l0 < a_loads; l0 += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const unsigned e = (unsigned)((l0 + u) * THREADS + tid);     // 4 lanes take one 64-byte half of a 128-byte row
          v[u] = base[((e >> 2) * 8 + (e & 3)) & 32767u];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += v[u][0] ^ v[u][3];
      }
    }
    if (phases & 2) {                                          // B: the hop stream's trip
      for (int t = 0; t < trips; ++t) {
        f32x4 r0 = img[(idx & (unsigned)(lds_rows * 4 - 1))];
        f32x4 r1 = img[((idx >> 3) & (unsigned)(lds_rows * 4 - 1))];
        f32x4 r2 = img[((idx >> 7) & (unsigned)(lds_rows * 4 - 1))];
        f32x4 r3 = img[((idx >> 11) & (unsigned)(lds_rows * 4 - 1))];
        acc0 += f32x2{r0[0], r0[1]}; acc1 += f32x2{r0[2], r0[3]};
        acc0 += f32x2{r1[0], r1[1]}; acc1 += f32x2{r1[2], r1[3]};
        acc0 += f32x2{r2[0], r2[1]}; acc1 += f32x2{r2[2], r2[3]};
        acc0 += f32x2{r3[0], r3[1]}; acc1 += f32x2{r3[2], r3[3]};
        idx = idx * 1664525u + 1013904223u + (unsigned)__float_as_uint(r3[3]) % 3u;      // the next addresses depend on the data
      }
      __syncthreads();
    }
    if (phases & 4) {                                          // C: 8-byte pieces, 4 lanes per 32-byte piece, one piece per 128-byte line
      u32x4* dbase = dst + ((wg * 37 + it) % 2048) * 2048;
      for (int s = 0; s < c_stores; ++s) {
        const size_t line = (size_t)(s * (THREADS / 4) + (tid >> 2));
        reinterpret_cast<uint2*>(dbase)[line * 16 + (tid & 3)] = make_uint2(sum + s, idx);
      }
    }
  }
  out[wg * THREADS + tid] = acc0[0] + acc0[1] + acc1[0] + acc1[1] + (float)sum;
}

static float run(int threads, int wgs, int lds_bytes, const u32x4* src, u32x4* dst, float* out, int items, int a_loads, int trips,
                 int c_stores, int lds_rows, int phases) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if (threads == 512) { hipFuncSetAttribute((const void*)probe<512>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      probe<512><<<wgs, 512, lds_bytes>>>(src, dst, out, items, a_loads, trips, c_stores, lds_rows, phases); }
    else { hipFuncSetAttribute((const void*)probe<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      probe<256><<<wgs, 256, lds_bytes>>>(src, dst, out, items, a_loads, trips, c_stores, lds_rows, phases); }
  };
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  u32x4 *src, *dst; float* out;
  const size_t win = 8ull * 32768 * 16;                        // 4 MiB of source windows
  hipMalloc(&src, win); hipMalloc(&dst, 2048ull * 2048 * 16 + (4u << 20)); hipMalloc(&out, 512 * 512 * 4);
  hipMemset(src, 1, win);
  const int items = 16;
  // variant 1: 512 threads, A = 256 KiB (32 loads of 16 B per lane), B = 92 trips per wave, C = 32 KiB (8 stores of 8 B per lane); LDS 150 KiB
  // variant 2: 256 threads x 2 per CU, A = 256 KiB (64 loads per lane), B = 92 trips per wave (half image), C = 16 KiB (8 stores per lane)
  const char* names[4] = {"A+B+C", "A only", "B only", "C only"};
  const int ph[4] = {7, 1, 2, 4};
  // (hipcc's loop for B is not the hand-pipelined stream: 92 trips take 16.6 us here against 10.7 us in the step kernel; 60 trips give
  //  B the weight it has there)
  for (int trips : {92, 60}) {
    printf("B = %d trips per wave\n", trips);
    for (int p = 0; p < 4; ++p) {
      const float t1 = run(512, 256, 150 * 1024, src, dst, out, items, 32, trips, 8, 1024, ph[p]);
      const float t2 = run(256, 512, 75 * 1024, src, dst, out, items, 64, trips, 8, 512, ph[p]);
      printf("%-6s  one 512-thread workgroup per CU: %7.1f us per item   two 256-thread workgroups per CU: %7.1f us per item pair\n",
             names[p], t1 / items, t2 / items);
    }
  }
  return 0;
}
