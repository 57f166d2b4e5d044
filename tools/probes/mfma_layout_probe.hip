// Standalone probe of the 16x16x4 fp64 / fp32 MFMA register layouts on gfx950:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_layout_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct V4;
template <> struct V4<double> { typedef d4 type; };
template <> struct V4<float> { typedef f4 type; };
__device__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// assumed: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k; output dumped raw: out[lane][r]
template <typename T>
__global__ void probe(const T* A, const T* B, T* out) {
  const int l = threadIdx.x;
  typename V4<T>::type c = {0, 0, 0, 0};
  c = mfma(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <typename T>
void run(const char* name) {
  T hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (T)((i * 7 % 13) - 6); hB[i] = (T)((i * 5 % 11) - 5); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { T s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  T *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  probe<T><<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  // candidate output layouts
  int okA = 1, okB = 1, okC = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const T v = hD[l * 4 + r];
    if (v != ref[(4 * (l >> 4) + r) * 16 + (l & 15)]) okA = 0;      // row = 4 (l / 16) + r, col = l % 16
    if (v != ref[((l >> 4) + 4 * r) * 16 + (l & 15)]) okB = 0;      // row = l / 16 + 4 r,   col = l % 16
    if (v != ref[(l & 15) * 16 + 4 * (l >> 4) + r]) okC = 0;        // row = l % 16,          col = 4 (l / 16) + r
  }
  printf("%s: layout A (row = 4*(l/16) + r, col = l%%16): %d | B (row = l/16 + 4 r): %d | C (transposed A): %d\n", name, okA, okB, okC);
}
int main() { run<double>("f64 16x16x4"); run<float>("f32 16x16x4"); return 0; }
