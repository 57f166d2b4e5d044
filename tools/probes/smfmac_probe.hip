// Standalone probe of v_smfmac_f32_16x16x64_bf16 (2:4 structured-sparse A) on gfx950: which (lane, element) of the dense B operand
// (16 bf16 per lane) is which K index and which output column, and how the compressed A operand (8 bf16 per lane) and the index
// register map onto dense K positions. Used to decide whether the one-hot hop-sum trick can take FOUR gathered
// neighbour rows per matrix instruction instead of two.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/smfmac_probe.hip -o /tmp/smfmac_probe && /tmp/smfmac_probe
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) __bf16 bf16x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// A compressed: lane l, element e carries the value aval[l][e]; B dense: bval[l][e]; idx per lane
__global__ void probe(const float* aval, const float* bval, const int* idx, float* out) {
  const int l = threadIdx.x;
  bf16x8 a;
  bf16x16 b;
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)aval[l * 8 + e];
  for (int e = 0; e < 16; ++e) b[e] = (__bf16)bval[l * 16 + e];
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, c, idx[l], 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

int main() {
  float *dA, *dB, *dD; int* dI;
  hipMalloc(&dA, 64 * 8 * 4); hipMalloc(&dB, 64 * 16 * 4); hipMalloc(&dD, 256 * 4); hipMalloc(&dI, 64 * 4);
  std::vector<float> hA(64 * 8), hB(64 * 16), hD(256);
  std::vector<int> hI(64);
  // hypothesis for the compressed A: lane (i = l & 15, sg = l >> 4) holds compressed slots s = 8 sg + e of row i; value = s + 1
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) hA[l * 8 + e] = (float)(8 * (l >> 4) + e + 1);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  const int idxs[4] = {0x44444444, (int)0xEEEEEEEE, (int)0x88888888 /* (0,2) */, (int)0xDDDDDDDD /* (1,3) */};
  for (int t = 0; t < 4; ++t) {
    for (int l = 0; l < 64; ++l) hI[l] = idxs[t];
    hipMemcpy(dI, hI.data(), 64 * 4, hipMemcpyHostToDevice);
    printf("== idx pattern 0x%08x: for each B (lane, elem): value seen in D column(s) [all 16 rows should agree] ==\n", idxs[t]);
    for (int lb = 0; lb < 64; ++lb) {
      printf("B lane %2d:", lb);
      for (int eb = 0; eb < 16; ++eb) {
        std::fill(hB.begin(), hB.end(), 0.f);
        hB[lb * 16 + eb] = 1.f;
        hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dI, dD);
        hipMemcpy(hD.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        // find nonzero outputs: assume D layout of the dense 16x16 MFMA: lane l, reg r -> row 4 (l >> 4) + r, col l & 15
        int col = -1; float v = 0.f; int nz = 0; bool same = true;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hD[l * 4 + r] != 0.f) {
          if (nz == 0) { col = l & 15; v = hD[l * 4 + r]; } else if ((l & 15) != col || hD[l * 4 + r] != v) same = false;
          ++nz;
        }
        if (nz == 0) printf(" [--]");
        else printf(" [c%d v%g n%d%s]", col, v, nz, same ? "" : "!");
      }
      printf("\n");
    }
  }
  // second experiment: row dependence of A -- value = row + 1 for every slot, B = all ones, idx (0,1): D[i][j] = 32 (i + 1)?
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) hA[l * 8 + e] = (float)((l & 15) + 1);
  for (auto& x : hB) x = 1.f;
  for (int l = 0; l < 64; ++l) hI[l] = 0x44444444;
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dI, hI.data(), 64 * 4, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dI, dD);
  hipMemcpy(hD.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
  printf("== A value = (lane & 15) + 1, B = ones: D[lane][reg] ==\n");
  for (int l = 0; l < 64; ++l) printf("l%2d: %g %g %g %g\n", l, hD[l * 4], hD[l * 4 + 1], hD[l * 4 + 2], hD[l * 4 + 3]);
  return 0;
}
