// Calibration of the WRITE_SIZE counter for the state stores of the sequence-resident kernels:
// known-byte stores in the kernels' access patterns, so that rocprofv3 --pmc WRITE_SIZE can be compared with the bytes actually stored.
//   pattern 0: full lines   -- every lane 16 B, a wave covers 1 KiB contiguous (the reference point)
//   pattern 1: 64-byte row pieces of 128-byte rows (the wide kernel: lane (r, q) stores 16 B at row r * 128 + q * 16), the other half of
//              every row written by a SECOND pass of the same launch ~N rows later (two pieces of a line, far apart in time)
//   pattern 2: 32-byte row pieces (round 3's kernel: 8 B per lane, four passes per row)
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/write_size_probe.hip -o /tmp/wsp && rocprofv3 --pmc WRITE_SIZE --kernel-trace
//   --output-format csv -d /tmp/wsp_out -- /tmp/wsp   (tools/write_size_calibration.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void full_lines(uint4* out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = uint4{1u, 2u, 3u, (unsigned)i};
}
// rows of 128 B; pass p writes bytes [p * 64, p * 64 + 64) of every row of this workgroup's slab, pass after pass
__global__ void pieces64(char* out, size_t rows) {
  const size_t per = rows / gridDim.x, r0 = (size_t)blockIdx.x * per;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  for (int pass = 0; pass < 2; ++pass)
    for (size_t t = wave; t * 16 < per; t += blockDim.x / 64)
      *reinterpret_cast<uint4*>(out + (r0 + t * 16 + r) * 128 + pass * 64 + q * 16) = uint4{1u, 2u, 3u, (unsigned)t};
}
__global__ void pieces32(char* out, size_t rows) {
  const size_t per = rows / gridDim.x, r0 = (size_t)blockIdx.x * per;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  for (int pass = 0; pass < 4; ++pass)
    for (size_t t = wave; t * 16 < per; t += blockDim.x / 64)
      *reinterpret_cast<uint2*>(out + (r0 + t * 16 + r) * 128 + pass * 32 + q * 8) = uint2{1u, (unsigned)t};
}

int main() {
  const size_t bytes = (size_t)1 << 30;      // 1 GiB per pattern: far beyond L2 + Infinity Cache
  char* buf;
  if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
  const size_t rows = bytes / 128;
  for (int rep = 0; rep < 2; ++rep) {
    full_lines<<<256, 512>>>(reinterpret_cast<uint4*>(buf), bytes / 16);
    pieces64<<<256, 512>>>(buf, rows);
    pieces32<<<256, 512>>>(buf, rows);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("stored per launch: %zu bytes = %zu KiB (each of full_lines, pieces64, pieces32)\n", bytes, bytes / 1024);
  return 0;
}
