// Shared helpers for the gfx950 kernels of libgcrnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/gcrnn.h"

// hipGetLastError() is per-thread and sticky until read: other runtime users (PyTorch) leave benign
// codes behind, so clear it before a launch and read it right after.
#define GCRNN_PRE_LAUNCH() (void)hipGetLastError()
extern "C" void gcrnn_note_hip_error(int code, const char* what);
#define GCRNN_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) {                                   \
      gcrnn_note_hip_error((int)e__, hipGetErrorString(e__));  \
      return GCRNN_ERR_LAUNCH;                                 \
    }                                                          \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Workgroups of a persistent launch of the sequence-resident kernels: one per compute unit, read from the device ONCE (256 on MI355X;
// the same number where no device answers -- the CPU-side queries of the tests).
static inline int gcrnn_persistent_grid() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    else cus = 256;
  }
  return cus;
}
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 16-byte vector of T (4 x f32 or 2 x f64)
template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float4 type; static constexpr int n = 4; };
template <> struct Vec16<double> { typedef double2 type; static constexpr int n = 2; };
