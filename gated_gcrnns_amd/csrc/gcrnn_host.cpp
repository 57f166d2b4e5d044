// Host-side entry points of libgcrnn_hip.so: version/status strings and graph preparation
// (dense GSO -> CSR, degree ordering). No GPU is touched here, so these run in CPU-only tests.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <vector>
#include "../../include/gcrnn.h"

extern "C" int gcrnn_version(void) { return 100; }  // 0.1.0

extern "C" const char* gcrnn_status_string(int status) {
  switch (status) {
    case GCRNN_OK: return "ok";
    case GCRNN_ERR_BAD_DTYPE: return "unsupported dtype for this entry point";
    case GCRNN_ERR_BAD_SHAPE: return "bad shape (zero/negative size or a grid limit exceeded)";
    case GCRNN_ERR_NULL_POINTER: return "null pointer for a required argument";
    case GCRNN_ERR_UNSUPPORTED: return "unsupported argument combination";
    case GCRNN_ERR_LAUNCH: return "HIP kernel launch failed";
    case GCRNN_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

// last HIP runtime error seen by a launch check (diagnostics for GCRNN_ERR_LAUNCH)
static thread_local char g_last_hip_error[256] = "";
extern "C" void gcrnn_note_hip_error(int code, const char* what) {
  snprintf(g_last_hip_error, sizeof(g_last_hip_error), "hip error %d: %s", code, what ? what : "?");
}
extern "C" const char* gcrnn_last_hip_error(void) { return g_last_hip_error; }

static inline bool keep(double v, double tol) { return std::fabs(v) > tol; }

// entry (i, j) of the operator whose CSR we build: S^T[i][j] = S[j][i]; identity optionally added
static inline double entry(const double* S, int64_t N, int transpose, int add_identity, int64_t i, int64_t j) {
  double v = transpose ? S[j * N + i] : S[i * N + j];
  if (add_identity && i == j) v += 1.0;
  return v;
}

extern "C" int gcrnn_csr_count(const double* S, int64_t N, int transpose, int add_identity, double tol, int64_t* nnz) {
  if (!S || !nnz) return GCRNN_ERR_NULL_POINTER;
  if (N <= 0 || tol < 0) return GCRNN_ERR_BAD_SHAPE;
  int64_t c = 0;
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = 0; j < N; ++j) c += keep(entry(S, N, transpose, add_identity, i, j), tol);
  *nnz = c;
  return GCRNN_OK;
}

extern "C" int gcrnn_csr_fill(const double* S, int64_t N, int transpose, int add_identity, double tol, int32_t* rowptr,
                              int32_t* col, double* val) {
  if (!S || !rowptr) return GCRNN_ERR_NULL_POINTER;
  if (N <= 0 || N > 2147483647LL || tol < 0) return GCRNN_ERR_BAD_SHAPE;
  int64_t c = 0;
  rowptr[0] = 0;
  for (int64_t i = 0; i < N; ++i) {
    for (int64_t j = 0; j < N; ++j) {
      const double v = entry(S, N, transpose, add_identity, i, j);
      if (keep(v, tol)) {
        if (c >= 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
        if (col) col[c] = (int32_t)j;
        if (val) val[c] = v;
        ++c;
      }
    }
    rowptr[i + 1] = (int32_t)c;
  }
  return GCRNN_OK;
}

extern "C" int gcrnn_degree_order(const int32_t* rowptr, int64_t N, int32_t* order) {
  if (!rowptr || !order) return GCRNN_ERR_NULL_POINTER;
  if (N <= 0 || N > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  std::iota(order, order + N, 0);
  std::stable_sort(order, order + N, [rowptr](int32_t a, int32_t b) {
    return (rowptr[a + 1] - rowptr[a]) > (rowptr[b + 1] - rowptr[b]);
  });
  return GCRNN_OK;
}

// ---- sliced ELL for the fused kernels ---------------------------------------------------------
// Nodes are renumbered by `order` (position p holds original node order[p]); tiles of `tile` positions.
// Tile t stores deg_t = max degree in the tile rounded up to `pad` entries per position, laid out
// [entry][position-in-tile] so that one wave reads 16 consecutive (col, val) pairs per entry.
// colpos = neighbour's POSITION (renumbered); padding entries are (0, 0.0).
static int ell_tile_deg(const int32_t* rowptr, const int32_t* order, int64_t N, int64_t t, int tile, int pad) {
  int d = 0;
  for (int r = 0; r < tile; ++r) {
    const int64_t p = t * tile + r;
    if (p < N) {
      const int32_t n = order ? order[p] : (int32_t)p;
      d = std::max(d, rowptr[n + 1] - rowptr[n]);
    }
  }
  return (d + pad - 1) / pad * pad;
}

extern "C" int gcrnn_ell_size(const int32_t* rowptr, int64_t N, const int32_t* order, int tile, int pad,
                              int64_t ntiles, int64_t* nentries) {
  if (!rowptr || !nentries) return GCRNN_ERR_NULL_POINTER;
  if (N <= 0 || tile <= 0 || pad <= 0 || ntiles * tile < N) return GCRNN_ERR_BAD_SHAPE;
  int64_t tot = 0;
  for (int64_t t = 0; t < ntiles; ++t) tot += ell_tile_deg(rowptr, order, N, t, tile, pad);
  *nentries = tot;   // in units of `tile` (col,val) pairs
  return GCRNN_OK;
}

extern "C" int gcrnn_ell_fill(const int32_t* rowptr, const int32_t* col, const double* val, int64_t N,
                              const int32_t* order, int tile, int pad, int64_t ntiles, int32_t* tile_off,
                              int32_t* ell_col, float* ell_val) {
  if (!rowptr || !col || !val || !tile_off || !ell_col || !ell_val) return GCRNN_ERR_NULL_POINTER;
  if (N <= 0 || tile <= 0 || pad <= 0 || ntiles * tile < N) return GCRNN_ERR_BAD_SHAPE;
  std::vector<int32_t> pos(N);
  for (int64_t p = 0; p < N; ++p) pos[order ? order[p] : p] = (int32_t)p;
  int64_t off = 0;
  for (int64_t t = 0; t < ntiles; ++t) {
    tile_off[t] = (int32_t)off;
    const int d = ell_tile_deg(rowptr, order, N, t, tile, pad);
    for (int e = 0; e < d; ++e)
      for (int r = 0; r < tile; ++r) {
        const int64_t p = t * tile + r;
        int32_t c = 0;
        float v = 0.f;
        if (p < N) {
          const int32_t n = order ? order[p] : (int32_t)p;
          const int32_t j = rowptr[n] + e;
          if (j < rowptr[n + 1]) { c = pos[col[j]]; v = (float)val[j]; }
        }
        ell_col[(off + e) * tile + r] = c;
        ell_val[(off + e) * tile + r] = v;
      }
    off += d;
    if (off > 2147483647LL / tile) return GCRNN_ERR_BAD_SHAPE;
  }
  tile_off[ntiles] = (int32_t)off;
  return GCRNN_OK;
}
