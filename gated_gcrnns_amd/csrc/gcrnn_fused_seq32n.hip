// The node-gated cell's two state-size passes on the wide sequence-resident kernel (round 5; gcrnn_fused_seq32.h modes 3 and 4): the
// filter-output pass A(S) x_t + b over all (t, b) items and the recurrence with per-node gates in its epilogue. Until round 4 both ran
// round 3's 16-feature kernel (1.45 + 1.75 ms per forward at the bench size). Reference: Utils/graphML.py:2379-2407, 2420-2423.
#include "gcrnn_fused_step.h"
#define GCRNN_SEQ32_STAMP_READER_NAME gcrnn_debug_read_seq32n_stamps
#include "gcrnn_fused_seq32.h"

static bool seq32n_wanted(int64_t rounds_of) {
  const char* off = getenv("GCRNN_SEQ32");
  if (off && off[0] == '0') return false;
  const char* offn = getenv("GCRNN_SEQ32_NODE");      // =0: the node-gated cell keeps round 3's kernels for these passes (A/B)
  if (offn && offn[0] == '0') return false;
  const char* mb = getenv("GCRNN_SEQ32_MIN_B");
  if (mb) return rounds_of >= (atoi(mb) < 1 ? 1 : atoi(mb));
  return rounds_of >= 129;      // (as the un-gated forward: one workgroup per sequence pays from three rounds of the chunk-parallel kernel)
}

template <int K, int HS, int XS, int VAR, int MODE>
static int seq32n_launch(const Seq32Args& sa, hipStream_t st) {
  const size_t lds = Seq32Map<K, HS, XS>::lds_bytes(sa.entries, (VAR & 1) != 0, false);
  if (!lds) return GCRNN_ERR_UNSUPPORTED;
  auto sk = fused_seq32_kernel<K, HS, XS, VAR, MODE>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  sk<<<(unsigned)(sa.B < gcrnn_persistent_grid() ? sa.B : gcrnn_persistent_grid()), STHREADS, lds, st>>>(sa);      // one workgroup per CU (count read from the device once)
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

template <int K, int HS, int XS>
static size_t seq32n_lds(int64_t entries, bool pack) { return Seq32Map<K, HS, XS>::lds_bytes(entries, pack, false); }

static size_t seq32n_lds_for(int64_t F, int64_t G, int64_t K, int64_t entries, bool pack = false) {
#define GCRNN_SEQ32_CASE(KK, HH, XX) if (K == KK && F == 32 * HH && G == 32 * XX) return seq32n_lds<KK, HH, XX>(entries, pack);
  GCRNN_SEQ32_CASE(5, 2, 2) GCRNN_SEQ32_CASE(4, 2, 2) GCRNN_SEQ32_CASE(3, 2, 2) GCRNN_SEQ32_CASE(2, 2, 2)
  GCRNN_SEQ32_CASE(5, 2, 1) GCRNN_SEQ32_CASE(4, 2, 1) GCRNN_SEQ32_CASE(3, 2, 1) GCRNN_SEQ32_CASE(2, 2, 1)
  GCRNN_SEQ32_CASE(5, 1, 1) GCRNN_SEQ32_CASE(4, 1, 1) GCRNN_SEQ32_CASE(3, 1, 1) GCRNN_SEQ32_CASE(2, 1, 1)
  GCRNN_SEQ32_CASE(5, 2, 0) GCRNN_SEQ32_CASE(4, 2, 0) GCRNN_SEQ32_CASE(3, 2, 0) GCRNN_SEQ32_CASE(2, 2, 0)
  GCRNN_SEQ32_CASE(5, 1, 0) GCRNN_SEQ32_CASE(4, 1, 0) GCRNN_SEQ32_CASE(3, 1, 0) GCRNN_SEQ32_CASE(2, 1, 0)
#undef GCRNN_SEQ32_CASE
  return 0;
}

// A(S) x_t + b for every (t, b) item as ONE launch (graphML.py:2402-2403; the node-gated cell multiplies it by the input gate per node, the
// edge-gated cell feeds it to its attention): xs [T][B][NPad][G] bf16 sequence-major, wpack = gcrnn_fused_pack_weights_wide(Fout = F) of the
// input taps with ZERO state taps (the operand is [0 | x_t]: its state half is neither loaded nor multiplied), bias [F] fp32 or NULL, out
// [T][B][NPad][F] bf16. x_user (or NULL): the user-layout X [B][T][G][N] -- the caller has laid out only the leading time steps of xs (the
// count gcrnn_fused_filter_output_wide_supported(..., with_pack = 1) returns), the items lay out the rest while they run (every item the
// operand of its workgroup's NEXT item, as the gate pair pre-pass does).
extern "C" int gcrnn_fused_filter_output_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t G, int64_t K, int64_t entries,
                                                        double uniform_w, int img16, int with_pack) {
  if (uniform_w == 0.0 || img16 != 1 || N <= 0 || N > NP || B <= 0 || T <= 0 || G <= 0 || entries <= 0 || entries % 4) return 0;
  if (B * T * (NP * (F > G ? F : G) * 2) > 2147483647LL || B * T > (1 << 24)) return 0;
  if (with_pack && (N % 8 || T * G * N > 2147483647LL)) return 0;
  if (!seq32n_wanted(B * T)) return 0;
  if (!seq32n_lds_for(F, G, K, entries, with_pack != 0)) return 0;
  if (!with_pack) return 1;
  const int64_t first = B * T < gcrnn_persistent_grid() ? B * T : gcrnn_persistent_grid();      // the items of the first round of workgroups
  return (int)((first + B - 1) / B);
}

extern "C" int gcrnn_fused_filter_output_wide_bf16(const void* xs, const void* wpack, const float* bias, void* out, const int32_t* tile_nodes,
                                                   const int32_t* tile_off, const void* ell_col4, int64_t entries, int64_t B, int64_t T,
                                                   int64_t N, int64_t F, int64_t G, int64_t K, const void* x_user, void* stream) {
  if (!xs || !wpack || !out || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  const int64_t items = B * T;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || items > (1 << 24) || entries <= 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  if (items * (NP * (F > G ? F : G) * 2) > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (x_user && (N % 8 != 0 || (reinterpret_cast<uintptr_t>(x_user) & 15) || T * G * N > 2147483647LL)) return GCRNN_ERR_BAD_SHAPE;
  Seq32Args sa{};
  sa.x0 = (const uint16_t*)xs;
  sa.hfirst = nullptr; sa.hmod = (int)B;
  sa.out0 = (uint16_t*)out;
  sa.wpack = (const uint4*)wpack; sa.bias = bias;
  sa.tile_nodes = tile_nodes; sa.tile_off = tile_off; sa.ell_col4 = (const uint2*)ell_col4;
  sa.entries = (int)entries; sa.B = (int)items; sa.N = (int)N;
  sa.nsteps = 1;
  if (x_user) {
    sa.pk_src0 = (const uint16_t*)x_user; sa.pksrc_stride = G * N; sa.pk_stride = (int)(T * G * N);
    sa.pk_dst0 = (uint16_t*)const_cast<void*>(xs);
  }
  hipStream_t st = as_stream(stream);
#define GCRNN_SEQ32_CASE(KK, HH, XX) if (K == KK && F == 32 * HH && G == 32 * XX) return x_user ? seq32n_launch<KK, HH, XX, 1, 3>(sa, st) : seq32n_launch<KK, HH, XX, 0, 3>(sa, st);
  GCRNN_SEQ32_CASE(5, 2, 2) GCRNN_SEQ32_CASE(4, 2, 2) GCRNN_SEQ32_CASE(3, 2, 2) GCRNN_SEQ32_CASE(2, 2, 2)
  GCRNN_SEQ32_CASE(5, 2, 1) GCRNN_SEQ32_CASE(4, 2, 1) GCRNN_SEQ32_CASE(3, 2, 1) GCRNN_SEQ32_CASE(2, 2, 1)
  GCRNN_SEQ32_CASE(5, 1, 1) GCRNN_SEQ32_CASE(4, 1, 1) GCRNN_SEQ32_CASE(3, 1, 1) GCRNN_SEQ32_CASE(2, 1, 1)
#undef GCRNN_SEQ32_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// The node-gated recurrence as ONE launch (graphML.py:2379-2407, 2420-2423):  h_t = tanh(gi ni_t . Yx_t + gf nf_t . (B(S) h_{t-1} + b)).
// gcrnn_fused_node_forward_bf16's contract without yh_out (inference): h0s [B][NPad][F], hs [T][B][NPad][F] (out), yx [T][B][NPad][F] = the
// filter-output pass, ngates fp32 [T][2][B][N], gi / gf fp32 [T][B] or both NULL, wpackB = gcrnn_fused_pack_weights_wide of the state taps
// alone (G = 0), Huser [B][T or 1][F][N] bf16 or NULL.
extern "C" int gcrnn_fused_node_forward_wide_supported(int64_t B, int64_t T, int64_t N, int64_t F, int64_t K, int64_t entries, double uniform_w,
                                                       int img16) {
  if (uniform_w == 0.0 || img16 != 1 || N <= 0 || N > NP || B <= 0 || T <= 0 || entries <= 0 || entries % 4) return 0;
  if (B * (NP * F * 2) > 2147483647LL || T * F * N > 2147483647LL || B * N * 8 > 2147483647LL) return 0;
  if (!seq32n_wanted(B)) return 0;
  return seq32n_lds_for(F, 0, K, entries) ? 1 : 0;
}

extern "C" int gcrnn_fused_node_forward_wide_bf16(const void* h0s, void* hs, const void* yx, const float* ngates, const float* gi, const float* gf,
                                                  const void* wpackB, const float* bias, const int32_t* tile_nodes, const int32_t* tile_off,
                                                  const void* ell_col4, int64_t entries, int64_t B, int64_t T, int64_t N, int64_t F, int64_t K,
                                                  void* Huser, int huser_last_only, void* stream) {
  if (!h0s || !hs || !yx || !ngates || !wpackB || !tile_nodes || !tile_off || !ell_col4) return GCRNN_ERR_NULL_POINTER;
  if ((gi == nullptr) != (gf == nullptr)) return GCRNN_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || N <= 0 || N > NP || B > (1 << 24) || entries <= 0 || entries % 4) return GCRNN_ERR_BAD_SHAPE;
  if (B * (NP * F * 2) > 2147483647LL || T * F * N > 2147483647LL || B * N * 8 > 2147483647LL) return GCRNN_ERR_BAD_SHAPE;
  if (Huser && (N % 8 != 0 || (reinterpret_cast<uintptr_t>(Huser) & 15))) return GCRNN_ERR_BAD_SHAPE;
  const int64_t hstep = B * NP * F;
  Seq32Args sa{};
  sa.hfirst = (const uint16_t*)h0s;
  sa.out0 = (uint16_t*)hs; sa.ostride = hstep;
  sa.wpack = (const uint4*)wpackB; sa.bias = bias;
  sa.a1 = (const uint16_t*)Huser; sa.a1stride = F * N; sa.a1_last_only = huser_last_only ? 1 : 0;
  sa.ubstride = (int)((huser_last_only ? 1 : T) * F * N);
  sa.tile_nodes = tile_nodes; sa.tile_off = tile_off; sa.ell_col4 = (const uint2*)ell_col4;
  sa.entries = (int)entries; sa.B = (int)B; sa.N = (int)N;
  sa.nsteps = (int)T;
  sa.dh0_ = (const uint16_t*)yx; sa.dhstride = hstep;
  sa.ng0 = ngates; sa.ngstride = 2 * B * N; sa.nghalf = B * N;
  sa.gi0 = gi; sa.gf0 = gf; sa.gstride = B;
  hipStream_t st = as_stream(stream);
#define GCRNN_SEQ32_CASE(KK, HH) if (K == KK && F == 32 * HH) return Huser ? seq32n_launch<KK, HH, 0, 2, 4>(sa, st) : seq32n_launch<KK, HH, 0, 0, 4>(sa, st);
  GCRNN_SEQ32_CASE(5, 2) GCRNN_SEQ32_CASE(4, 2) GCRNN_SEQ32_CASE(3, 2) GCRNN_SEQ32_CASE(2, 2)
  GCRNN_SEQ32_CASE(5, 1) GCRNN_SEQ32_CASE(4, 1) GCRNN_SEQ32_CASE(3, 1) GCRNN_SEQ32_CASE(2, 1)
#undef GCRNN_SEQ32_CASE
  return GCRNN_ERR_UNSUPPORTED;
}

// The time gates' read-out, finished in ONE launch (reference Utils/graphML.py:2364-2366, 2372-2374: gate = sigmoid(w . vec(c) + c0)): the pair
// pre-pass leaves per item and gate `nparts` partial dot products (one per 32-feature chunk and wave, gcrnn_fused_gate_pair_prepass_wide_bf16);
// this adds them in a FIXED order (j = 0 .. nparts-1: the same bits on every run), adds the read-out's bias and applies the sigmoid.
// parts [items][2][nparts] fp32, lb_in / lb_f device scalars (fp32) or NULL, gi / gf [items] fp32 (items = T B, item = t B + b).
// Until round 4 this was five torch launches (sum, add, sigmoid, add, sigmoid) between the pre-pass and the recurrence.
__global__ void gate_readout_finish_kernel(const float* __restrict__ parts, int nparts, const float* __restrict__ lb_in,
                                           const float* __restrict__ lb_f, float* __restrict__ gi, float* __restrict__ gf, int64_t items) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= items) return;
  const float* p = parts + i * 2 * nparts;
  float a = 0.f, b = 0.f;
  for (int j = 0; j < nparts; ++j) { a += p[j]; b += p[nparts + j]; }
  if (lb_in) a += lb_in[0];
  if (lb_f) b += lb_f[0];
  gi[i] = 1.f / (1.f + __expf(-a));
  gf[i] = 1.f / (1.f + __expf(-b));
}

extern "C" int gcrnn_gate_readout_finish(const float* parts, int64_t nparts, const float* lb_in, const float* lb_f, float* gi, float* gf,
                                         int64_t items, void* stream) {
  if (!parts || !gi || !gf) return GCRNN_ERR_NULL_POINTER;
  if (items <= 0 || nparts <= 0 || nparts > 4096) return GCRNN_ERR_BAD_SHAPE;
  GCRNN_PRE_LAUNCH();
  gate_readout_finish_kernel<<<(unsigned)cdiv(items, 256), 256, 0, as_stream(stream)>>>(parts, (int)nparts, lb_in, lb_f, gi, gf, items);
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}
