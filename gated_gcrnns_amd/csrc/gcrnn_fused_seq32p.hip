// Instantiations and launcher of the hand-allocated-hop flagship forward (gcrnn_fused_seq32p.h).
#include "gcrnn_fused_step.h"
#define GCRNN_SEQ32_STAMP_READER_NAME gcrnn_debug_read_seq32p_stamps      // (diagnostic builds: this unit's own stamp array and reader)
#include "gcrnn_fused_seq32.h"
#include "gcrnn_fused_seq32p.h"

template <int K, int HS, int XS, int VAR>
static int seq32p_launch_v(const Seq32Args& sa, size_t lds, hipStream_t st) {
  auto sk = fused_seq32p_kernel<K, HS, XS, VAR>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GCRNN_ERR_LAUNCH;
  GCRNN_PRE_LAUNCH();
  sk<<<(unsigned)(sa.B < gcrnn_persistent_grid() ? sa.B : gcrnn_persistent_grid()), STHREADS, lds, st>>>(sa);      // one workgroup per CU (count read from the device once)
  GCRNN_CHECK_LAUNCH();
  return GCRNN_OK;
}

template <int K, int HS, int XS>
static int seq32p_launch(const Seq32Args& sa, bool inline_pack, size_t lds, hipStream_t st) {
  const int var = (inline_pack ? 1 : 0) | (sa.a1 ? 2 : 0);
  switch (var) {
    case 0: return seq32p_launch_v<K, HS, XS, 0>(sa, lds, st);
    case 1: return seq32p_launch_v<K, HS, XS, 1>(sa, lds, st);
    case 2: return seq32p_launch_v<K, HS, XS, 2>(sa, lds, st);
    default: return seq32p_launch_v<K, HS, XS, 3>(sa, lds, st);
  }
}

// the un-gated persistent forward on a uniform-weight graph (gcrnn_fused_seq32.hip dispatches here unless GCRNN_SEQ32P=0)
int gcrnn_seq32p_forward(const Seq32Args& sa, int K, int HS, int XS, bool inline_pack, size_t lds, hipStream_t st) {
#define GCRNN_SEQ32P_CASE(KK, HH, XX) if (K == KK && HS == HH && XS == XX) return seq32p_launch<KK, HH, XX>(sa, inline_pack, lds, st);
  GCRNN_SEQ32P_CASE(5, 2, 2) GCRNN_SEQ32P_CASE(4, 2, 2) GCRNN_SEQ32P_CASE(3, 2, 2) GCRNN_SEQ32P_CASE(2, 2, 2)
  GCRNN_SEQ32P_CASE(5, 2, 1) GCRNN_SEQ32P_CASE(4, 2, 1) GCRNN_SEQ32P_CASE(3, 2, 1) GCRNN_SEQ32P_CASE(2, 2, 1)
  GCRNN_SEQ32P_CASE(5, 1, 1) GCRNN_SEQ32P_CASE(4, 1, 1) GCRNN_SEQ32P_CASE(3, 1, 1) GCRNN_SEQ32P_CASE(2, 1, 1)
#undef GCRNN_SEQ32P_CASE
  return GCRNN_ERR_UNSUPPORTED;
}
