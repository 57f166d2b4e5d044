# Timing experiments (WRONG results) on ONE streaming wave per workgroup (wave 0 alone: no partner on its SIMD, no other gathers in the CU):
# the stream as it is / without its s_waitcnt / without its matrix instruction / with three groups in flight -- where does a trip stall?
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "=== $1"; shift; env "$@" GCRNN_STAMP_FLAGS="-DGCRNN_EXPERIMENT_STREAM_WAVES=&&wave<1" python3 $R/tools/seq_stamps.py 2>&1 | grep -E "c1 hop [12]"; }
run "as it is" GCRNN_HOP16_SUMS_DEPTH=2
run "no s_waitcnt in the trip" GCRNN_HOP16_SUMS_DEPTH=2 GCRNN_HOP16_EXPERIMENT_NO_WAIT=1
run "no matrix instruction" GCRNN_HOP16_SUMS_DEPTH=2 GCRNN_HOP16_EXPERIMENT_NO_MFMA=1
run "neither" GCRNN_HOP16_SUMS_DEPTH=2 GCRNN_HOP16_EXPERIMENT_NO_MFMA=1 GCRNN_HOP16_EXPERIMENT_NO_WAIT=1
run "three groups in flight" GCRNN_HOP16_SUMS_DEPTH=3
