# Timing experiment (WRONG results): only some waves of a workgroup run the hop stream. Waves w and w + 4 share a SIMD; with waves 0..3
# alone every streaming wave has its SIMD to itself. If a wave's stream time does not change, it is bound by its own dependent chain
# (LDS latency per trip); if it shrinks, by the SIMD's issue slots it shares with its partner; if it shrinks further with fewer SIMDs
# streaming, by a unit shared across the CU (LDS).  bash tools/stream_waves_experiment.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" "-DGCRNN_EXPERIMENT_STREAM_WAVES=&&wave<4" "-DGCRNN_EXPERIMENT_STREAM_WAVES=&&wave<1" "-DGCRNN_EXPERIMENT_STREAM_WAVES=&&(wave&3)==0"; do
  echo "=== stamps build $v"
  GCRNN_STAMP_FLAGS="$v" python3 $R/tools/seq_stamps.py 2>&1 | grep -E "c1 (taps|hop)"
done
