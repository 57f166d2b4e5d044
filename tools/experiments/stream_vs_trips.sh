# The stream's time against its trip count: the same kernel on graphs of 0.6x / 1x / 2x the edge density, one streaming wave per workgroup.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for d in 0.6 1 2; do
  echo "=== density x$d"
  GCRNN_STAMP_DENSITY=$d GCRNN_STAMP_FLAGS="-DGCRNN_EXPERIMENT_STREAM_WAVES=&&wave<1" python3 $R/tools/seq_stamps.py 2>&1 | grep -E "graph:|c1 (taps\(hop 2|hop [12])"
done
