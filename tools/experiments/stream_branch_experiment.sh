# Timing experiment, ONE streaming wave per workgroup: the full trip with 0 / 2 / 4 extra never-taken branches, or 4 extra scalar moves.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "=== $1"; shift; env "$@" GCRNN_STAMP_FLAGS="-DGCRNN_EXPERIMENT_STREAM_WAVES=&&wave<1" python3 $R/tools/seq_stamps.py 2>&1 | grep -E "c1 hop [12]"; }
run "as it is" GCRNN_HOP16_SUMS_DEPTH=2
run "+2 not-taken branches per trip" GCRNN_HOP16_EXPERIMENT_EXTRA_BRANCHES=2
run "+4 not-taken branches per trip" GCRNN_HOP16_EXPERIMENT_EXTRA_BRANCHES=4
run "+4 scalar moves per trip" GCRNN_HOP16_EXPERIMENT_EXTRA_SALU=4
