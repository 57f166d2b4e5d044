# Timing experiments (WRONG results), ONE streaming wave per workgroup: the trip without single instruction groups. bash tools/stream_trip_experiment.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "=== $1"; shift; env "$@" GCRNN_HOP16_EXPERIMENT_NO_WAIT=1 GCRNN_HOP16_EXPERIMENT_NO_MFMA=1 GCRNN_STAMP_FLAGS="-DGCRNN_EXPERIMENT_STREAM_WAVES=&&wave<1" python3 $R/tools/seq_stamps.py 2>&1 | grep -E "c1 hop [12]"; }
run "no wait, no matrix instruction (base of this series)" GCRNN_HOP16_EXPERIMENT_TRIP=base
run "... plain v_and instead of v_xor_sdwa" GCRNN_HOP16_EXPERIMENT_TRIP=plainxor
run "... no address VALU at all (gathers from one fixed address)" GCRNN_HOP16_EXPERIMENT_TRIP=noxor
run "... no gathers" GCRNN_HOP16_EXPERIMENT_TRIP=nogather
run "... no gathers, no address VALU" GCRNN_HOP16_EXPERIMENT_TRIP=nogather,noxor
run "... no column read / clamp / pointer add" GCRNN_HOP16_EXPERIMENT_TRIP=nocol
run "... only branch + counter left" GCRNN_HOP16_EXPERIMENT_TRIP=nogather,noxor,nocol
