# Same-box A/B of prebuilt library variants (tools/build_variant_lib.sh): alternates bench.py runs over the given libraries.
#   bash tools/ab_libs.sh [reps] default lpe8 fpe2 ...      ("default" = the in-tree library; "old" = the in-tree library with GCRNN_SEQ32P=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=$1; shift
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), "%.4f ms" % d["ms_per_step"], "as issued %.1f native %.1f us" % (d["roofline"]["kernel_avg_us"], d["roofline_native_layout"]["kernel_avg_us"]))'
for rep in $(seq $REPS); do
  for v in "$@"; do
    echo -n "$v: "
    if [ "$v" = default ]; then timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "$show"
    elif [ "$v" = old ]; then GCRNN_SEQ32P=0 timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "$show"
    else GCRNN_LIBPATH=$R/gated_gcrnns_amd/lib/variants/$v.so timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "$show"; fi
  done
done
