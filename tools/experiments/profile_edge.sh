# Kernel-trace stats of the edge-gated bench lines (on the GPU box, from the repo root): bash tools/profile_edge.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_edge_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python3 $R/tools/rocprof_db_stats.py $O/kt_$n > $O/${TAG}_${n}_kernel_stats.csv 2>/dev/null
  echo "$n: $(head -c 300 $O/bench_$n.json)"
}
stats bench_fwd_edgegated --spatial-gating edge --steps 2 --warmup 1
[ "$2" = fwd ] || stats bench_train_edgegated --mode train --spatial-gating edge --steps 2 --warmup 1
rm -rf $O/kt_*
