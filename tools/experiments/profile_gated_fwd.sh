# Kernel-trace stats of the gated forward bench lines (on the GPU box, from the repo root): bash tools/profile_gated_fwd.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_gated_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python3 $R/tools/rocprof_db_stats.py $O/kt_$n > $O/${TAG}_${n}_kernel_stats.csv 2>/dev/null
  rm -rf $O/kt_$n
  echo "$n: $(head -c 200 $O/bench_$n.json)"
}
stats bench_fwd_nodegated --spatial-gating node --steps 3 --warmup 1
stats bench_fwd_timegated --time-gating --steps 3 --warmup 1
