R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_tr
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in plain node edge; do
  a=""; [ $n = node ] && a="--spatial-gating node"; [ $n = edge ] && a="--spatial-gating edge"
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python3 $R/bench.py --mode train $a --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python3 $R/tools/rocprof_db_stats.py $O/kt_$n > $O/${n}_kernel_stats.csv 2>/dev/null
done
rm -rf $O/kt_*
