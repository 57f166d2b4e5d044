# A/B kernel traces of the sequence-resident kernel (GCRNN_SEQ_KERNEL=1/0), training and forward, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_seq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export GCRNN_SEQ_KERNEL=$v
  rocprofv3 --kernel-trace --stats -d $O/kt_tr$v -- python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_tr$v.json 2> $O/bench_tr$v.err
  python3 $R/tools/rocprof_db_stats.py $O/kt_tr$v > $O/train_seq${v}_kernel_stats.csv 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $O/kt_fw$v -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fw$v.json 2> $O/bench_fw$v.err
  python3 $R/tools/rocprof_db_stats.py $O/kt_fw$v > $O/fwd_seq${v}_kernel_stats.csv 2>/dev/null
done
rm -rf $O/kt_*
