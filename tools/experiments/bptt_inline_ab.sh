R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/bptt_ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/a -- python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $O/a.json 2>/dev/null
python3 $R/tools/rocprof_db_stats.py $O/a > $O/a.csv 2>/dev/null
export GCRNN_NO_INLINE_PACK=1
rocprofv3 --kernel-trace --stats -d $O/b -- python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $O/b.json 2>/dev/null
python3 $R/tools/rocprof_db_stats.py $O/b > $O/b.csv 2>/dev/null
for f in a b; do echo $f; grep "fused_step_kernel\|seq_pack\|seq_layout" $O/$f.csv | cut -c1-60,150-230; done
rm -rf $O/a $O/b
