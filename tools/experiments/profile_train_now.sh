# Kernel-trace stats of the bf16 training line (on the GPU box): bash tools/profile_train_now.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_train_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python3 $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 $R/tools/rocprof_db_stats.py $O/kt > $O/${TAG}_bench_train_bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/kt
