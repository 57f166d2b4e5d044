# Round-4 profile run (on the GPU box, from the repo root): bash tools/profile_r04.sh [quick]
# Kernel-trace stats of the bench lines; PMC sets of the dominant kernel (fused_seq32_kernel: one launch = T steps; the three ways it is issued
# are kernel symbols of their own: <..,3> as the module issues it, <..,2> caller-packed X, <..,0> sequence-major in and out), one counter set per pass.
TAG=r04
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python3 $R/bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.json 2> $O/bench_$n.err
  python3 $R/tools/rocprof_db_stats.py $O/kt_$n > $O/${TAG}_${n}_kernel_stats.csv 2>/dev/null
  rm -rf $O/kt_$n
  echo "$n: $(head -c 160 $O/bench_$n.json)"
}
stats bench_b256 --steps 5 --warmup 2
stats bench_fwd_timegated --time-gating --steps 3 --warmup 1
stats bench_train_bf16 --mode train --steps 3 --warmup 1
if [ "$1" != "quick" ]; then
stats bench_train_timegated --mode train --time-gating --steps 3 --warmup 1
stats bench_fwd_nodegated --spatial-gating node --steps 3 --warmup 1
stats bench_fwd_edgegated --spatial-gating edge --steps 3 --warmup 1
stats bench_f32_x3 --dtype f32 --steps 3 --warmup 1
# the time-gated cell's training at 1e-5 (ops._FusedTimeCellX3), per kernel
rocprofv3 --kernel-trace --stats -d $O/kt_x3g -- python3 $R/tools/x3_gated_train_profile.py > $O/x3g_train.log 2>&1
python3 $R/tools/rocprof_db_stats.py $O/kt_x3g > $O/${TAG}_x3_timegated_train_kernel_stats.csv 2>/dev/null
rm -rf $O/kt_x3g
grep "time-gated x3" $O/x3g_train.log
# dispatch timelines of one forward (which kernels a captured forward replays: layout kernels + main kernels only)
rocprofv3 --kernel-trace -d $O/kt_tl -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 $R/tools/rocprof_db_timeline.py $O/kt_tl 24 > $O/${TAG}_forward_dispatch_timeline.txt 2>/dev/null
rm -rf $O/kt_tl
rocprofv3 --kernel-trace -d $O/kt_tl -- python3 $R/bench.py --time-gating --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 $R/tools/rocprof_db_timeline.py $O/kt_tl 24 > $O/${TAG}_forward_timegated_dispatch_timeline.txt 2>/dev/null
rm -rf $O/kt_tl
# un-profiled lines, as the driver runs them
python3 $R/bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $O/${TAG}_bench_driver_form.json 2>/dev/null      # the driver's command line
python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_bf16.json 2>/dev/null
python3 $R/bench.py --time-gating --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_timegated.json 2>/dev/null
python3 $R/bench.py --mode train --time-gating --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_timegated.json 2>/dev/null
python3 $R/bench.py --in-features 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_g1.json 2>/dev/null
python3 $R/bench.py --steps 20 --warmup 5 --settle-ms 0 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_default_cold_start.json 2>/dev/null
python3 $R/bench.py --mode train --batch 100 --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_bf16_b100.json 2>/dev/null
python3 $R/bench.py --time-gating --batch 100 --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_timegated_b100.json 2>/dev/null
python3 $R/bench.py --gso normalized --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_normalized_adjacency.json 2>/dev/null
python3 $R/bench.py --gso normalized --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_normalized_adjacency.json 2>/dev/null
python3 $R/bench.py --gso normalized --time-gating --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_timegated_normalized_adjacency.json 2>/dev/null
python3 $R/bench.py --gso normalized --time-gating --mode train --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_train_timegated_normalized_adjacency.json 2>/dev/null
# ---- PMC: the wide sequence-resident kernel at B = 256, T = 32 (user-layout API, inline pack), one set per pass
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_seq/$n -- python3 $R/tools/step_kernel_probe.py 256 32 1 > $O/pmc_seq.$n.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_seq fused_seq32 > $O/${TAG}_seq32_kernel_pmc.txt
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_nat/$set -- python3 $R/tools/step_kernel_probe.py 256 32 1 native > $O/pmc_nat.$set.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_nat fused_seq32 > $O/${TAG}_seq32_kernel_native_pmc.txt
cat $O/${TAG}_seq32_kernel_pmc.txt $O/${TAG}_seq32_kernel_native_pmc.txt
rm -rf $O/pmc_seq $O/pmc_nat
fi
ls $O | head -60
# ---- traces that contain ONLY one way of issuing the kernel (2 x 60 back-to-back forwards: the clock transient behind an idle gap, profiles/r04_clock_transient.txt, weighs ~2 % in the average): the average of the
# fused_seq32_kernel row of each CSV is the duration the bench line's roofline.frac / roofline_native_layout.frac rest on
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt_asissued -- python3 $R/tools/step_kernel_probe.py 256 32 60 > $O/probe_as_issued.log 2>&1
python3 $R/tools/rocprof_db_stats.py $O/kt_asissued > $O/${TAG}_seq32_as_issued_only_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/kt_native -- python3 $R/tools/step_kernel_probe.py 256 32 60 native > $O/probe_native.log 2>&1
python3 $R/tools/rocprof_db_stats.py $O/kt_native > $O/${TAG}_seq32_native_only_kernel_stats.csv 2>/dev/null
rm -rf $O/kt_asissued $O/kt_native
head -3 $O/${TAG}_seq32_as_issued_only_kernel_stats.csv $O/${TAG}_seq32_native_only_kernel_stats.csv
