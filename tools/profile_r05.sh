# Round-5 profile run (on the GPU box, from the repo root): bash tools/profile_r05.sh [stats|lines|pmc]   (three parts: a gpurun call is limited to 20 minutes)
# Kernel-trace stats of the bench lines; PMC sets of the two dominant kernel symbols -- fused_seq32_kernel<5,2,2,3> (the forward as the module
# issues it: inline pack + user-layout copy) and fused_seq32p_kernel<5,2,2,0> (sequence-major in and out: the hand-allocated-hop kernel) -- one
# counter set per pass; traces that contain only one of them; cfg4 / cfg5 lines on this tree.
TAG=r05
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
PART=${1:-all}
if [ $PART = all ] || [ $PART = stats ]; then
stats bench_b256 --steps 5 --warmup 2
stats bench_fwd_timegated --time-gating --steps 3 --warmup 1
stats bench_fwd_nodegated --spatial-gating node --steps 3 --warmup 1
stats bench_fwd_edgegated --spatial-gating edge --steps 3 --warmup 1
stats bench_train_bf16 --mode train --steps 3 --warmup 1
stats bench_f32_x3 --dtype f32 --steps 3 --warmup 1
stats bench_cfg4 --config cfg4 --steps 5 --warmup 2
stats bench_cfg5_bf16 --config cfg5 --steps 3 --warmup 1
fi
if [ $PART = all ] || [ $PART = lines ]; then
echo "--- un-profiled lines"
python3 $R/bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $O/${TAG}_bench_driver_form.json 2>/dev/null      # the driver's command line
python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_bf16.json 2>/dev/null
python3 $R/bench.py --time-gating --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_timegated.json 2>/dev/null
python3 $R/bench.py --spatial-gating node --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_nodegated.json 2>/dev/null
python3 $R/bench.py --spatial-gating edge --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_edgegated.json 2>/dev/null
python3 $R/bench.py --config cfg4 > $O/${TAG}_bench_cfg4.json 2>/dev/null
python3 $R/bench.py --config cfg5 > $O/${TAG}_bench_cfg5_bf16.json 2>/dev/null
python3 $R/bench.py --config cfg5 --dtype f32 --no-cpu-baseline > $O/${TAG}_bench_cfg5_f32.json 2>/dev/null
for f in default driver_form train_bf16 fwd_timegated fwd_nodegated fwd_edgegated cfg4 cfg5_bf16 cfg5_f32; do echo "$f: $(head -c 220 $O/${TAG}_bench_$f.json)"; done
fi
if [ $PART = all ] || [ $PART = pmc ]; then
echo "--- PMC"
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
for set in "${SETS[@]}"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_seq/$n -- python3 $R/tools/step_kernel_probe.py 256 32 1 > $O/pmc_seq.$n.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_nat/$n -- python3 $R/tools/step_kernel_probe.py 256 32 1 native > $O/pmc_nat.$n.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_seq fused_seq32_kernel > $O/${TAG}_seq32_kernel_pmc.txt
python3 $R/tools/pmc_summary.py $O/pmc_nat fused_seq32p_kernel > $O/${TAG}_seq32p_kernel_native_pmc.txt
cat $O/${TAG}_seq32_kernel_pmc.txt; echo; cat $O/${TAG}_seq32p_kernel_native_pmc.txt
rm -rf $O/pmc_seq $O/pmc_nat
echo "--- traces of one way of issuing the kernel"
rocprofv3 --kernel-trace --stats -d $O/kt_asissued -- python3 $R/tools/step_kernel_probe.py 256 32 60 > $O/probe_as_issued.log 2>&1
python3 $R/tools/rocprof_db_stats.py $O/kt_asissued > $O/${TAG}_seq32_as_issued_only_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/kt_native -- python3 $R/tools/step_kernel_probe.py 256 32 60 native > $O/probe_native.log 2>&1
python3 $R/tools/rocprof_db_stats.py $O/kt_native > $O/${TAG}_seq32p_native_only_kernel_stats.csv 2>/dev/null
rm -rf $O/kt_asissued $O/kt_native
head -3 $O/${TAG}_seq32_as_issued_only_kernel_stats.csv $O/${TAG}_seq32p_native_only_kernel_stats.csv
fi
rm -f $O/*.err $O/pmc_*.log $O/probe_*.log
ls $O
