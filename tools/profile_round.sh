# Round profile run (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# Kernel-trace stats of the bench lines, PMC sets of the two dominant kernels (fused step kernel, streaming SpMM), one counter set per pass.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python3 $R/tools/rocprof_db_stats.py $O/kt_$n > $O/${TAG}_${n}_kernel_stats.csv 2>/dev/null
  echo "$n: $(head -c 300 $O/bench_$n.json)"
}
stats bench_b256 --steps 5 --warmup 2
stats bench_f32_x3 --dtype f32 --steps 3 --warmup 1
stats bench_train_bf16 --mode train --steps 3 --warmup 1
stats bench_train_timegated --mode train --time-gating --steps 3 --warmup 1
stats bench_train_nodegated --mode train --spatial-gating node --steps 3 --warmup 1
stats bench_fwd_edgegated --spatial-gating edge --steps 3 --warmup 1
stats bench_train_edgegated --mode train --spatial-gating edge --steps 3 --warmup 1
stats bench_cfg5_bf16 --config cfg5 --steps 3 --warmup 1
stats bench_cfg5_f32 --config cfg5 --dtype f32 --steps 2 --warmup 1
stats bench_cfg4 --config cfg4 --steps 5 --warmup 2
# un-profiled default line (with the host baseline), as the driver runs it
python3 $R/bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
# ---- PMC: fused step kernel (B = 256), one set per pass
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_step/$n -- python3 $R/tools/step_kernel_probe.py 256 4 1 > $O/pmc_step.$n.log 2>&1
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_step/fetch -- python3 $R/tools/step_kernel_probe.py 256 4 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_step/write -- python3 $R/tools/step_kernel_probe.py 256 4 1 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $O/pmc_step fused_step > $O/${TAG}_step_kernel_pmc.txt
# ---- PMC: streaming SpMM at cfg5 (bf16)
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_spmm/$n -- python3 $R/bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_spmm.$n.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_spmm spmm_stream > $O/${TAG}_cfg5_spmm_pmc.txt
ls $O | head -50
# keep what travels back small: drop the rocpd databases and raw counter dumps, keep the summaries
rm -rf $O/kt_* $O/pmc_step $O/pmc_spmm
