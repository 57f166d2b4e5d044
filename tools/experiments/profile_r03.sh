# Round-3 profile run (on the GPU box, from the repo root): bash tools/profile_r03.sh
# Kernel-trace stats of the bench lines, PMC sets of the dominant kernel (fused_seq_kernel: one launch = T steps), one counter set per pass.
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python3 $R/bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.json 2> $O/bench_$n.err
  python3 $R/tools/rocprof_db_stats.py $O/kt_$n > $O/${TAG}_${n}_kernel_stats.csv 2>/dev/null
  rm -rf $O/kt_$n
  echo "$n: $(head -c 200 $O/bench_$n.json)"
}
stats bench_b256 --steps 5 --warmup 2
stats bench_train_bf16 --mode train --steps 3 --warmup 1
stats bench_f32_x3 --dtype f32 --steps 3 --warmup 1
stats bench_train_f32_x3 --dtype f32 --mode train --steps 2 --warmup 1
stats bench_fwd_timegated --time-gating --steps 3 --warmup 1
stats bench_train_timegated --mode train --time-gating --steps 3 --warmup 1
stats bench_fwd_nodegated --spatial-gating node --steps 3 --warmup 1
stats bench_fwd_edgegated --spatial-gating edge --steps 3 --warmup 1
# un-profiled default line (host baseline and secondary points), as the driver runs it
python3 $R/bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_bf16.json 2>/dev/null
python3 $R/bench.py --dtype f32 --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_train_f32_x3.json 2>/dev/null
python3 $R/bench.py --time-gating --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_timegated.json 2>/dev/null
python3 $R/bench.py --spatial-gating node --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_nodegated.json 2>/dev/null
python3 $R/bench.py --spatial-gating edge --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_edgegated.json 2>/dev/null
python3 $R/bench.py --in-features 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_fwd_g1.json 2>/dev/null
python3 $R/bench.py --mode train --time-gating --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_timegated.json 2>/dev/null
python3 $R/bench.py --mode train --spatial-gating node --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_nodegated.json 2>/dev/null
python3 $R/bench.py --mode train --spatial-gating edge --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_train_edgegated.json 2>/dev/null
python3 $R/bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_f32_x3.json 2>/dev/null
python3 $R/tools/driver_shape_bench.py > $O/${TAG}_driver_shape_f20.jsonl 2>/dev/null
# ---- PMC: the sequence-resident kernel at B = 256, T = 32 (user-layout API, inline pack), one set per pass
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_seq/$n -- python3 $R/tools/step_kernel_probe.py 256 32 1 > $O/pmc_seq.$n.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_seq fused_seq > $O/${TAG}_seq_kernel_pmc.txt
# the same kernel on the sequence-major arrays alone (forward_native): traffic of the native-layout path
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_nat/$set -- python3 $R/tools/step_kernel_probe.py 256 32 1 native > $O/pmc_nat.$set.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_nat fused_seq > $O/${TAG}_seq_kernel_native_pmc.txt
cat $O/${TAG}_seq_kernel_pmc.txt $O/${TAG}_seq_kernel_native_pmc.txt
rm -rf $O/pmc_seq $O/pmc_nat
ls $O | head -60
