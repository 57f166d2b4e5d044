R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_step_new
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/$n -- python3 $R/tools/step_kernel_probe.py 256 4 1 > $O/pmc.$n.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc fused_step > $O/step_pmc_new.txt
cat $O/step_pmc_new.txt
rm -rf $O/pmc
