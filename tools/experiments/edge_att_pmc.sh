# PMC sets of the fused edge-attention kernel (on the GPU box, from the repo root): bash tools/edge_att_pmc.sh <tag> [items] [mode]
TAG=${1:-r02}; ITEMS=${2:-2048}; MODE=${3:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_edge_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/edge_att_probe.py $ITEMS 5 $MODE > $O/probe.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/$n -- python3 $R/tools/edge_att_probe.py $ITEMS 2 $MODE > $O/pmc.$n.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc edge_att > $O/${TAG}_edge_att_pmc.txt
cat $O/probe.txt $O/${TAG}_edge_att_pmc.txt
rm -rf $O/pmc
