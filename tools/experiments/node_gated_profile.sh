# Node-gated forward (bench line + per-kernel stats): bash tools/node_gated_profile.sh  (on the GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ng
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --spatial-gating node --no-cpu-baseline --no-secondary > $O/r04_nodegated_bench.json 2> $O/bench.err || exit 1
GCRNN_NO_NODE_GATE_FILTER=1 python3 $R/bench.py --spatial-gating node --no-cpu-baseline --no-secondary > $O/r04_nodegated_bench_hop_per_launch.json 2>> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/kt -- python3 $R/bench.py --spatial-gating node --steps 10 --no-cpu-baseline --no-secondary > $O/prof.log 2>&1 || exit 1
python3 $R/tools/rocprof_db_stats.py $O/kt > $O/r04_nodegated_kernel_stats.csv
rm -rf $O/kt
python3 - <<PY
import json,csv
for f in ("r04_nodegated_bench.json","r04_nodegated_bench_hop_per_launch.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"]), round(d["ms_per_step"],3))
for i,r in enumerate(csv.reader(open("$O/r04_nodegated_kernel_stats.csv"))):
    if i>6: break
    print(r[0][:70], r[1], r[3])
PY
# ---- PMC of the one-pass gate filter (node_gate_filter_kernel), one counter set per pass
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/$n -- python3 $R/bench.py --spatial-gating node --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-secondary > $O/pmc.$n.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $O/pmc node_gate_filter > $O/r04_node_gate_filter_pmc.txt
rm -rf $O/pmc
cat $O/r04_node_gate_filter_pmc.txt
