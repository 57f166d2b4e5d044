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
