# Forward throughput at batches that do not fill the chip: the wide sequence-resident kernel (one workgroup per sequence, persistent; or -- B <= 128 --
# one workgroup per (sequence, 32-feature chunk), one launch per step) against the chunk-parallel kernel (GCRNN_SEQ32=0 GCRNN_SEQ_KERNEL=0). usage on the GPU box: bash tools/small_batches.sh "64 100 128 160 192 256"
R=${GRAFT_REPO_ROOT:-$(pwd)}
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), "seq/s", round(d["ms_per_step"],3), "ms")'
for B in ${1:-64 100 128 160 192 256}; do
  echo -n "B=$B default dispatch: "; python3 $R/bench.py --batch $B --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$show"
  echo -n "B=$B split sequences forced (B <= 128): "; GCRNN_SEQ32_SPLIT=1 python3 $R/bench.py --batch $B --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$show"
  echo -n "B=$B persistent wide kernel forced: "; GCRNN_SEQ32_MIN_B=1 python3 $R/bench.py --batch $B --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$show"
  echo -n "B=$B chunk-parallel kernel: "; GCRNN_SEQ32=0 GCRNN_SEQ_KERNEL=0 python3 $R/bench.py --batch $B --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$show"
done
