# Same-box A/B of ONE source file against an older copy of it: builds the library twice on the GPU box (the tree as it is, and with
# csrc/<file> replaced by <old copy>) and runs the bench command with each, alternating.
# usage: bash tools/ab_src.sh gcrnn_fused_wgrad.hip tools/probes/ab_old_wgrad.hip.txt "bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline" [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
F=$1; OLD=$R/$2; CMD="$3"; REPS=${4:-2}
rm -rf /tmp/abs; mkdir -p /tmp/abs/src /tmp/abs/o /tmp/include
cp $C/* /tmp/abs/src/; cp $OLD /tmp/abs/src/$F; cp $R/include/gcrnn.h /tmp/include/
mkdir -p /tmp/abs/include && cp $R/include/gcrnn.h /tmp/abs/include/ 2>/dev/null
( for f in /tmp/abs/src/*.hip /tmp/abs/src/gcrnn_host.cpp; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c $f -o /tmp/abs/o/$(basename $f).o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abs/lib_old.so /tmp/abs/o/*.o ) 2>&1 | grep -E "error" | head -3
for rep in $(seq $REPS); do
  echo -n "new: "; python3 $R/$CMD 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
  echo -n "old $F: "; GCRNN_LIBPATH=/tmp/abs/lib_old.so python3 $R/$CMD 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
