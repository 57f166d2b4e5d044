# Same-box A/B of the csrc/ tree against an older copy of it (a directory with every source, e.g. `git show` of a commit into
# tools/probes/old_csrc/): builds both libraries on the GPU box and alternates the bench command.
# usage: bash tools/ab_tree.sh tools/probes/old_csrc "bench.py --no-cpu-baseline --no-secondary" [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OLD=$R/$1; CMD="$2"; REPS=${3:-3}
rm -rf /tmp/abt; mkdir -p /tmp/abt/pkg/csrc /tmp/abt/include /tmp/abt/o; cp $OLD/* /tmp/abt/pkg/csrc/; cp $R/include/gcrnn.h /tmp/abt/include/
( for f in /tmp/abt/pkg/csrc/*.hip /tmp/abt/pkg/csrc/gcrnn_host.cpp; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -w -c $f -o /tmp/abt/o/$(basename $f).o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abt/lib_old.so /tmp/abt/o/*.o ) 2>&1 | grep -E "error" | head -3
for rep in $(seq $REPS); do
  echo -n "new: "; python3 $R/$CMD 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
  echo -n "old: "; GCRNN_LIBPATH=/tmp/abt/lib_old.so python3 $R/$CMD 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
