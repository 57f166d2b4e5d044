# Same-box A/B of a compile-time switch: builds the library twice on the GPU box (default flags, and with the given -D flags) and
# runs the given bench command with each, alternating.  usage: bash tools/ab_build.sh "-DGCRNN_FENCED_BARRIERS" "bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline" [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
FL="$1"; CMD="$2"; REPS=${3:-2}
mkdir -p /tmp/ab/a /tmp/ab/b
( for f in $C/*.hip $C/gcrnn_host.cpp; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC $FL -c $f -o /tmp/ab/b/$(basename $f).o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab/lib_b.so /tmp/ab/b/*.o ) 2>&1 | grep -E "error" | head -3
for rep in $(seq $REPS); do
  echo -n "default: "; python3 $R/$CMD 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
  echo -n "$FL: "; GCRNN_LIBPATH=/tmp/ab/lib_b.so python3 $R/$CMD 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
