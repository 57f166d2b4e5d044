# as tools/seq32_ab.sh, for the time-gated forward line (gate-pair pre-pass + gated recurrence)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
L=$R/gated_gcrnns_amd/lib
mkdir -p /tmp/ab32
i=0
for FL in "$@"; do
  i=$((i+1))
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC $FL -c $C/gcrnn_fused_seq32.hip -o /tmp/ab32/v$i.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab32/lib_$i.so /tmp/ab32/v$i.o $(ls $L/*.o | grep -v gcrnn_fused_seq32) ) 2>&1 | grep -E "error" | head -3 &
done
wait
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"], 3))'
for rep in 1 2; do
  echo -n "default: "; python3 $R/bench.py ${BENCH_ARGS:---time-gating} --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$show"
  i=0
  for FL in "$@"; do
    i=$((i+1))
    echo -n "$FL: "; GCRNN_LIBPATH=/tmp/ab32/lib_$i.so python3 $R/bench.py ${BENCH_ARGS:---time-gating} --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$show"
  done
done
