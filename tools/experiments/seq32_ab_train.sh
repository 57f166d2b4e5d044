# Same-box A/B of compile-time switches of the wide kernel on the TRAINING line (BPTT chain: MODE 2): as tools/seq32_ab.sh.
# usage on the GPU box: bash tools/seq32_ab_train.sh "-DGCRNN_SEQ32_EP_FIRST=1" "-DGCRNN_SEQ32_EP_EARLY=1" ...
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
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"], 3), "ms per training step")'
for rep in 1 2; do
  echo -n "default: "; python3 $R/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$show"
  i=0
  for FL in "$@"; do
    i=$((i+1))
    echo -n "$FL: "; GCRNN_LIBPATH=/tmp/ab32/lib_$i.so python3 $R/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$show"
  done
done
