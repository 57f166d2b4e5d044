# A/B of the stream's code size: the D phases laid out R times per loop iteration (GCRNN_HOP16_UNROLL=R; R = 1 halves the stream's code and
# doubles its taken branches): bash tools/hop16_unroll_ab.sh "1 2"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for d in ${1:-1 2}; do
  rm -rf /tmp/hu$d && mkdir -p /tmp/hu$d/pkg /tmp/hu$d/include && cp -r $R/gated_gcrnns_amd/csrc /tmp/hu$d/pkg/csrc && cp $R/include/gcrnn.h /tmp/hu$d/include/
  GCRNN_HOP16_UNROLL=$d python3 $R/tools/gen_hop_asm.py > /tmp/hu$d/pkg/csrc/gcrnn_hop_asm.inc
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -o /tmp/hu$d/lib.so /tmp/hu$d/pkg/csrc/*.hip /tmp/hu$d/pkg/csrc/gcrnn_host.cpp &
done
wait
for rep in 1 2; do for d in ${1:-1 2}; do
  echo -n "unroll $d: "; GCRNN_LIBPATH=/tmp/hu$d/lib.so python3 $R/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_avg_us'], d['roofline']['inline_pack']['bare_kernel_avg_us'])"
done; done
