# A/B of the groups in flight of the SUMMING bf16-image hop stream (GCRNN_HOP16_SUMS_DEPTH=n tools/gen_hop_asm.py): bash tools/hop16_sums_depth_ab.sh "2 3"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for d in ${1:-2 3}; do
  rm -rf /tmp/hd$d && mkdir -p /tmp/hd$d/pkg /tmp/hd$d/include && cp -r $R/gated_gcrnns_amd/csrc /tmp/hd$d/pkg/csrc && cp $R/include/gcrnn.h /tmp/hd$d/include/
  GCRNN_HOP16_SUMS_DEPTH=$d python3 $R/tools/gen_hop_asm.py > /tmp/hd$d/pkg/csrc/gcrnn_hop_asm.inc
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -o /tmp/hd$d/lib.so /tmp/hd$d/pkg/csrc/*.hip /tmp/hd$d/pkg/csrc/gcrnn_host.cpp &
done
wait
for rep in 1 2; do for d in ${1:-2 3}; do
  echo -n "depth $d: "; GCRNN_LIBPATH=/tmp/hd$d/lib.so python3 $R/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_avg_us'], d['roofline']['inline_pack']['bare_kernel_avg_us'])"
done; done
