# Timing experiment (WRONG results): the bf16-image hop stream with tile exits that only switch tiles (no wait states, no acc += w sum):
# what do the exits cost?  bash tools/hop16_exit_experiment.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in base cheap; do
  rm -rf /tmp/he$v && mkdir -p /tmp/he$v/pkg /tmp/he$v/include && cp -r $R/gated_gcrnns_amd/csrc /tmp/he$v/pkg/csrc && cp $R/include/gcrnn.h /tmp/he$v/include/
  if [ $v = cheap ]; then GCRNN_HOP16_EXPERIMENT_CHEAP_EXIT=1 python3 $R/tools/gen_hop_asm.py > /tmp/he$v/pkg/csrc/gcrnn_hop_asm.inc; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -o /tmp/he$v/lib.so /tmp/he$v/pkg/csrc/*.hip /tmp/he$v/pkg/csrc/gcrnn_host.cpp &
done
wait
for rep in 1 2; do for v in base cheap; do
  echo -n "$v: "; GCRNN_LIBPATH=/tmp/he$v/lib.so python3 $R/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_avg_us'], d['roofline']['inline_pack']['bare_kernel_avg_us'])"
done; done
