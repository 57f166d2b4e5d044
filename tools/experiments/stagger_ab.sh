# Experiment: does de-synchronising the CUs (half of the workgroups start GCRNN_STAGGER_US late) pay for itself inside one launch?
# usage on the GPU box: bash tools/stagger_ab.sh "0 6 12"
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/stg
for v in ${1:-0 6 12}; do
  D=""; [ "$v" != "0" ] && D="-DGCRNN_STAGGER_US=$v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w $D -o /tmp/stg/lib_$v.so $C/*.hip $C/gcrnn_host.cpp &
done
wait
for rep in 1 2; do
for v in ${1:-0 6 12}; do
  echo -n "stagger $v us: "; GCRNN_LIBPATH=/tmp/stg/lib_$v.so python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_avg_us'])"
done
done
