# A/B of the cross-item fragment prefetch depth (-DGCRNN_P1_AHEAD=n tiles requested during the last hop), same box.
# usage on the GPU box: bash tools/p1_ahead_ab.sh "0 2 4 6"
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/pab
for n in ${1:-0 2 4 6}; do
  ( for f in $C/*.hip $C/gcrnn_host.cpp; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DGCRNN_P1_AHEAD=$n -c $f -o /tmp/pab/$(basename $f).$n.o & done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pab/lib_$n.so /tmp/pab/*.$n.o ) 2>&1 | grep -E "error" | head -3
done
for rep in 1 2; do
  for n in ${1:-0 2 4 6}; do
    echo -n "ahead=$n: "; GCRNN_LIBPATH=/tmp/pab/lib_$n.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
  done
done
