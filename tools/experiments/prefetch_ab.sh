# A/B of the next-sequence L2 prefetch of the step kernel on one box: default (at the start of the hops) vs none vs before the last hop.
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/pfab
for p in 0 2; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGCRNN_PREFETCH_AT=$p -o /tmp/pfab/lib_p$p.so $C/*.hip $C/gcrnn_host.cpp & done
wait
for rep in 1 2 3; do
  echo -n "hop start : "; python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
  echo -n "none      : "; GCRNN_LIBPATH=/tmp/pfab/lib_p0.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
  echo -n "last hop  : "; GCRNN_LIBPATH=/tmp/pfab/lib_p2.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
done
