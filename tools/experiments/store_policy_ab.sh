# A/B of the state stores' cache policy on one box: sc1 (default) vs plain vs nt.  usage on the GPU box: bash tools/store_policy_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/spab
for p in 0 2; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGCRNN_STORE_POLICY=$p -o /tmp/spab/lib_p$p.so $C/*.hip $C/gcrnn_host.cpp & done
wait
for rep in 1 2 3; do
  echo -n "sc1   : "; python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
  echo -n "plain : "; GCRNN_LIBPATH=/tmp/spab/lib_p0.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
  echo -n "nt    : "; GCRNN_LIBPATH=/tmp/spab/lib_p2.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
done
