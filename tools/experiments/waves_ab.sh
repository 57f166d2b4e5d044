# A/B of the step kernel's waves per workgroup with phase ablations (profiling builds; ablated ones compute wrong results):
#   bash tools/waves_ab.sh      (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/wab
for w in 8 16; do for v in full nohops nophase1; do
  case $v in full) D="";; nohops) D="-DGCRNN_ABLATE_HOPS";; nophase1) D="-DGCRNN_ABLATE_PHASE1";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGCRNN_STEP_WAVES=$w -DGCRNN_DIAGNOSTIC_STREAMS $D -o /tmp/wab/lib_${w}_$v.so $C/*.hip $C/gcrnn_host.cpp &
done; done
wait
for w in 8 16; do for v in full nohops nophase1; do
  echo -n "waves=$w $v: "; GCRNN_LIBPATH=/tmp/wab/lib_${w}_$v.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
done; done
