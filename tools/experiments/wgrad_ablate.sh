# Phase ablation of the fused weight-gradient kernel (profiling builds; ablated ones compute wrong results):
#   bash tools/wgrad_ablate.sh      (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/wgab
for v in full nohop norefetch nohop_norefetch; do
  case $v in full) D="";; nohop) D="-DGCRNN_WGRAD_ABLATE_HOP";; norefetch) D="-DGCRNN_WGRAD_ABLATE_REFETCH";; nohop_norefetch) D="-DGCRNN_WGRAD_ABLATE_HOP -DGCRNN_WGRAD_ABLATE_REFETCH";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $D -o /tmp/wgab/lib_$v.so $C/*.hip $C/gcrnn_host.cpp &
done
wait
for v in full nohop norefetch nohop_norefetch; do
  echo -n "$v: "; GCRNN_LIBPATH=/tmp/wgab/lib_$v.so python3 $R/tools/wgrad_probe.py 256 32 3 2>&1 | tail -1
done
