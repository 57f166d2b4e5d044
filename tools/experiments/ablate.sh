# Phase ablation of the fused step kernel (profiling only; ablated builds compute wrong results by construction).
# usage on the GPU box: bash tools/ablate.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/abl
VARIANTS="full nohops nophase1 neither nohops_noloads nohops_nomfma"
for v in $VARIANTS; do
  case $v in full) D="";; nohops) D="-DGCRNN_ABLATE_HOPS";; nophase1) D="-DGCRNN_ABLATE_PHASE1";; neither) D="-DGCRNN_ABLATE_HOPS -DGCRNN_ABLATE_PHASE1";; nohops_noloads) D="-DGCRNN_ABLATE_HOPS -DGCRNN_ABLATE_P1_LOADS";; nohops_nomfma) D="-DGCRNN_ABLATE_HOPS -DGCRNN_ABLATE_P1_MFMA";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $D -o /tmp/abl/lib_$v.so $C/*.hip $C/gcrnn_host.cpp &
done
wait
for v in $VARIANTS; do
  echo -n "$v: "; GCRNN_LIBPATH=/tmp/abl/lib_$v.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
done
