# Phase ablation of the BPTT data-chain launch (profiling only; ablated builds compute wrong results by construction): bash tools/bptt_ablate.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/abl
VARIANTS="full nohops nophase1 neither noepipf"
for v in $VARIANTS; do
  case $v in full) D="";; nohops) D="-DGCRNN_ABLATE_HOPS";; nophase1) D="-DGCRNN_ABLATE_PHASE1";; neither) D="-DGCRNN_ABLATE_HOPS -DGCRNN_ABLATE_PHASE1";; noepipf) D="-DGCRNN_EPI_PREFETCH=0";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $D -o /tmp/abl/lib_$v.so $C/*.hip $C/gcrnn_host.cpp &
done
wait
for v in $VARIANTS; do
  for inl in 1 0; do
    echo -n "$v: "; GCRNN_LIBPATH=/tmp/abl/lib_$v.so python3 $R/tools/bptt_chain_probe.py 256 16 3 $inl 2>&1 | tail -1
  done
done
