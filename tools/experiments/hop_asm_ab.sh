# A/B of the hop gather stream on one box: uniform-weight asm stream (default on the bench graph) vs weighted asm stream vs the
# two-deep macro stream.  usage on the GPU box: bash tools/hop_asm_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/hab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGCRNN_HOP_ASM=0 -DGCRNN_DIAGNOSTIC_STREAMS -o /tmp/hab/lib_macro.so $C/*.hip $C/gcrnn_host.cpp &
wait
for rep in 1 2 3; do
  echo -n "asm uniform : "; python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
  echo -n "asm weighted: "; GCRNN_NO_UNIFORM=1 python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
  echo -n "macro       : "; GCRNN_NO_UNIFORM=1 GCRNN_LIBPATH=/tmp/hab/lib_macro.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
done
