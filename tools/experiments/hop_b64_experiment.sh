# Timing experiment (results are WRONG by construction): the uniform hop stream with 8-byte gathers instead of 16-byte ones -- an upper
# bound on what a bf16 hop-state image would buy. usage on the GPU box: bash tools/hop_b64_experiment.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/hx && mkdir -p /tmp/hx/pkg /tmp/hx/include && cp -r $R/gated_gcrnns_amd/csrc /tmp/hx/pkg/csrc && cp $R/include/gcrnn.h /tmp/hx/include/
C=/tmp/hx/pkg/csrc
GCRNN_HOP_EXPERIMENT_B64=1 python3 $R/tools/gen_hop_asm.py > $C/gcrnn_hop_asm.inc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -o /tmp/hx/lib_b64.so $C/*.hip $C/gcrnn_host.cpp 2>&1 | grep -E "error" | head
ls -la /tmp/hx/lib_b64.so
for rep in 1 2; do
  echo -n "16-byte gathers: "; python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
  echo -n " 8-byte gathers: "; GCRNN_LIBPATH=/tmp/hx/lib_b64.so python3 $R/tools/step_kernel_probe.py 256 16 3 2>&1 | tail -1
done
