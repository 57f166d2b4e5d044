# Timing experiment (results are WRONG by construction): the uniform hop stream with a bf16 state image gathered 16 bytes per lane (two
# gathers per four entries) and summed on the matrix cores through a one-hot A operand (two v_mfma_f32_16x16x32_bf16 per four entries
# instead of eight v_pk_add_f32) -- an upper bound on what that hop design would buy. usage on the GPU box: bash tools/hop_mfma_experiment.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/hm && mkdir -p /tmp/hm/pkg /tmp/hm/include && cp -r $R/gated_gcrnns_amd/csrc /tmp/hm/pkg/csrc && cp $R/include/gcrnn.h /tmp/hm/include/
C=/tmp/hm/pkg/csrc
GCRNN_HOP_EXPERIMENT_MFMA=1 python3 $R/tools/gen_hop_asm.py > $C/gcrnn_hop_asm.inc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -o /tmp/hm/lib_mfma.so $C/*.hip $C/gcrnn_host.cpp 2>&1 | grep -E "error" | head
ls -la /tmp/hm/lib_mfma.so
for rep in 1 2; do
  echo -n "fp32 image, packed adds: "; python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_avg_us'], d['roofline']['inline_pack']['bare_kernel_avg_us'])"
  echo -n "bf16 image, MFMA sums:   "; GCRNN_LIBPATH=/tmp/hm/lib_mfma.so python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_avg_us'], d['roofline']['inline_pack']['bare_kernel_avg_us'])"
done
