# A/B: de-synchronised workgroup starts in the wide sequence-resident kernel (GCRNN_SEQ32_STAGGER = shader cycles per phase group)
# usage on the GPU box: bash tools/seq32_stagger_ab.sh "0 6000 12000 16000"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in ${1:-0 6000 12000 16000}; do
  echo -n "stagger $v cycles: "; GCRNN_SEQ32_STAGGER=$v python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us'], d['roofline']['inline_pack']['bare_kernel_avg_us'], d['roofline_native_layout']['kernel_avg_us'])"
done
done
