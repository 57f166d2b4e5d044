# Un-profiled bench lines of every mode the docs quote (on the GPU box, from the repo root): bash tools/bench_lines.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/lines_$TAG
mkdir -p $O
line() { n=$1; shift; python3 $R/bench.py "$@" --no-cpu-baseline > $O/${TAG}_bench_$n.json 2> $O/$n.err; python3 -c "import json;d=json.load(open('$O/${TAG}_bench_$n.json'));print('%-22s %10.1f %s  %8.3f ms' % ('$n', d['value'], d['unit'], d['ms_per_step']))"; }
line fwd_timegated --time-gating
line fwd_nodegated --spatial-gating node
line fwd_edgegated --spatial-gating edge
line fwd_timeedge --spatial-gating edge --time-gating
line train_bf16 --mode train
line train_timegated --mode train --time-gating
line train_nodegated --mode train --spatial-gating node
line train_edgegated --mode train --spatial-gating edge
line train_timeedge --mode train --spatial-gating edge --time-gating
line f32_x3 --dtype f32
line f32_x3_timegated --dtype f32 --time-gating --steps 5 --warmup 2
line f64_b128 --dtype f64 --batch 128
line cfg4 --config cfg4
line cfg5_bf16 --config cfg5
line cfg5_f32 --config cfg5 --dtype f32
python3 $R/bench.py > $O/${TAG}_bench_default.json 2> $O/default.err
python3 -c "import json;d=json.load(open('$O/${TAG}_bench_default.json'));print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"
