# WRITE_SIZE against known stored bytes in the state stores' access patterns (tools/probes/write_size_probe.hip). On the GPU box: bash tools/write_size_calibration.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/probes/write_size_probe.hip -o /tmp/wsp || exit 1
rm -rf /tmp/wsp_out
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/wsp_out -- /tmp/wsp
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/wsp_out/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'WRITE_SIZE':
            acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
stored = 1 << 20      # KiB per launch
for k, v in acc.items():
    print('%-12s WRITE_SIZE %.0f KiB (avg over %d launches) = %.3f x the %d KiB stored' % (k, sum(v) / len(v), len(v), sum(v) / len(v) / stored, stored))
PY
