#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: python3 tools/pmc_summary.py gpurun_out/<dir> [kernel-substring]"""
import csv, glob, collections, sys
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else 'fused_step'
tot = collections.OrderedDict()
for f in sorted(glob.glob(d + '/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        tot[k] = (sum(v) / len(v), len(v))
for k, (v, n) in tot.items():
    print('%-28s %16.0f  (avg over %d launches)' % (k, v, n))
