#!/usr/bin/env python3
"""tests/golden/g9_gradient_gates.json from a measuring run of the G9 reference-autograd tests:
    GCRNN_TOL_REPORT=/tmp/tol.txt python -m pytest tests/test_fused.py -m gpu -q -k bptt_matches_reference_autograd_fixture
    python3 tools/make_gradient_gates.py /tmp/tol.txt
One entry per (fixture, loss, parameter): [max error / max |gradient|, mean error / max |gradient|] as measured on an MI355X; the test gates each
parameter at 2x its own entry (one pair of numbers for every parameter was 10-50x loose for most of them)."""
import json, os, re, sys
out = {}
for line in open(sys.argv[1]):
    m = re.match(r'(g9 \S+ \S+ \S+) max (\S+) mean (\S+)', line)
    if m and not m.group(1).endswith((' dX', ' dh0')):
        out[m.group(1)] = [float(m.group(2)), float(m.group(3))]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'g9_gradient_gates.json')
json.dump(out, open(dst, 'w'), indent=0, sort_keys=True)
print(len(out), 'entries ->', dst)
