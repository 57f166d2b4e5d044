#!/usr/bin/env python3
"""The last N kernel dispatches of a rocprofv3 --kernel-trace run (rocpd database), in start order: start offset, duration and the gap to the
previous kernel's end, in microseconds:  python3 tools/rocprof_db_timeline.py <dir or .db> [N]"""
import glob, os, sqlite3, sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, '**', '*_results.db'), recursive=True))[0]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev = None
for name, s, e in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print('%10.1f us  dur %9.1f us  gap %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, name[:90]))
    prev = e
