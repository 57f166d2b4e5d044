#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share, min, max) of a rocprofv3 --kernel-trace run that wrote its default
rocpd database:  python3 tools/rocprof_db_stats.py <dir or .db> > profiles/<name>.csv
Same columns as rocprofv3's own kernel_stats.csv."""
import glob, os, sqlite3, subprocess, sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, '**', '*_results.db'), recursive=True))[0]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute("select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
                 "max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
tot = float(sum(r[2] for r in rows))


def demangle(n):
    try:
        return subprocess.run(['c++filt', n.replace('.kd', '')], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for n, calls, total, avg, mn, mx in rows:
    name = demangle(n)
    if len(name) > 160:
        name = name[:157] + '...'
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name.replace('"', "'"), calls, total, avg, 100.0 * total / tot, mn, mx))
