#!/usr/bin/env python3
"""Per-kernel statistics (calls, mean / min / max duration) from a rocprofv3 rocpd database (the --stats CSV is not
always written by this rocprofv3 build): tools/rocpd_stats.py results.db [calls_per_step] > profiles/xxx.csv"""
import re
import sqlite3
import sys

db = sys.argv[1]
per_step = float(sys.argv[2]) if len(sys.argv) > 2 else 0
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % disp)]
scols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
name_col = 'display_name' if 'display_name' in scols else 'kernel_name'
q = 'select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc' % (name_col, disp, sym, name_col)
rows = list(c.execute(q))
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for name, n, total, mn, mx in rows:
    name = re.sub(r'\s+', ' ', name)
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, n, total, total / n, 100.0 * total / tot, mn, mx))
