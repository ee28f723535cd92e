#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 rocpd database: every kernel of the LAST complete step with its start
offset, duration, queue (stream) and the gap since the previous kernel on the same queue ended.
    tools/rocpd_timeline.py results.db [first_kernel_substring] > profiles/xxx.txt"""
import re
import sqlite3
import sys

db = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else 'prep_moments_kernel'
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % disp)]
scols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
name_col = 'display_name' if 'display_name' in scols else 'kernel_name'
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
q = 'select s.%s, d.start, d.end, %s from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, 'd.' + qcol if qcol else '0', disp, sym)
rows = [(re.sub(r'\s+', ' ', n), a, b, qq) for n, a, b, qq in c.execute(q)]
starts = [i for i, r in enumerate(rows) if first in r[0]]
if len(starts) < 3:
    sys.exit('fewer than three steps in the trace')
i0, i1 = starts[-2], starts[-1]
step = rows[i0:i1]
t0 = step[0][1]
last_end = {}
print('step of %d kernels, %.1f us from the first kernel start to the next step\'s first kernel start' % (len(step), (rows[i1][1] - t0) / 1e3))
print('%9s %9s %8s %5s  %s' % ('start_us', 'dur_us', 'gap_us', 'queue', 'kernel'))
busy = 0
for n, a, b, qq in step:
    gap = (a - last_end[qq]) / 1e3 if qq in last_end else 0.0
    last_end[qq] = b
    short = re.sub(r'\(.*', '', n.replace('scs::', '').replace('void ', ''))[:90]
    print('%9.1f %9.1f %8.1f %5s  %s' % ((a - t0) / 1e3, (b - a) / 1e3, gap, qq, short))
