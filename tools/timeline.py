#!/usr/bin/env python3
"""Development aid: per-kernel timeline of a rocprofv3 rocpd database: every dispatch of the big kernels with start / end relative to the
first one (ms), to see what overlaps what.  Usage: timeline.py results.db [min_duration_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else 'kernel_name'
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = db.execute(f"select {name_col}, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
t0 = rows[0][1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows:
    n, s, e = r[0], r[1], r[2]
    if (e - s) / 1e6 >= thr:
        print('%10.2f %10.2f %8.2f  q=%s  %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r[3] if qcol else '-', n.split('(')[0][:90]))
