#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (default output format of ROCm 7.2): per-kernel stats (calls, total/avg/min/max
duration) and, when the run collected counters, per-kernel counter sums per dispatch.  Usage: rocpd_summary.py results.db [> out.txt]"""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = (n.split('(')[0] or n) if len(n) > 120 else n
    return n[:160]


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
    st = defaultdict(list)
    for n, s, e in rows:
        st[short(n)].append(e - s)
    tot = sum(sum(v) for v in st.values()) or 1
    print('# kernel stats (ns) from', sys.argv[1])
    print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
    for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
        print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
    try:
        ccols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        if ccols:
            kn = 'kernel_name' if 'kernel_name' in ccols else 'name'
            cn = 'counter_name' if 'counter_name' in ccols else 'pmc_name'
            did = 'dispatch_id' if 'dispatch_id' in ccols else 'id'
            rows = db.execute(f"select {kn}, {did}, {cn}, value from counters_collection").fetchall()
            if rows:
                per = defaultdict(lambda: defaultdict(float))
                for k, d, c, v in rows:
                    per[(short(k), d)][c] += v
                print('\n# counters per dispatch (summed over all instances/dimensions)')
                print('"Kernel","DispatchId","Counter","Value"')
                for (k, d), cs in sorted(per.items(), key=lambda kv: kv[0][1]):
                    for c, v in sorted(cs.items()):
                        print('"%s",%s,"%s",%.6g' % (k, d, c, v))
    except sqlite3.Error as e:
        print('# no counters:', e)


if __name__ == '__main__':
    main()
