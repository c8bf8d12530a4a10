#!/usr/bin/env python3
"""Development aid: per-KERNEL sums of the counters of a rocprofv3 --pmc run (rocpd database): dispatches, total duration, every counter summed over the
kernel's dispatches (and over XCDs / instances), plus two derived columns.  Usage: pmc_aggregate.py results.db [dispatches-to-skip-per-kernel]"""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = n.split('(')[0] if len(n) > 100 else n
    return n.replace('void ', '').replace('s2::', '')[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    dur = defaultdict(list)
    for n, s, e in db.execute(f"select {name_col}, start, end from kernels"):
        dur[short(n)].append(e - s)
    ccols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    kn = 'kernel_name' if 'kernel_name' in ccols else 'name'
    cn = 'counter_name' if 'counter_name' in ccols else 'pmc_name'
    cnt = defaultdict(lambda: defaultdict(float))
    for k, c, v in db.execute(f"select {kn}, {cn}, value from counters_collection"):
        cnt[short(k)][c] += v
    names = sorted({c for v in cnt.values() for c in v})
    print('kernel,dispatches,total_ms,' + ','.join(names) + ',valu_per_wave_quadcycle,wait_any_fraction')
    tot = defaultdict(float)
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        c = cnt.get(k, {})
        wc = c.get('SQ_WAVE_CYCLES', 0.0)
        print('%s,%d,%.3f,' % (k, len(v), sum(v) / 1e6) + ','.join('%.4g' % c.get(n, 0.0) for n in names) +
              ',%.4f,%.4f' % ((c.get('SQ_INSTS_VALU', 0.0) / wc) if wc else 0.0, (c.get('SQ_WAIT_ANY', 0.0) / wc) if wc else 0.0))
        for n in names:
            tot[n] += c.get(n, 0.0)
    print('ALL,%d,%.3f,' % (sum(len(v) for v in dur.values()), sum(sum(v) for v in dur.values()) / 1e6) + ','.join('%.4g' % tot[n] for n in names) + ',,')


if __name__ == '__main__':
    main()
