#!/usr/bin/env python3
"""Condense a rocprofv3 rocpd database (sqlite) into a small text summary for profiles/.

usage: rocpd_summary.py <results.db> [counter-name]
  * kernel table: calls, total/avg/min/max duration (us), share -- what `--stats` reports
  * if the run collected a PMC counter: per-kernel mean of that counter per dispatch
"""
import sqlite3
import sys


def short(name):
    name = name.replace('void ', '')
    return name if len(name) < 150 else name[:147] + '...'


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('# kernel-trace summary of %s' % db)
    print('%-8s %12s %12s %12s %12s %7s  %s' % ('calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'kernel'))
    for name, n, s, a, mn, mx in rows[:40]:
        print('%-8d %12.1f %12.1f %12.1f %12.1f %6.2f%%  %s' % (n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100 * s / tot, short(name)))
    try:
        crow = cur.execute("select counter_name, kernel_name, count(*), avg(value), min(value), max(value) "
                           "from counters_collection group by counter_name, kernel_name "
                           "order by sum(value) desc").fetchall()
    except sqlite3.Error:
        crow = []
    if crow:
        print('\n# PMC counters, per dispatch')
        print('%-12s %-8s %16s %16s %16s  %s' % ('counter', 'calls', 'mean', 'min', 'max', 'kernel'))
        # this library's kernels first (a counter that reads 0 would otherwise fall off the list)
        crow.sort(key=lambda r: 0 if 'gfft::' in r[1] else 1)
        for cn, kn, n, a, mn, mx in crow[:80]:
            print('%-12s %-8d %16.1f %16.1f %16.1f  %s' % (cn, n, a, mn, mx, short(kn)))


if __name__ == '__main__':
    main()
