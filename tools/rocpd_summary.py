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
    # the same kernel launched with different geometries (chunks of a pipelined stage, body / leftover
    # columns): one line per (kernel, grid size)
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
        gcol = next((c for c in ('grid_size_x', 'grid_x', 'grid_size') if c in cols), None)
        if gcol:
            grows = cur.execute("select name, %s, count(*), avg(duration), min(duration), max(duration) from kernels "
                                "where name like '%%gfft::%%' group by name, %s order by name, %s" % (gcol, gcol, gcol)).fetchall()
            if len(grows) > len([r for r in rows if 'gfft::' in r[0]]):
                print('\n# by launch geometry')
                print('%-8s %10s %12s %12s %12s  %s' % ('calls', 'grid', 'avg_us', 'min_us', 'max_us', 'kernel'))
                for name, g, n, a, mn, mx in grows[:60]:
                    print('%-8d %10d %12.1f %12.1f %12.1f  %s' % (n, g, a / 1e3, mn / 1e3, mx / 1e3, short(name)))
    except sqlite3.Error:
        pass
    try:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
        gcol = next((c for c in ('grid_size_x', 'grid_x', 'grid_size') if c in ccols), None)
        if gcol:
            crow2 = cur.execute("select counter_name, kernel_name, %s, count(*), avg(value) from counters_collection "
                                "where kernel_name like '%%gfft::%%' group by counter_name, kernel_name, %s" % (gcol, gcol)).fetchall()
            if crow2:
                print('\n# PMC counters by launch geometry, per dispatch')
                for cn, kn, g, n, a in crow2[:80]:
                    print('%-12s %-8d grid %8d %16.1f  %s' % (cn, n, g, a, short(kn)))
    except sqlite3.Error:
        pass
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
