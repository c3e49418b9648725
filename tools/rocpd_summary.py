#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel table:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('%-58s %7s %12s %12s %12s %12s %6s %5s %5s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'sgpr', 'lds'))
    for n, k, s, a, mn, mx, v, sg, l in rows:
        n = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        print('%-58s %7d %12.1f %12.2f %12.2f %12.2f %6.2f %5s %5s %7s' % (n[:58], k, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100. * s / tot, v, sg, l))
    try:
        pm = c.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name").fetchall()
    except Exception:
        pm = []
    if pm:
        print('\ncounters (per-dispatch average = sum over all counter instances / dispatches):')
        for r in pm:
            print('  %-50s %-28s %18.0f %6d' % (str(r[0]).replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:50], r[1], r[2] / max(1, r[3]), r[3]))


if __name__ == '__main__':
    main(sys.argv[1])
