#!/usr/bin/env python3
"""Timeline of the LAST search step in a rocprofv3 rocpd database: every kernel with its start offset, duration and the idle gap
before it, plus the gaps summed per following kernel.  python tools/rocpd_gaps.py x_results.db [anchor-kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]


def main(path, anchor='k1_query_frames'):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(starts) < 3:
        print('anchor kernel not found often enough'); return
    lo, hi = starts[-2], starts[-1]                      # one full step: anchor .. next anchor
    step = rows[lo:hi]
    t0 = step[0][1]
    busy, gaps, prev_end = 0, defaultdict(float), step[0][1]
    print('%-46s %10s %10s %10s' % ('kernel', 'start_us', 'dur_us', 'gap_us'))
    for n, s, e in step:
        gap = (s - prev_end) / 1e3
        print('%-46s %10.1f %10.1f %10.1f' % (short(n), (s - t0) / 1e3, (e - s) / 1e3, gap))
        gaps[short(n)] += max(gap, 0)
        busy += e - s
        prev_end = max(prev_end, e)
    span = (rows[hi][1] - t0) / 1e3
    print('\nstep span %.1f us, kernels busy %.1f us, idle %.1f us (%d launches)' % (span, busy / 1e3, span - busy / 1e3, len(step)))
    print('idle before kernel (top 15):')
    for k, v in sorted(gaps.items(), key=lambda x: -x[1])[:15]:
        print('  %-46s %8.1f' % (k, v))


if __name__ == '__main__':
    main(*sys.argv[1:])
