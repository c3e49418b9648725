#!/usr/bin/env python3
"""rocprofv3 PMC databases (one per counter pass) -> profiles/rNN_traffic.json (per-kernel per-dispatch averages, raw counter unit KiB)
    python tools/pmc_to_json.py out.json fetch_results.db write_results.db"""
import json
import sqlite3
import sys


def short(n):
    return str(n).replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def main(out, *dbs):
    kernels = {}
    for path in dbs:
        c = sqlite3.connect(path)
        for name, ctr, total, n in c.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name"):
            kernels.setdefault(short(name), {})[ctr] = int(round(total / max(1, n)))
    json.dump({'workload': '10000 genes x 1002 nt all-vs-all', 'unit': 'KiB per dispatch (raw counter)', 'kernels': kernels}, open(out, 'w'), indent=1)
    print('wrote', out, len(kernels), 'kernels')


if __name__ == '__main__':
    main(*sys.argv[1:])
