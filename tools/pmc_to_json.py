#!/usr/bin/env python3
"""rocprofv3 PMC databases (one per counter pass) -> profiles/rNN_counters.json: per kernel the per-dispatch average of every counter
(raw counter unit: FETCH_SIZE / WRITE_SIZE in KiB) and the average duration of the kernel in that pass.
    python tools/pmc_to_json.py out.json "n_genes | workload text[;steps per pass]" pass1_results.db pass2_results.db ..."""
import json
import sqlite3
import sys


def short(n):
    return str(n).replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def main(out, n_genes, *dbs):
    kernels = {}
    for path in dbs:
        c = sqlite3.connect(path)
        for name, ctr, total, n in c.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name"):
            kernels.setdefault(short(name), {})[ctr] = int(round(total / max(1, n)))
        for name, avg, n in c.execute("select name, avg(duration), count(*) from kernels group by name"):
            k = kernels.setdefault(short(name), {})
            k.setdefault('avg_us_in_pmc_passes', []).append(round(avg / 1e3, 2))
            k['dispatches_per_pass'] = n
    steps = 2
    if ';' in n_genes:
        n_genes, steps = n_genes.rsplit(';', 1)
        steps = int(steps)
    workload = '%s genes x 1002 nt all-vs-all (tools/one_search.py: 2 searches per pass)' % n_genes if n_genes.strip().isdigit() else n_genes
    json.dump({'workload': workload, 'steps_per_pass': steps,
               'unit': 'per-dispatch average of the raw counter (FETCH_SIZE / WRITE_SIZE: KiB; SQ_* cycle counters: quad-cycles summed over all SIMDs; '
                       'GRBM_GUI_ACTIVE: summed over the 8 XCDs)',
               'kernels': kernels}, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, len(kernels), 'kernels')


if __name__ == '__main__':
    main(*sys.argv[1:])
