"""module-level pieces the mapping workers of tests/test_mapbsn_golden.py import by reference"""
import numpy as np

from conftest import load_golden


def canned_search(prefix, clust, jobs, params):
    """the search of G15: the reference's own tables for the three genomes of tests/golden/g14_mapbsn.json, by job id"""
    g = load_golden('g14_mapbsn.json')
    for id, taxon, seq in jobs:
        case = g['cases'][id]
        table = np.empty([len(case['table']), 17], dtype=object)
        for i, r in enumerate(case['table']):
            table[i, :] = r
        yield table, np.array(case['overlap'], dtype=int).reshape(-1, 3)


def sleepy_search(prefix, clust, jobs, params):
    """canned_search for a worker that HANGS: while `<prefix>.hang_once` exists the first round that sees it removes it and sleeps for a
    minute; while `<prefix>.hang_always` exists every round sleeps"""
    import os
    import time
    once = prefix + '.hang_once'
    try:
        os.unlink(once)
        time.sleep(60)
    except FileNotFoundError:
        pass
    if os.path.exists(prefix + '.hang_always'):
        time.sleep(60)
    for r in canned_search(prefix, clust, jobs, params):
        yield r
