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
