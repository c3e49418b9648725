"""The passes of the hot call's host chain (csrc/hostchain.hip) on a table of the hot call's size - 52 000 nucleotide hits, 17 columns - by thread count:
pep_table_from_hits, pep_cols_fix_end, pep_cols_order, pep_cols_gather; best of 15 calls each, in ms.  No GPU work.  usage: python tools/host_chain_passes.py"""
import sys, time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
from peppan_amd import _native as N
from test_host_helpers import _random_hits
rng = np.random.default_rng(5)
n, n_q, n_t = 52000, 10000, 20000
q_len, r_len = rng.integers(900, 1100, n_q), rng.integers(900, 1100, n_t)
h, cigar = _random_hits(rng, n, n_q, n_t, 1)
t_seq, t_rev = np.arange(n_t) % 10000, np.arange(n_t) >= 10000
for th in (1, 2, 3, 4, 8):
    N.set_host_threads(th)
    best = [1e9] * 4
    for rep in range(15):
        t0 = time.perf_counter(); cols, arena = N.table_from_hits(1, h, cigar, q_len, r_len, 0.0, 0., 0., t_seq=t_seq, t_rev=t_rev); t1 = time.perf_counter()
        work = {k: np.ascontiguousarray(v.copy()) for k, v in cols.items()}
        t2 = time.perf_counter(); N.cols_fix_end(work, arena, 6., 6.); t3 = time.perf_counter()
        o = N.cols_order(work['qi'], work['ri'], work['score']); t4 = time.perf_counter()
        names = sorted(work); g = N.cols_gather([work[k] for k in names] + [work[k] for k in names[:5]], o); t5 = time.perf_counter()
        for k, v in enumerate((t1 - t0, t3 - t2, t4 - t3, t5 - t4)):
            best[k] = min(best[k], v)
    print('%d threads: table_from_hits %.2f, cols_fix_end %.2f, cols_order %.2f, cols_gather (22 columns) %.2f ms' % ((th,) + tuple(b * 1e3 for b in best)), flush=True)
