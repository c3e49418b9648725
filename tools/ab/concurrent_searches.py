"""Do independent searches overlap on one GPU?  T host threads, each with a context (= stream, work space) of its own, run the bench step
(10 000 genes all-vs-all by default) K times each; printed: steps per second of the whole process for T = 1, 2, 3, 4.
The seed stage waits for memory (VALU issue 0.2) and the alignment passes for the VALU (0.9, no memory traffic): on paper they fit beside each other.
usage: python tools/ab/concurrent_searches.py [n_genes] [steps]"""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import synth, _native as N, dist as pdist

n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
names, seqs = synth.make_genes(n_genes, 0, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
nts = [seqs[i] for i in order]

def worker_setup():
    ctx = N.Context(0)
    params = N.default_params(45.0, 25.0, 10, 5)
    sh = pdist.ShardedSearch(ctx, nts, nts, params, 0, 1)
    ctx.set_grouping(len(nts), sh.gene_of_target)
    for _ in range(3):
        sh.search(retranslate=True, copy=False)
    return ctx, sh

workers = [worker_setup() for _ in range(4)]
ref = None
for T in (1, 2, 3, 4, 1):
    go = threading.Barrier(T + 1)
    out = [None] * T
    def run(i):
        ctx, sh = workers[i]
        go.wait()
        for _ in range(K):
            h, c, st = sh.search(retranslate=True, copy=False)
        out[i] = (len(h), int(len(np.unique(ctx.labels))))
        go.wait()
    th = [threading.Thread(target=run, args=(i,)) for i in range(T)]
    for t in th: t.start()
    go.wait(); t0 = time.perf_counter()
    go.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    ref = ref or out[0]
    assert all(o == ref for o in out), out
    print('threads %d: %.3f ms per step of the whole process (%.1f steps/s), hits %d clusters %d' % (T, dt / (T * K) * 1e3, T * K / dt, ref[0], ref[1]), flush=True)
