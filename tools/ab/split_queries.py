"""One all-vs-all as TWO searches in flight on one GPU: the queries dealt to two contexts (each with the whole reference), a host thread each.
The halves' seed stages (memory-bound) and alignment passes (VALU-bound) overlap; the reference is streamed twice.  Printed: ms per whole all-vs-all.
usage: python tools/ab/split_queries.py [n_genes] [steps] [parts]"""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import synth, _native as N
n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
names, seqs = synth.make_genes(n_genes, 1002 if n_genes == 10000 else 0, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
nts = [seqs[i] for i in order]
p = N.default_params(45., 25., 10, 5)
one = N.Context(0); one.set_query_nt(nts, 11); one.set_ref_nt(nts, 6, 11)
for _ in range(3): h1, c1, st = one.search(p, copy=False)
n_hits = len(h1)
t0 = time.perf_counter()
for _ in range(K):
    one.invalidate_translation(); one.search(p, copy=False)
print('one search: %.3f ms per all-vs-all, %d hits' % ((time.perf_counter() - t0) / K * 1e3, n_hits))
bounds = [len(nts) * i // parts for i in range(parts + 1)]
ctxs = []
for a, b in zip(bounds[:-1], bounds[1:]):
    c = N.Context(0); c.set_query_nt(nts[a:b], 11); c.set_ref_nt(nts, 6, 11)
    for _ in range(3): c.search(p, copy=False)
    ctxs.append(c)
tot = [0] * parts
go = threading.Barrier(parts + 1)
def run(k):
    go.wait()
    for _ in range(K):
        go2.wait()                                 # every all-vs-all starts its parts together and ends when the last part is done
        ctxs[k].invalidate_translation(); h, c, st = ctxs[k].search(p, copy=False); tot[k] = len(h)
        go2.wait()
    go.wait()
go2 = threading.Barrier(parts)
th = [threading.Thread(target=run, args=(k,)) for k in range(parts)]
for t in th: t.start()
go.wait(); t0 = time.perf_counter(); go.wait(); dt = time.perf_counter() - t0
print('%d parts in flight: %.3f ms per all-vs-all, %d hits in all (the same: %s)' % (parts, dt / K * 1e3, sum(tot), sum(tot) == n_hits))
