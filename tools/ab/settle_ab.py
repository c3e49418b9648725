"""identical pairs settled by comparison (production) against swept like any other pair (params.reserved2 bit 1), same box, alternating:
ms per search of the bench workload, of the 8x1 cell and of the 50 000-gene search, K1 inside the search."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from peppan_amd import _native as N, synth
ctx = N.Context(0)
for label, n_genes, nq, reps in (('10k x 10k', 10000, 10000, 20), ('8x1 cell', 10000, 1250, 20), ('50k x 50k', 50000, 50000, 4)):
    names, seqs = synth.make_genes(n_genes, 1002, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])
    nts = [seqs[i] for i in order]
    ctx.set_query_nt(nts[:nq], 11)
    ctx.set_ref_nt(nts, 6, 11)
    res = {0: [], 2: []}
    for rep in range(5):
        for flag in (0, 2):
            p = N.default_params(45., 25., 10, 5)
            p.reserved2 = flag
            for _ in range(2):
                ctx.invalidate_translation(); ctx.search(p, copy=False)
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.invalidate_translation()
                h, c, st = ctx.search(p, copy=False)
            res[flag].append((time.perf_counter() - t0) / reps * 1e3)
    print('%-10s settled: %.3f ms (min %.3f)   all swept: %.3f ms (min %.3f)   %d of %d candidates settled' %
          (label, sorted(res[0])[2], min(res[0]), sorted(res[2])[2], min(res[2]), st['candidates_settled'] if flag == 0 else -1, st['candidates']), flush=True)
