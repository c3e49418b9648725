"""round 5: the seed stage with / without the self-search shortcut at 10 000 and 50 000 genes - step time (K1 inside), seed phase, hits, equality of the tables"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth
for n in [int(x) for x in sys.argv[1:]] or [10000, 50000]:
    names, seqs = synth.make_genes(n, 1002, seed=355)
    ctx = N.Context(0)
    ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
    ctx.set_timing(2)
    tabs = {}
    for off in (8, 0, 8, 0):
        p = N.default_params(45., 25., 10, 5)
        p.reserved2 = off
        for rep in range(3):
            ctx.invalidate_translation()
            h, c, st = ctx.search(p)
        ctx.set_timing(0)
        reps = 30 if n <= 10000 else 8
        t0 = time.perf_counter()
        for rep in range(reps):
            ctx.invalidate_translation()
            h, c, st2 = ctx.search(p)
        dt = (time.perf_counter() - t0) / reps * 1e3
        ctx.set_timing(2)
        tabs[off] = (h.tobytes(), c.tobytes())
        print('%d genes, shortcut %s: step %.3f ms; ms_seed %.3f ms_seed_match %.3f ms_sw %.3f ms_total %.3f; seed_hits %d self %d target_seeds %d candidates %d hits %d'
              % (n, 'off' if off else 'ON ', dt, st['ms_seed'], st['ms_seed_match'], st['ms_sw'], st['ms_total'], st['seed_hits'], st.get('seed_hits_self', 0), st['target_seeds'], st['candidates'], len(h)), flush=True)
    print('tables identical:', tabs[0] == tabs[8])
    ctx.close()
