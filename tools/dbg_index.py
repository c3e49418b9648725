import sys
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth
for n in (10000, 700, 30000):
    names, seqs = synth.make_genes(n, 0 if n != 10000 else 1002, seed=355)
    ctx = N.Context(0)
    ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
    out = {}
    for flag in (0, 1):
        p = N.default_params(45., 25., 10, 5)
        p.reserved[2] = flag
        for rep in range(2):
            h, c, st = ctx.search(p)
        out[flag] = (h.tobytes(), c.tobytes(), st['query_seeds'], st['target_seeds'], st['seed_hits'], st['candidates'])
        print(n, 'plain' if flag else 'partition', 'ms_seed %.3f' % st['ms_seed'], st['query_seeds'], st['seed_hits'], st['candidates'], len(h))
    assert out[0] == out[1]
    ctx.close()
print('partition build == plain build')
