import sys, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth
names, seqs = synth.make_genes(10000, 1002, seed=355)
ctx = N.Context(0)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
for dbg in (0, 1, 2, 3, 0):
    p = N.default_params(45., 25., 10, 5)
    p.reserved[0] = dbg
    for rep in range(3):
        h, c, st = ctx.search(p)
    print(dbg, 'ms_seed %.2f ms_sw %.2f tr %.2f trace %.2f total %.2f' % (st['ms_seed'], st['ms_sw'], st['ms_sw_trace'], st['ms_trace'], st['ms_total']), st['target_seeds'], st['seed_hits'], st['seed_hits_passed'], st['candidates'], len(h))
