"""seed_match decomposition with the stage limiter (params.reserved[0]): 1 = keys only, 2 = + filter and start[], 3 = + entry compares, 0 = all.
Prints the HIP-event time of the seed_match launches of one search (all shapes).  python3 tools/seed_stages.py [n_genes]"""
import sys
sys.path.insert(0, '.')
from peppan_amd import _native as N, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
names, seqs = synth.make_genes(n, 1002, seed=355)
ctx = N.Context(0)
ctx.set_timing(2)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
for stage in (1, 2, 3, 0):
    p = N.default_params(45., 25., 10, 5)
    p.reserved[0] = stage
    best, extra = None, ''
    for rep in range(4):
        try:
            h, c, st = ctx.search(p)
            best = st['ms_seed_match'] if best is None else min(best, st['ms_seed_match'])
            extra = 'ms_seed %.3f' % st['ms_seed']
        except Exception as e:
            extra = 'search ended with ' + str(e)[:60]
    print('stage %d: seed_match %.3f ms  %s' % (stage, best if best is not None else -1, extra), flush=True)
