"""seed_match decomposition with the stage limiter (params.reserved[0]): 1 = keys only, 2 = + filter and start[], 3 = + entries compare, 0 = all.
    rocprofv3 --kernel-trace --stats -d out -o x -- python3 tools/seed_stages.py <stage>"""
import sys
sys.path.insert(0, '.')
from peppan_amd import _native as N, synth
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
names, seqs = synth.make_genes(10000, 1002, seed=355)
ctx = N.Context(0)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
p = N.default_params(45., 25., 10, 5)
p.reserved[0] = stage
for rep in range(3):
    try:
        h, c, st = ctx.search(p)
    except Exception as e:
        print('stage', stage, 'search ended with', str(e)[:80])
print('done', stage)
