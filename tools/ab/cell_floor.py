"""seed_match of the 8x1 cell (1 250 queries x 10 000 reference genes) with and without its look-ups (params.reserved[0] = 1: keys only),
under rocprofv3: what an LDS-resident pre-filter could at best save.  usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/ab/cell_floor.py <debug>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from peppan_amd import _native as N, synth
names, seqs = synth.make_genes(10000, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
nts = [seqs[i] for i in order]
ctx = N.Context(0)
p = N.default_params(45., 25., 10, 5)
p.reserved[0] = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx.set_query_nt(nts[:1250], 11)
ctx.set_ref_nt(nts[:10000], 6, 11)
ctx.translate()
for _ in range(6):
    ctx.invalidate_translation()
    h, c, st = ctx.search(p, copy=False)
print(len(h), 'hits')
