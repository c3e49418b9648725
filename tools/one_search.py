import sys
sys.path.insert(0, '.')
from peppan_amd import _native as N, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
names, seqs = synth.make_genes(n, 1002, seed=355)
ctx = N.Context(0)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
p = N.default_params(45., 25., 10, 5)
p.reserved2 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    ctx.invalidate_translation()
    h, c, st = ctx.search(p)
print(st)
