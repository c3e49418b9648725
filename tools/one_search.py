"""a few searches of the headline's genes for the profiler: python tools/one_search.py [n_genes] [tool] [searches]
tool: diamond (default; the translated search, K1 inside) | blastn (the nucleotide tool: both sets packed as base codes, both strands, 17-mers)"""
import sys
sys.path.insert(0, '.')
from peppan_amd import _native as N, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
tool = sys.argv[2] if len(sys.argv) > 2 else 'diamond'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
flag = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # params.reserved[0]: 8 = the nucleotide tool's plain matcher, 10 = no self-hit shortcut
names, seqs = synth.make_genes(n, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
seqs = [seqs[i] for i in order]
ctx = N.Context(0)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
p = N.nucleotide_params(45., 25.) if tool == 'blastn' else N.default_params(45., 25., 10, 5)
p.reserved[0] = flag
for rep in range(reps):
    if tool == 'blastn':
        ctx.use_nt_as_residues(2)
    else:
        ctx.invalidate_translation()
    h, c, st = ctx.search(p)
print(len(h), st)
