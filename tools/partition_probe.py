"""Measurement behind DESIGN.md section 8 (partitioned join): params.reserved[0] = 7 makes the seed stage run, behind the matcher of the first
seed shape, the scatter pass a partitioned join would need (tgt_slab_probe: target keys that pass the filter -> the 2 048 coarse buckets of the
query index).  Run under rocprofv3 --kernel-trace --stats and compare tgt_slab_probe with seed_match.  python3 tools/partition_probe.py [n_genes]"""
import sys
sys.path.insert(0, '.')
from peppan_amd import _native as N, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
names, seqs = synth.make_genes(n, 1002, seed=355)
ctx = N.Context(0)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
p = N.default_params(45., 25., 10, 5)
p.reserved[0] = 7
for rep in range(3):
    h, c, st = ctx.search(p)
print(n, 'genes:', len(h), 'hits,', st['target_seeds'], 'target seeds,', st['seed_hits'], 'raw seed hits')
