"""K1 over the contigs of a mapping batch: python tools/ab/k1_long.py [genomes] [genes]
Wall time of pep_translate(force) - K1 of both sides, host waits included - for `genomes` synthetic genomes of a `genes`-gene pan-genome (7.7 Mb contigs at 50 000 genes)
against 10 queries; the chunk count.  Round 6: the chunk chain of long frames by segments (k1_ref_chunks_spec / _join)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth
n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
names, seqs = synth.make_genes(genes, 1002, seed=355)
contigs = [g[1] for g in synth.make_genomes(seqs, n_genomes, seed=355, presence=synth.PAN_GENOME_PRESENCE if genes >= 50000 else None)]
ctx = N.Context(0)
ctx.set_query_nt(seqs[:10], 11)
ctx.set_ref_nt(contigs, 6, 11)
ctx.translate(force=True)
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    ctx.translate(force=True)
    ts.append((time.perf_counter() - t0) * 1e3)
tm = ctx.target_meta()
print('%d contigs, %.1f Mnt: translate (K1, both sides) median %.2f ms, fastest %.2f ms; %d chunks, longest %d' % (
    len(contigs), sum(map(len, contigs)) / 1e6, sorted(ts)[len(ts) // 2], min(ts), len(tm), int(tm['aa_len'].max())))
