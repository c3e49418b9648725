import sys, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth, dist as pdist
names, seqs = synth.make_genes(10000, 1002, seed=355)
ctx = N.Context(0)
ctx.set_timing(2)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
p = N.default_params(45., 25., 10, 5)
gene_of_target = None
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.) + time.perf_counter() - t0
for rep in range(12):
    if rep == 2: T.clear()
    t0 = time.perf_counter(); ctx.translate(force=True); tick('translate', t0)
    t0 = time.perf_counter(); hits, cig, st = ctx.search(p, copy=False); tick('search', t0)
    if gene_of_target is None: gene_of_target = ctx.target_meta()['seq'].astype(np.uint32)
    t0 = time.perf_counter(); allh, allc = pdist.allgather_hits(hits, cig, 0); tick('allgather(n=1)', t0)
    t0 = time.perf_counter(); lab = ctx.components_of_hits(len(seqs), allh, gene_of_target); tick('components_of_hits', t0)
    T['gpu_total_events'] = T.get('gpu_total_events', 0.) + st['ms_total'] / 1e3
print({k: round(v / 10 * 1e3, 3) for k, v in T.items()})
