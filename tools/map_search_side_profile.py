"""The host side of a mapping round's searches, alone on one thread: python tools/map_search_side_profile.py [genomes] [per batch]
mapbsn._gpu_search (two batched searches per round, then per genome: K7 cut, -f, -m, fixEnd, -O, final order) over synthetic genomes under cProfile - what a pool
worker's search thread does with the interpreter between its library calls."""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, '.')
os.environ.setdefault('PEPPAN_LOG', '0')
import numpy as np
from peppan_amd import mapbsn, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
names, seqs = synth.make_genes(10000, 0, seed=355)
os.chdir(tempfile.mkdtemp())
with open('m.clust.exemplar', 'w') as f:
    for i, s in enumerate(seqs):
        f.write('>%d\n%s\n' % (i, s.decode()))
jobs = [(g, 900000 + g, [[100000 + g, contig.decode()]]) for g, (gname, contig, ann) in enumerate(synth.make_genomes(seqs, n, seed=355))]
params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
              match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
for _ in mapbsn._gpu_search('m', 'm.clust.exemplar', jobs[:per], params, genomes_per_batch=per):
    pass
pr = cProfile.Profile()
t0 = time.perf_counter()
c0 = time.process_time()
pr.enable()
rows = 0
for tab, ovl in mapbsn._gpu_search('m', 'm.clust.exemplar', jobs, params, genomes_per_batch=per):
    rows += len(tab)
pr.disable()
dt, cpu = time.perf_counter() - t0, time.process_time() - c0
print('%d genomes in batches of %d: %.1f ms per genome wall, %.1f ms of this process\'s CPU per genome, %d rows per genome' % (n, per, dt / n * 1e3, cpu / n * 1e3, rows // n))
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats('tottime').print_stats(32)
for l in out.getvalue().splitlines()[4:]:
    if l.strip():
        print(l[:180])
