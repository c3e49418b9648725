"""Which part of a mapping round gets slower when N processes share the GPU?  N copies of this script (started by the first) each map the same
16 synthetic genomes R times (batched search of both tools + filters, no groups, no stores); copy 0 runs under cProfile and prints its top functions.
usage: python tools/ab/shared_gpu_search.py N [rounds]"""
import sys, os, time, subprocess, io, contextlib
sys.path.insert(0, '.')
n_proc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
me = int(os.environ.get('SHARED_RANK', '0'))
kids = []
if me == 0:
    for r in range(1, n_proc):
        kids.append(subprocess.Popen([sys.executable, __file__, str(n_proc), str(R)], env=dict(os.environ, SHARED_RANK=str(r)), stdout=subprocess.DEVNULL))
import numpy as np
from peppan_amd import mapbsn, synth
import tempfile
names, seqs = synth.make_genes(10000, 0, seed=355)
os.chdir(tempfile.mkdtemp())
with open('m.clust.exemplar', 'w') as f:
    for i, q in enumerate(seqs): f.write('>%d\n%s\n' % (i, q.decode()))
jobs = [(g, 900000 + g, [[100000 + g, contig.decode()]]) for g, (gname, contig, ann) in enumerate(synth.make_genomes(seqs, 16, seed=355 + me))]
params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
              match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
def one():
    with contextlib.redirect_stderr(io.StringIO()):
        return sum(len(t) for t, o in mapbsn._gpu_search('m', 'm.clust.exemplar', jobs, params, genomes_per_batch=16))
one(); one()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
if me == 0: pr.enable()
for _ in range(R): rows = one()
if me == 0: pr.disable()
dt = time.perf_counter() - t0
if me == 0:
    for k in kids: k.wait()
    print('%d process(es): %.1f ms per genome in process 0 (%d rows per round)' % (n_proc, dt / R / 16 * 1e3, rows))
    st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(14)
