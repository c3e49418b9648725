"""Where the genes->genomes mapping (get_map_bsn, PEPPAN.py:907-989) spends its time: N genomes x G exemplar genes, batched GPU search
then the per-genome host bookkeeping.  usage: python tools/map_bsn_timing.py [n_genes] [n_genomes] [genomes_per_batch]"""
import sys, time, io, contextlib, os, tempfile, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import mapbsn, synth
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
nG = int(sys.argv[2]) if len(sys.argv) > 2 else 32
per = int(sys.argv[3]) if len(sys.argv) > 3 else 32
names, seqs = synth.make_genes(ng, 0, seed=11)
os.chdir(tempfile.mkdtemp())
with open('m.clust.exemplar', 'w') as f:
    for i, s in enumerate(seqs): f.write('>%d\n%s\n' % (i, s.decode()))
worlds = synth.make_genomes(seqs, nG, seed=5)
genomes, tot = {}, 0
with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
    for g, (gname, contig, ann) in enumerate(worlds):
        genomes[100000 + g] = [900000 + g, contig.decode()]; tot += len(contig)
        op.save(100000 + g, np.array([[k, s, e, st, 1] for k, s, e, st in ann], dtype=object))
np.save('m.self_bsn.npy', np.array([[0, 1, 9000]], dtype=int))
params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
              match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
print('exemplars', ng, 'genomes', nG, 'genome nt total %.1f M' % (tot / 1e6))
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    with contextlib.redirect_stderr(io.StringIO()):
        with mapbsn.MapBsn('t.npz', 'w') as c0, mapbsn.MapBsn('s.npz', 'w') as c1, mapbsn.MapBsn('m.npz', 'w') as c2, mapbsn.MapBsn('c.npz', 'w') as c3:
            tm = {}
            if rep: pr.enable()
            mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', c0, c1, c2, c3, True, params,
                               search=lambda *a: mapbsn._gpu_search(*a, genomes_per_batch=per), timing=tm)
            if rep: pr.disable()
            t_in = time.perf_counter() - t0
    dt = time.perf_counter() - t0
    print('rep', rep, 'seconds %.2f  -> %.1f genomes/s' % (dt, nG / dt), 'inside %.2f, closing the archives %.2f;' % (t_in, dt - t_in), {k: round(v, 2) for k, v in tm.items()})
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.print_callers("acquire")
