"""The store half of get_map_bsn alone: the groups of N genomes are computed first (GPU search + filters + build_groups), then handed to the
stores on one thread under cProfile.  usage: python tools/map_store_profile.py [n_genes] [n_genomes]"""
import sys, time, io, contextlib, os, tempfile, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import mapbsn, synth
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nG = int(sys.argv[2]) if len(sys.argv) > 2 else 64
names, seqs = synth.make_genes(ng, 0, seed=11)
os.chdir(tempfile.mkdtemp())
with open('m.clust.exemplar', 'w') as f:
    for i, s in enumerate(seqs): f.write('>%d\n%s\n' % (i, s.decode()))
worlds = synth.make_genomes(seqs, nG, seed=5)
jobs = []
with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
    for g, (gname, contig, ann) in enumerate(worlds):
        jobs.append((g, 900000 + g, [[100000 + g, contig.decode()]]))
        op.save(100000 + g, np.array([[k, s, e, st, 1] for k, s, e, st in ann], dtype=object))
params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
              match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
ortho = mapbsn.OrthoRelation(np.array([[0, 1, 9000]], dtype=int))
with contextlib.redirect_stderr(io.StringIO()):
    Gs = [mapbsn.build_groups(tab, ovl, job[2], ortho, 'm.old_prediction.npz', params) for job, (tab, ovl) in zip(jobs, mapbsn._gpu_search('m', 'm.clust.exemplar', jobs, params, genomes_per_batch=64))]
print('genomes', nG, 'groups per genome %.0f, stored rows per genome %.0f, conflicts per genome %.0f' % (np.mean([len(G) for G in Gs]), np.mean([len(G.rows) for G in Gs]), np.mean([len(G.ovl) for G in Gs])))
for rep in range(2):
    pr = cProfile.Profile()
    with mapbsn.MapBsn('t.npz', 'w') as c0, mapbsn.MapBsn('s.npz', 'w') as c1, mapbsn.MapBsn('m.npz', 'w') as c2, mapbsn.MapBsn('c.npz', 'w') as c3:
        w = mapbsn._StoreWriter(c0, c1, c2, c3, True)
        ts = []
        pr.enable()
        for G in Gs:
            t0 = time.perf_counter(); w.add(G, 5); ts.append(time.perf_counter() - t0)
        pr.disable()
        t0 = time.perf_counter(); w.close(); t_close = time.perf_counter() - t0
        t0 = time.perf_counter()
    print('rep', rep, 'add: mean %.1f ms, median %.1f, max %.1f; close %.2f s; archives %.2f s' % (np.mean(ts) * 1e3, np.median(ts) * 1e3, np.max(ts) * 1e3, t_close, time.perf_counter() - t0))
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(14)
