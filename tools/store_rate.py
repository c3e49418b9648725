"""How many genomes per second can ONE process put into the four stores (the parent's side of get_map_bsn with worker processes)?
Made-up GenomeGroups of a mapped genome's size (6 600 groups, ~2 hit rows each, packed alleles of ~110 bytes) are handed to _StoreWriter
on a keeper thread exactly as get_map_bsn does; runs without a GPU.   usage: python tools/store_rate.py [genomes] [groups per genome] [--profile]"""
import sys, time, os, tempfile, pickle, cProfile, pstats
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from peppan_amd import mapbsn
from test_mapbsn_golden import _random_groups

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 6600
rng = np.random.default_rng(5)
world = []
if '--real' in sys.argv:
    # the groups of 8 really mapped synthetic genomes (needs the GPU): 10 000 exemplar genes, 2.2 Mb genomes
    from peppan_amd import synth
    names, seqs = synth.make_genes(10000, 0, seed=355)
    os.chdir(tempfile.mkdtemp())
    with open('m.clust.exemplar', 'w') as f:
        for i, q in enumerate(seqs): f.write('>%d\n%s\n' % (i, q.decode()))
    jobs = []
    with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
        for g, (gname, contig, ann) in enumerate(synth.make_genomes(seqs, 8, seed=355)):
            jobs.append((g, 900000 + g, [[100000 + g, contig.decode()]]))
            op.save(100000 + g, np.array([[k, a, b, st, 1] for k, a, b, st in ann[::2]], dtype=object))
    og = np.array([[0, 1, 9000], [4, 5, -2]], dtype=int)
    params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
                  match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
    import io, contextlib
    with contextlib.redirect_stderr(io.StringIO()):
        for job, (tab, ovl) in zip(jobs, mapbsn._gpu_search('m', 'm.clust.exemplar', jobs, params, genomes_per_batch=8)):
            world.append(mapbsn.build_groups(tab, ovl, job[2], mapbsn.OrthoRelation(og), 'm.old_prediction.npz', params))
    print('real genomes: groups', [len(G) for G in world], 'hit rows', [len(G.rows) for G in world])
else:
    for g in range(8):
        G = _random_groups(rng, g, ng, 10000)
        pl = rng.integers(60, 160, size=ng); G.pack_off = np.concatenate([[0], np.cumsum(pl)]).astype(np.int64)
        G.packed = rng.integers(0, 125, size=int(G.pack_off[-1])).astype(np.uint8)
        world.append(G)
# thread-seconds inside the members' making / deflating and inside the appends to the archives
import threading
spent, lk = dict(pack=0., append=0., members=0, bytes_in=0, bytes_out=0), threading.Lock()
_pack, _app = mapbsn._pack_member, mapbsn._append_member
def pack(data, *a):
    t = time.perf_counter(); r = _pack(data, *a); d = time.perf_counter() - t
    with lk: spent['pack'] += d; spent['members'] += 1; spent['bytes_in'] += r[2]; spent['bytes_out'] += len(r[0])
    return r
def app(zf, name, packed):
    t = time.perf_counter(); _app(zf, name, packed); d = time.perf_counter() - t
    with lk: spent['append'] += d
mapbsn._pack_member, mapbsn._append_member = pack, app
blob = [pickle.dumps(mapbsn.StoreBlock(G), protocol=pickle.HIGHEST_PROTOCOL) for G in world]
print('pickled GenomeGroups: %.2f MB, StoreBlock %.2f MB' % (len(pickle.dumps(world[0], protocol=pickle.HIGHEST_PROTOCOL)) / 1e6, len(blob[0]) / 1e6))
os.chdir(tempfile.mkdtemp())
pr = cProfile.Profile() if '--profile' in sys.argv else None
for rep in range(2):
    t0 = time.perf_counter()
    with mapbsn.MapBsn('t.npz', 'w') as c0, mapbsn.MapBsn('s.npz', 'w') as c1, mapbsn.MapBsn('m.npz', 'w') as c2, mapbsn.MapBsn('c.npz', 'w') as c3:
        w = mapbsn._StoreWriter(c0, c1, c2, c3, True)
        if pr and rep: pr.enable()
        t_add = 0.
        for g in range(n):
            G = pickle.loads(blob[g % 8])
            t1 = time.perf_counter()
            w.add(G, 7)
            t_add += time.perf_counter() - t1
        t_fed = time.perf_counter() - t0
        w.write_table(); t_tab = time.perf_counter() - t0 - t_fed
        w.close()
        t_closed = time.perf_counter() - t0
        if pr and rep: pr.disable()
    dt = time.perf_counter() - t0
    print('   thread-seconds: making + deflating members %.2f, appending %.2f; %d members, %.1f MB -> %.1f MB' % (spent['pack'], spent['append'], spent['members'], spent['bytes_in'] / 1e6, spent['bytes_out'] / 1e6)); spent.update(pack=0., append=0., members=0, bytes_in=0, bytes_out=0)
    print('rep %d: %d genomes in %.2f s = %.1f genomes/s   (adds %.2f s, fed after %.2f, gene table %.2f, writer closed %.2f, archives closed %.2f)   sizes MB: %s' % (
        rep, n, dt, n / dt, t_add, t_fed, t_tab, t_closed, dt, ' '.join('%s %.1f' % (f, os.path.getsize(f) / 1e6) for f in ('t.npz', 's.npz', 'm.npz', 'c.npz'))))
if pr:
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
