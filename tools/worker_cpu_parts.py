"""What ONE mapping worker spends its CPU on, part by part, single-threaded so that time.thread_time() sees everything: rounds of 16 genomes (10 000 exemplars, 2.2 Mb genomes)
through the worker's own sequence - _gpu_search, build_groups_round, StoreBlock, round_members with the packers run inline - and inside round_members the member emitters,
the deflaters and the CRC by store, with their bytes.  A .mat and a .seq member are left under gpurun_out/ (r6_member_mat.bin, r6_member_seq.bin) for work on the coders.
usage: python tools/worker_cpu_parts.py [rounds] [genomes per round]"""
import sys, os, time, zlib, tempfile, collections
sys.path.insert(0, '.')
os.environ.setdefault('PEPPAN_LOG', '0')
import numpy as np
from peppan_amd import mapbsn, synth, uberBlast as UB, _native as N

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
per = int(sys.argv[2]) if len(sys.argv) > 2 else 16
names, seqs = synth.make_genes(10000, 0, seed=355)
params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
              match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
os.chdir(tempfile.mkdtemp(prefix='pep_parts_'))
with open('m.clust.exemplar', 'w') as f:
    for i, s in enumerate(seqs):
        f.write('>%d\n%s\n' % (i, s.decode()))
worlds = synth.make_genomes(seqs, rounds * per, seed=355)
jobs = []
with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
    for g, (gname, contig, ann) in enumerate(worlds):
        jobs.append((g, 900000 + g, [[100000 + g, contig.decode()]]))
        op.save(100000 + g, np.array([[k, a, b, st, 1] for k, a, b, st in ann[::2]], dtype=object))
ortho = mapbsn.OrthoRelation(np.array([[0, 1, 9000], [4, 5, -2]], dtype=int))
old = mapbsn.MapBsn('m.old_prediction.npz')
ctx2 = N.Context(0)

spent, nbytes = collections.Counter(), collections.Counter()
cpu = time.thread_time


class Inline(object):
    def submit(self, f, *a):
        class R(object):
            pass
        r = R()
        v = f(*a)
        r.result = lambda: v
        return r


mapbsn._PACKERS = Inline()
plain_pack = mapbsn._pack_member
kept = {}


def pack(data, strategy=zlib.Z_DEFAULT_STRATEGY):
    which = 'mat' if strategy == mapbsn.MAT_STRATEGY else 'seq'
    t0 = cpu()
    raw = data() if callable(data) else data
    t1 = cpu()
    out = plain_pack(raw, strategy)
    t2 = cpu()
    zlib.crc32(raw)
    t3 = cpu()
    spent[which + '_emit'] += t1 - t0
    spent[which + '_deflate+crc'] += t2 - t1
    spent[which + '_crc_alone'] += t3 - t2
    nbytes[which + '_raw'] += len(raw)
    nbytes[which + '_packed'] += len(out[0])
    nbytes[which + '_members'] += 1
    kept.setdefault(which, raw)
    return out


mapbsn._pack_member = pack
first = 0
for r in range(rounds + 1):                     # (round 0 warms up: contexts, the exemplar set's index, numpy's first calls)
    mine = jobs[max(0, r - 1) * per:(max(0, r - 1) + 1) * per]
    if r == 1:
        spent.clear(); nbytes.clear()
        wall0 = time.perf_counter()
    t0 = cpu()
    found = list(mapbsn._gpu_search('m', 'm.clust.exemplar', mine, params, genomes_per_batch=per))
    t1 = cpu()
    groups = mapbsn.build_groups_round([(tab, ovl, job[2]) for job, (tab, ovl) in zip(mine, found)], ortho, old, params, ctx2)
    t2 = cpu()
    blocks = [mapbsn.StoreBlock(G) for G in groups]
    t3 = cpu()
    P = mapbsn.round_members(blocks, [job[1] for job in mine], first, True)
    t4 = cpu()
    first += P['n']
    spent['search (this thread only: the tools run on threads of their own)'] += t1 - t0
    spent['build_groups_round'] += t2 - t1
    spent['StoreBlock'] += t3 - t2
    spent['round_members'] += t4 - t3
n = rounds * per
tm = os.times()
print('%d rounds of %d genomes, one thread, %.2f s of wall clock; ms of CPU per genome:' % (rounds, per, time.perf_counter() - wall0))
for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
    print('  %-70s %7.2f' % (k, v / n * 1e3))
print('bytes per genome:', {k: int(v / n) for k, v in nbytes.items()})
print('process user %.2f system %.2f s since start (synthetic genomes and warm-up included)' % (tm.user, tm.system))
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
for which, raw in kept.items():
    open(os.path.join(root, 'gpurun_out', 'r6_member_%s.bin' % which), 'wb').write(raw)
