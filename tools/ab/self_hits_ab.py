"""A/B on one box: the self-search with the diagonal-0 self hits dropped in the matcher (default) against the plain stream (params.reserved[0] = 10).
    python tools/ab/self_hits_ab.py [n_genes] [steps]
Step wall time (K1 inside, as bench.py's step) and the seed phases (HIP events, pep_set_timing 2), alternating; tables compared byte for byte."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
names, seqs = synth.make_genes(n, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
seqs = [seqs[i] for i in order]
flags = tuple(int(x) for x in sys.argv[3].split(',')) if len(sys.argv) > 3 else (0, 10)        # e.g. 0,11: the filter word asked for self positions too (11)
blastn = len(sys.argv) > 4 and sys.argv[4] == 'blastn'                                        # the nucleotide tool's search instead of the translated one
LABEL = {0: 'default', 10: 'plain stream', 11: 'filter asked too', 12: 'two start[] loads', 8: 'plain matcher'}
ctx = N.Context(0)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
keep = {}


def again():
    if blastn:
        ctx.use_nt_as_residues(2)
    else:
        ctx.invalidate_translation()


for rep in range(2):
    for flag in flags:
        p = N.nucleotide_params(45., 25.) if blastn else N.default_params(45., 25., 10, 5)
        p.reserved[0] = flag
        for _ in range(5):
            again(); ctx.search(p, copy=False)
        ctx.set_timing(0)
        t0 = time.perf_counter()
        for _ in range(steps):
            again(); h, c, st = ctx.search(p, copy=False)
        wall = (time.perf_counter() - t0) / steps * 1e3
        ctx.set_timing(2)
        acc = {}
        for _ in range(20):
            again(); h, c, st = ctx.search(p, copy=False)
            for k in ('ms_seed', 'ms_seed_match', 'ms_sw', 'ms_sw_trace', 'ms_total'):
                acc[k] = acc.get(k, 0.) + st[k] / 20
        ctx.set_timing(0)
        h, c = np.array(h), np.array(c)
        print('%-18s step %.3f ms   seed stage %.3f  (matchers %.3f)  sw %.3f  trace %.3f  gpu total %.3f   raw hits %d candidates %d hits %d' % (
            LABEL.get(flag, str(flag)), wall, acc['ms_seed'], acc['ms_seed_match'], acc['ms_sw'], acc['ms_sw_trace'], acc['ms_total'], st['seed_hits'], st['candidates'], len(h)), flush=True)
        key = (h.tobytes(), c.tobytes(), st['seed_hits'], st['target_seeds'], st['candidates'])
        keep.setdefault(flag, key)
        assert keep[flag] == key
print('identical tables / raw hit counts / candidates:', all(keep[f] == keep[flags[0]] for f in flags))
