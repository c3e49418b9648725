"""seed_match against the size of the filter in front of the index: one search per filter shift (params.reserved[0] = 15 .. 18 -> shift 5 .. 8),
phase timers on.  python3 tools/filter_shift_ab.py [n_genes]"""
import sys
sys.path.insert(0, '.')
from peppan_amd import _native as N, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
names, seqs = synth.make_genes(n, 1002, seed=355)
ctx = N.Context(0)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
ctx.set_timing(2)
ref = None
for sw in (0, 15, 16, 17, 18, 0):
    p = N.default_params(45., 25., 10, 5)
    p.reserved[0] = sw
    ctx.search(p)
    acc = {}
    for rep in range(3):
        h, c, st = ctx.search(p)
        for k in ('ms_seed', 'ms_seed_match', 'ms_total'):
            acc[k] = acc.get(k, 0.) + st[k] / 3
    key = (len(h), int(h['score'].sum()), st['candidates'])
    ref = ref or key
    print('switch %2d (shift %s): seed %.3f ms, seed_match (both shapes) %.3f ms, search %.3f ms; hits %d candidates %d %s' %
          (sw, sw - 10 if sw else 'auto', acc['ms_seed'], acc['ms_seed_match'], acc['ms_total'], len(h), st['candidates'], 'same table' if key == ref else 'DIFFERENT'))
