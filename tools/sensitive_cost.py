"""What the sensitive mode of the translated search costs (four seed shapes instead of two: pep_set_sensitivity level 1, RunBlast(sensitive=True),
uberBlast --sensitive): one all-vs-all search per mode at N genes, phase timers on.  python3 tools/sensitive_cost.py [n_genes ...]"""
import sys
sys.path.insert(0, '.')
from peppan_amd import _native as N, synth
for n in [int(x) for x in sys.argv[1:]] or [10000, 50000]:
    names, seqs = synth.make_genes(n, 1002, seed=355)
    with N.Context(0) as ctx:
        ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
        ctx.set_timing(2)
        for sensitive in (False, True):
            p = N.default_params(45., 25., 10, 5, sensitive=sensitive)
            ctx.search(p)
            acc = {}
            for rep in range(3):
                h, c, st = ctx.search(p)
                for k in ('ms_seed', 'ms_sw', 'ms_sw_trace', 'ms_total'):
                    acc[k] = acc.get(k, 0.) + st[k] / 3
            print('%6d genes, %d seed shapes: search %.2f ms (seed stage %.2f, score pass %.2f, traceback pass %.2f); %d raw seed hits, %d candidates, %d hits'
                  % (n, p.n_shapes, acc['ms_total'], acc['ms_seed'], acc['ms_sw'], acc['ms_sw_trace'], st['seed_hits'], st['candidates'], len(h)))
