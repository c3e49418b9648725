"""get_similar_pairs at 10 000 genes, exemplar file new for every call: the search in front of it and the parts of the decision pass (timing= of pipeline.get_similar_pairs), best of the calls.
usage: python tools/gsp_parts.py [n_genes] [calls]"""
import contextlib, io, os, shutil, sys, tempfile
sys.path.insert(0, '.')
os.environ.setdefault('PEPPAN_LOG', '0')
import numpy as np
from peppan_amd import synth, pipeline as PL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
names, seqs = synth.make_genes(n, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, 'exemplar.fa')
    with open(fa, 'w') as f:
        for i in order:
            f.write('>%s\n%s\n' % (names[i], seqs[i].decode()))
    prio = {int(names[i]): [k, 0, 0] for k, i in enumerate(order)}
    params = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=1, match_frag_prop=0.25, gtable=11, clust_identity=0.9, clust_match_prop=0.8,
                  incompleteCDS='', match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
    ex = os.path.join(tmp, 'p.clust.exemplar')
    np.save(os.path.join(tmp, 'p.clust.npy'), np.zeros((0, 3), dtype=int))
    seen = []
    for _ in range(calls):
        shutil.copy(fa, ex)
        tm = {}
        with contextlib.redirect_stderr(io.StringIO()):
            pairs = PL.get_similar_pairs(ex, prio, dict(params, clust=ex), timing=tm)
        seen.append(tm)
    best = min(seen[2:], key=lambda t: t['search_ms'] + t['decide_ms'])
    print('%d calls; the fastest: search %.2f ms + decision pass %.2f ms = %.2f ms (%d rows, %d pairs); parts of the pass: %s' % (
        calls, best['search_ms'], best['decide_ms'], best['search_ms'] + best['decide_ms'], best['rows'], len(pairs), ' '.join('%s %.2f' % kv for kv in best['decide_parts_ms'].items())))
    med = sorted(t['search_ms'] + t['decide_ms'] for t in seen[2:])[len(seen[2:]) // 2]
    print('median of the calls after the first two: %.2f ms' % med)
