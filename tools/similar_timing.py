"""Timing of the consumer of the all-vs-all table on the bench workload: pipeline.get_similar_pairs (search as numeric table -> host scan
-> K14 on the GPU -> resolve) and the pieces inside it.  python3 tools/similar_timing.py [n_genes]"""
import contextlib
import cProfile
import io
import os
import pstats
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    import numpy as np
    from peppan_amd import synth, pipeline as PL
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    names, seqs = synth.make_genes(n, 1002, seed=355)
    prio = {i: [0, -len(s), i] for i, s in enumerate(seqs)}
    params = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=2, match_frag_prop=0.25, gtable=11, clust_identity=0.9, clust_match_prop=0.8,
                  incompleteCDS='', match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, 'src.fa')
        with open(src, 'w') as f:
            for i, s in enumerate(seqs):
                f.write('>%d\n%s\n' % (i, s.decode()))
        np.save(os.path.join(tmp, 'ex.clust.npy'), np.zeros((0, 3), dtype=int))
        ex = os.path.join(tmp, 'ex.clust.exemplar')
        for rep in range(4):
            shutil.copy(src, ex)
            tm = {}
            pr = cProfile.Profile() if rep == 3 else None
            with contextlib.redirect_stderr(io.StringIO()):
                t = time.perf_counter()
                if pr:
                    pr.enable()
                pairs = PL.get_similar_pairs(ex, prio, dict(params, clust=ex), timing=tm)
                if pr:
                    pr.disable()
                dt = time.perf_counter() - t
            print('get_similar_pairs: %.1f ms total (search %.1f ms, decision pass %.1f ms), %d rows -> %d pairs' % (dt * 1e3, tm['search_ms'], tm['decide_ms'], tm['rows'], len(pairs)))
            print('   decision pass by part (ms):', {k: round(v, 2) for k, v in tm.get('decide_parts_ms', {}).items()})
    for key in ('cumulative', 'tottime'):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30)
        print(s.getvalue())
    # the exemplar rewrite by step
    import numpy as np
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, 'src.fa')
        with open(src, 'w') as f:
            for i, sq in enumerate(seqs):
                f.write('>%d\n%s\n' % (i, sq.decode()))
        for rep in range(3):
            t = [time.perf_counter()]
            data = open(src, 'rb').read(); t.append(time.perf_counter())
            find = data.find
            starts = [0]; p = find(b'\n>')
            while p >= 0:
                starts.append(p + 1); p = find(b'\n>', p + 2)
            t.append(time.perf_counter())
            heads = [data[x + 1:x + 65] for x in starts]
            names = [h.split(None, 1)[0] for h in heads]
            ids = np.array(names, dtype='S64').astype(np.int64); t.append(time.perf_counter())
            with open(os.path.join(tmp, 'out.fa'), 'wb') as fo:
                fo.write(memoryview(data)[:len(data) * 6 // 7])
            t.append(time.perf_counter())
            print('rewrite steps (ms): read %.2f, find %.2f, names %.2f, write %.2f' % tuple((b - a) * 1e3 for a, b in zip(t, t[1:])))


if __name__ == '__main__':
    main()
