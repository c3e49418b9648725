"""Per-rank search time of the shard shapes an 8-rank run could use, measured on ONE GPU: the (query range x reference range) cell of a
2x4 grid, of a 1x8 grid (reference-only split) and of an 8x1 grid (query-only split) of the bench workload.  No exchange, no merge."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    from peppan_amd import _native as N, synth
    names, seqs = synth.make_genes(10000, 1002, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])
    nts = [seqs[i] for i in order]
    ctx = N.Context(0)
    p = N.default_params(45., 25., 10, 5)
    for label, nq, nr in (('1x1 (whole workload)', 10000, 10000), ('2x4 cell', 5000, 2500), ('1x8 cell', 10000, 1250), ('8x1 cell', 1250, 10000),
                          ('1x2 cell', 10000, 5000), ('2x2 cell', 5000, 5000), ('2x1 cell', 5000, 10000), ('4x1 cell', 2500, 10000)):
        ctx.set_query_nt(nts[:nq], 11)
        ctx.set_ref_nt(nts[:nr], 6, 11)
        ctx.translate()
        for _ in range(3):
            ctx.search(p, copy=False)
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.invalidate_translation()                 # K1 again, inside the search
            h, c, st = ctx.search(p, copy=False)
        dt = (time.perf_counter() - t0) / 10
        print('%-22s %5d queries x %5d reference genes: %.3f ms per search (K1 included), %d hits' % (label, nq, nr, dt * 1e3, len(h)), flush=True)


if __name__ == '__main__':
    main()
