"""Rule 5a and the comparison of identical pairs on real genes: how many of the traced pairs of an all-vs-all search are settled as one ungapped run (no traceback sweep).
Fixtures: golden G16 (1 644 genes of the reference's examples/) and G17 (the 8 441 unique genes of its four example genomes).
python3 tools/gapless_real.py"""
import gzip
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)


def genes_of(name):
    seqs = []
    with gzip.open(os.path.join(ROOT, 'tests', 'golden', name), 'rt') as f:
        for rec in f.read().split('>')[1:]:
            seqs.append(''.join(rec.split('\n')[1:]).strip().encode())
    return seqs


def main():
    from peppan_amd import _native as N
    ctx = N.Context(0)
    ctx.set_timing(2)
    for label, fn in (('G16', 'g16_real_genes.fa.gz'), ('G17', 'g17_examples_genes.fa.gz')):
        seqs = genes_of(fn)
        ctx.set_query_nt(seqs, 11)
        ctx.set_ref_nt(seqs, 6, 11)
        for flag, what in ((0, 'production'), (2, 'every pair swept'), (4, 'identical pairs compared whatever the size')):
            p = N.default_params(45., 25., 10, 5)
            p.reserved2 = flag
            for _ in range(3):
                h, c, st = ctx.search(p)
            print('%s (%s): %d genes, %d candidates, %d of them identical pairs scored by comparison, %d traced pairs, %d of them one ungapped run (rule 5a), %d hits; '
                  'score pass %.3f ms, traceback pass %.3f ms, search %.3f ms'
                  % (label, what, len(seqs), st['candidates'], st['candidates_settled'], st['tracebacks'], st['tracebacks_gapless'], len(h), st['ms_sw'], st['ms_sw_trace'], st['ms_total']))


if __name__ == '__main__':
    main()
