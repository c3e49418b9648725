"""GPU replacement for the `mmseqs createdb / linclust / createtsv` triple of clust.py:62-66: returns the relation
"representative, member" over the sequences of a FASTA file (every sequence appears once as a member)."""
import numpy as np

from .clust import readFasta
from .uberBlast import get_context

_NT = np.full(256, 4, dtype=np.uint8)
for _c, _v in zip('ACGTacgt', (0, 1, 2, 3, 0, 1, 2, 3)):
    _NT[ord(_c)] = _v
_AA = np.full(256, 20, dtype=np.uint8)
for _i, _c in enumerate('ACDEFGHIKLMNPQRSTVWY'):
    _AA[ord(_c)] = _i


def encode(seq, protein=False):
    return (_AA if protein else _NT)[np.frombuffer(seq.encode('ascii'), dtype=np.uint8)]


def looks_like_protein(seqs):
    sample = ''.join(s[:200] for s in seqs[:50]).upper()
    return len(sample) > 0 and sum(c in 'ACGTN' for c in sample) < 0.9 * len(sample)


def linclust_file(fasta, identity, coverage, device=None):
    recs = readFasta(fasta)
    names = [n for n, _ in recs]
    seqs = [s for _, s in recs]
    protein = looks_like_protein(seqs)
    # all sequences encoded in one table look-up over their concatenation (an array per sequence cost a second per 300 k genes)
    try:
        text = ''.join(seqs).encode('ascii')
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(np.fromiter(map(len, seqs), dtype=np.int64, count=len(seqs)))
        codes = ((_AA if protein else _NT)[np.frombuffer(text, dtype=np.uint8)] if len(text) else np.zeros(1, np.uint8), off)
        n = len(seqs)
    except UnicodeEncodeError:
        codes = [encode(s, protein) for s in seqs]
        n = len(codes)
    base, k = (20, 7) if protein else (4, 17)
    rep, _ = get_context(device).linclust(codes, float(identity), float(coverage), base=base, k=k, m=20) if n else (np.zeros(0, np.uint32), None)
    return [(names[r], names[i]) for i, r in enumerate(rep.tolist())]
