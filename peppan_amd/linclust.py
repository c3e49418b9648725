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
    protein = looks_like_protein([s for _, s in recs])
    codes = [encode(s, protein) for _, s in recs]
    base, k = (20, 7) if protein else (4, 17)
    rep, _ = get_context(device).linclust(codes, float(identity), float(coverage), base=base, k=k, m=20)
    return [(names[r], names[i]) for i, r in enumerate(rep.tolist())]
