"""GPU replacement for the `mmseqs createdb / linclust / createtsv` triple of clust.py:62-66: returns the relation
"representative, member" over the sequences of a FASTA file (every sequence appears once as a member)."""
import numpy as np

from . import _native as N
from .clust import blocks_of, read_blocks, sequence_of, _read_text
from .uberBlast import get_context

_NT = np.full(256, 4, dtype=np.uint8)
for _c, _v in zip('ACGTacgt', (0, 1, 2, 3, 0, 1, 2, 3)):
    _NT[ord(_c)] = _v
_AA = np.full(256, 20, dtype=np.uint8)
for _i, _c in enumerate('ACDEFGHIKLMNPQRSTVWY'):
    _AA[ord(_c)] = _AA[ord(_c.lower())] = _i              # (sequences are upper-cased before they are encoded: the table does it)


def encode(seq, protein=False):
    return (_AA if protein else _NT)[np.frombuffer(seq.encode('ascii'), dtype=np.uint8)]


def looks_like_protein(seqs):
    sample = ''.join(s[:200] for s in seqs[:50]).upper()
    return len(sample) > 0 and sum(c in 'ACGTN' for c in sample) < 0.9 * len(sample)


def linclust_file(fasta, identity, coverage, device=None):
    text = _read_text(fasta)
    return linclust_text(text, blocks_of(text), identity, coverage, device)


def linclust_text(text, blocks, identity, coverage, device=None):
    """the relation over the records `blocks` of the FASTA text `text` (clust.blocks_of).  The sequences become one code array + offsets in one
    C pass over the text (pep_fasta_scan); a text it does not take (non-ASCII) goes record by record through clust.sequence_of."""
    names = [blk.name for blk in blocks]
    n = len(names)
    protein = looks_like_protein([sequence_of(blk) for blk in blocks[:50]])
    table = _AA if protein else _NT
    codes = None
    try:
        codes = N.fasta_scan(text.encode('ascii'), table, n)
    except UnicodeEncodeError:
        pass
    if codes is None:
        codes = [encode(sequence_of(blk), protein) for blk in blocks]
    elif len(codes[0]) == 0:
        codes = (np.zeros(1, np.uint8), codes[1])
    base, k = (20, 7) if protein else (4, 17)
    rep, _ = get_context(device).linclust(codes, float(identity), float(coverage), base=base, k=k, m=20) if n else (np.zeros(0, np.uint32), None)
    return [(names[r], names[i]) for i, r in enumerate(rep.tolist())]
