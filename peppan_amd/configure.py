"""Host-side utilities with the reference's names and semantics (modules/configure.py of PEPPAN) but none of its
external-binary discovery: nothing here shells out, the arithmetic lives in libpeppan_hip.so.

    logger                 configure.py:197-199
    uopen                  configure.py:90-115   (gzip handled in-process instead of a pigz/gzip pipe)
    readFasta / readFastq  configure.py:118-150
    rc                     configure.py:152-154
    transeq                configure.py:160-194  (host mirror; the search path translates on the GPU, K1)
    blosum62               configure.py:49-87    (same 32-stride letter indexing, built from the standard matrix)
"""
import gzip
import os
import io
import re
import sys
from datetime import datetime

import numpy as np

xrange = range
asc2int = np.uint32


def logger(log, pipe=None):
    """the reference's progress lines (configure.py:23-25: time stamp, tab, text, to stderr).  PEPPAN_LOG=0 in the environment silences them
    (bench.py does: a run of the mapping legs writes 1 500 of them, and worker processes inherit the setting); the stream is looked up per
    call, so contextlib.redirect_stderr works on it too"""
    if os.environ.get('PEPPAN_LOG', '1') == '0':
        return
    pipe = pipe if pipe is not None else sys.stderr
    pipe.write('{0}\t{1}\n'.format(str(datetime.now()), log))
    pipe.flush()


def effective_cpus():
    """CPUs this process may actually use: the affinity mask, cut down to the control group's CPU allowance when there is one (a container
    that shows 256 hardware threads and grants 16 CPUs' worth of time runs 256 threads no faster than 16 - and polling threads steal from
    working ones).  Not part of the reference: its pools are sized by --n_thread."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                         # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:      # cgroup v1
                quota, period = int(f.read()), int(g.read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return n


class uopen(object):
    """context manager over a plain or gzipped text file ('r' or 'w')"""

    def __init__(self, fname, label='r'):
        gz = fname.lower().endswith('gz')
        if 'r' in label:
            self.fstream = io.TextIOWrapper(gzip.open(fname, 'rb'), encoding='utf-8') if gz else open(fname)
        else:
            self.fstream = io.TextIOWrapper(gzip.open(fname, 'wb'), encoding='utf-8')

    def __enter__(self):
        return self.fstream

    def __exit__(self, *exc):
        self.fstream.close()

    def __iter__(self):
        return iter(self.fstream)

    def write(self, doc):
        self.fstream.write(doc)

    def close(self):
        self.fstream.close()


def _record_text(body):
    """the text behind a record's header -> its sequence: comment lines (leading '#') left out, all white space dropped, upper case"""
    if '#' in body:
        body = ''.join(ln for ln in body.split('\n') if not ln.startswith('#'))      # (lines end at '\n' only, as for the reference's line iterator)
    return ''.join(body.split()).upper()


def _read_bytes(fname):
    with (gzip.open(fname, 'rb') if fname.lower().endswith('gz') else open(fname, 'rb')) as fin:
        return fin.read()


def _fasta_text_records(text, headOnly=False):
    """FASTA text -> {name: sequence}, record by record (any text: Unicode names and blanks, headers without a name raise as the reference's do)"""
    records = {}
    for block in ('\n' + text).split('\n>')[1:]:                           # (what stands in front of the first header belongs to no record)
        header, _, body = block.partition('\n')
        records[header.split()[0]] = '' if headOnly else _record_text(body)
    return records


def _fasta_records(data, headOnly=False):
    """FASTA file content (bytes) -> {name: sequence}.  Plain ASCII without carriage returns - every file PEPPAN writes for its own searches - is cut by
    one pass of the library's host code (pep_fasta_records: names, and all sequences cleaned into one text that is sliced here); anything else is
    decoded as the text-mode reader would and goes record by record."""
    if not headOnly and data.isascii() and b'\r' not in data:
        from . import _native
        got = _native.fasta_records(data, as_dict=True)
        if got is not None:
            return got
    return _fasta_text_records(io.TextIOWrapper(io.BytesIO(data), encoding='utf-8').read(), headOnly)


def readFasta(fasta, headOnly=False):
    """FASTA file -> {name: sequence}: the name is the first word of a header line, the sequence everything up to the next header
    without its white space, upper-cased; lines that start with '#' are comments; of two records with one name the later counts
    (configure.py:118-128)."""
    return _fasta_records(_read_bytes(fasta), headOnly)


_PSEUDO_QUALITY = bytes(ord('I') if chr(c) in 'ACGTacgt' else ord('!') for c in range(256))


def readFastq(fastq, with_qual=True):
    """FASTQ file -> ({name: sequence}, {name: quality string}), four lines per read, the name being the first word behind the '@'.
    A file that does not start with '@' is taken as FASTA and every sequence gets a made-up quality: 'I' under A, C, G, T and '!' under
    anything else (configure.py:129-147).  with_qual=False leaves the made-up qualities out (nothing on the search path reads them, and
    for a genome they cost more than its search)."""
    data = _read_bytes(fastq)
    if not data.startswith(b'@'):
        seq = _fasta_records(data)
        if not with_qual:
            return seq, None
        return seq, {name: s.encode('latin-1', 'replace').translate(_PSEUDO_QUALITY).decode('ascii') for name, s in seq.items()}
    text = io.TextIOWrapper(io.BytesIO(data), encoding='utf-8').read()
    lines = text.splitlines()
    lines += [''] * (-len(lines) % 4)
    seq, qual = {}, {}
    for header, bases, _, scores in zip(lines[0::4], lines[1::4], lines[2::4], lines[3::4]):
        name = header[1:].split()[0]
        seq[name], qual[name] = _record_text(bases), ''.join(scores.split())
    return seq, qual


_COMP = str.maketrans('ACGTN', 'TGCAN')


def rc(seq, missingValue='N'):
    s = seq.upper()
    if missingValue == 'N' and not re.search(r'[^ACGTN]', s):
        return s.translate(_COMP)[::-1]
    comp = {'A': 'T', 'T': 'A', 'G': 'C', 'C': 'G', 'N': 'N'}
    return ''.join(comp.get(c, missingValue) for c in reversed(s))


# ---------------------------------------------------------------------------------------------- translation
_AA64 = 'KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVXYXYSSSSXCWCLFLF'


def _codon_table(transl_table, markStarts):
    tab = np.array(list(_AA64 + '-'))
    if transl_table == 4:
        tab[56] = 'W'
    if markStarts:
        tab[[46, 62]] = 'M'
    return tab


_NT = np.full(256, -1, dtype=np.int64)
_NT[[ord(c) for c in 'ACGT']] = (0, 1, 2, 3)
_NT[ord('-')] = -2


def _translate_codes(codes, tab):
    """codes: int array (0..3, -1 ambiguous, -2 gap) of any length -> protein string"""
    n = codes.size
    if n == 0:
        return ''
    pad = (-n) % 3
    if pad:
        codes = np.concatenate([codes, np.full(pad, -1, dtype=np.int64)])
    c = codes.reshape(-1, 3)
    idx = (c[:, 0] << 4) | (c[:, 1] << 2) | c[:, 2]
    bad = (c < 0).any(1)
    gap = (c == -2).any(1)
    idx = np.where(bad, 0, idx)
    aa = tab[idx]
    aa[bad] = 'X'
    aa[gap] = '-'
    return ''.join(aa.tolist())


_FRAME_SETS = {'F': (1, 2, 3), 'R': (4, 5, 6), '7': (1, 2, 3, 4, 5, 6)}


def _frame_list(frame):
    """'F' | 'R' | '7' (any case; 7 as a number too) or a comma list of frame numbers 1..6"""
    key = str(frame).upper()
    return _FRAME_SETS[key] if key in _FRAME_SETS else tuple(int(f) for f in key.split(','))


def _proteins(nt, frames, tab):
    """the translations of one nucleotide string in the asked frames (4..6: the reverse strand, complemented)"""
    fw = _NT[np.frombuffer(nt.upper().encode('ascii'), dtype=np.uint8)]
    rv = np.where(fw >= 0, 3 - fw, fw)[::-1] if max(frames) > 3 else None
    return [_translate_codes(fw[f - 1:] if f <= 3 else rv[f - 4:], tab) for f in frames]


def transeq(seq, frame=7, transl_table=None, markStarts=False):
    """dict name->nt (or list of [name, nt]) -> dict name->[protein per frame] (or list of [name, [..]]).
    frame: 'F' (1,2,3), 'R' (4,5,6), '7' (all six) or a comma list; stops and ambiguous codons are 'X',
    codons containing '-' are '-', a trailing partial codon is 'X'; transl_table 4 reads TGA as W (configure.py:160-194)."""
    frames, tab = _frame_list(frame), _codon_table(transl_table, markStarts)
    if isinstance(seq, dict):
        return {name: _proteins(nt, frames, tab) for name, nt in seq.items()}
    return [[name, _proteins(nt, frames, tab)] for name, nt in seq]


# ---------------------------------------------------------------------------------------------- BLOSUM62
def _blosum62():
    order = 'ARNDCQEGHILKMFPSTWYVBZX*'
    rows = """ 4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0 -2 -1  0 -4
-1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3 -1  0 -1 -4
-2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3  3  0 -1 -4
-2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3  4  1 -1 -4
 0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1 -3 -3 -2 -4
-1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2  0  3 -1 -4
-1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4
 0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3 -1 -2 -1 -4
-2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3  0  0 -1 -4
-1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3 -3 -3 -1 -4
-1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1 -4 -3 -1 -4
-1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2  0  1 -1 -4
-1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1 -3 -1 -1 -4
-2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1 -3 -3 -1 -4
-1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2 -2 -1 -2 -4
 1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2  0  0  0 -4
 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0 -1 -1  0 -4
-3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3 -4 -3 -2 -4
-2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1 -3 -2 -1 -4
 0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4 -3 -2 -1 -4
-2 -1  3  4 -3  0  1 -1  0 -3 -4  0 -3 -3 -2  0 -1 -4 -3 -3  4  1 -1 -4
-1  0  0  1 -3  3  4 -2  0 -3 -3  1 -1 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4
 0 -1 -1 -1 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -2  0  0 -2 -1 -1 -1 -1 -1 -4
-4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4  1"""
    m = np.array([[int(v) for v in r.split()] for r in rows.split('\n')])
    # the reference stores '*' under the letter U ("* is designated as U", configure.py:48)
    letters = [c if c != '*' else 'U' for c in order]
    tab = np.zeros(858, dtype=float)
    for i, a in enumerate(letters):
        for j, b in enumerate(letters):
            tab[(ord(a) - 65) * 32 + ord(b) - 65] = m[i, j]
    return tab


blosum62 = _blosum62()
