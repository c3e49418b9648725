"""Drop-in for PEPPAN's modules/uberBlast.py with the external aligners replaced by libpeppan_hip.so.

Same entry points and table format:
    uberBlast(args, extPool=None)                       uberBlast.py:564-613   same argparse flags (+ --gpu, --device)
    RunBlast().run(ref, qry, methods, min_id, ...)      uberBlast.py:326-376   same positional / keyword arguments
    tools: 'diamond', 'diamondself', 'gpu'  -> translated search on the MI355X (K1..K8), replaces runDiamond
                                              (uberBlast.py:513-560) and parseDiamond (uberBlast.py:16-70)
           'blastn'                          -> nucleotide search on the same engine (runBlast), replaces
                                              runBlast / poolBlast / parseBlast / getCIGAR (uberBlast.py:274-320, 482-509)
Each tool returns the reference's 15-column object rows (names str, CIGAR as [[n, op], ...] in nucleotides);
run() then applies reScore (K7 on the GPU for mode 1), the -f/-m filters, fixEnd, -O overlaps and the final
string-keyed sort exactly as the reference does.  There is no CPU fallback: without the HIP library or a GPU
the tools raise.
"""
import os
import sys
import numpy as np

from . import _native as N
from . import mapfilters
from .hittable import HitTable
from .configure import logger, readFastq, blosum62, asc2int

_OPS = np.array(['M', 'I', 'D'])
_OP_CODE = {'M': 0, 'I': 1, 'D': 2}

# ---- tables of the rescoring modes (uberBlast.py:270-272)
nucEncoder = np.repeat(2, 255).astype(int)
nucEncoder[[ord(c) for c in 'ACGT']] = (0, 1, 3, 4)
# codon (base-5 digits of the A0 C1 N2 G3 T4 code) -> amino-acid letter index; built from the standard table
def _make_gtable():
    aa64 = 'KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVXYXYSSSSXCWCLFLF'   # A0 C1 G2 T3 order
    five = {0: 0, 1: 1, 2: 3, 3: 4}
    t = np.full(125, ord('X') - 65, dtype=int)
    for i, a in enumerate(aa64):
        d = [five[(i >> 4) & 3], five[(i >> 2) & 3], five[i & 3]]
        t[d[0] * 25 + d[1] * 5 + d[2]] = ord(a) - 65
    return t


gtable = _make_gtable()

_CTX = {}
_FASTA_CACHE = {}


def _read_cached(path, with_key=False):
    """sequences of a FASTA/FASTQ file, re-parsed only when the file changed: PEPPAN maps the SAME gene file against
    every genome (PEPPAN.py:768-772), one uberBlast call per genome.  with_key: also the cache key (path, mtime, size) that identifies
    this content - the prepared form of a side (_prepare_side) and what a context holds on the device are remembered under it."""
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    hit = _FASTA_CACHE.get(key)
    if hit is None:
        while len(_FASTA_CACHE) >= 4:             # the oldest file goes (one at a time: letting go of four files' 40 000 strings at once was 3 ms of the call that met the limit)
            old = next(iter(_FASTA_CACHE))
            del _FASTA_CACHE[old]
            _SIDE_CACHE.pop(old, None)
        hit = _FASTA_CACHE[key] = readFastq(path, with_qual=False)[0]
    seqs = dict(hit)         # callers may mutate their copy (the reference's reScore does, uberBlast.py:402-405)
    return (seqs, key) if with_key else seqs


def _reference_seqs(ref):
    """the sequences of one reference of a batch: a FASTA / FASTQ path, or the sequences themselves - a dict or a list of (name, sequence)
    pairs, read as a FASTA file of those records would be (names as text, first token; upper case; no white space) without the file"""
    if isinstance(ref, (str, bytes, os.PathLike)):
        return _read_cached(ref)
    out = {}
    for n, s in (ref.items() if isinstance(ref, dict) else ref):
        raw = isinstance(s, (bytes, bytearray))                                     # (ASCII bytes are taken as they are: the mapping workers pass them)
        codes = np.frombuffer(s if raw else s.encode('ascii', 'replace'), dtype=np.uint8)
        if len(codes) and (codes.min() <= 32 or codes.max() >= 97):               # white space or lower case somewhere: as readFasta would read it
            s = b''.join(bytes(s).split()).upper() if raw else ''.join(s.split()).upper()
        out[str(n).split()[0]] = s
    return out


_SIDE_CACHE = {}
_Q_CODES = {}            # the base codes of the last query file the nucleotide tool's window path encoded (one entry)


def _prepare_side(seqs, key, names=None):
    """one side of a search ready for the device: names in the reference's FASTA order (sorted, uberBlast.py:527, 537; or as given), the
    shared str name table, nucleotide lengths, and the packed (bytes, offsets) pair Context.set_*_nt takes.  Remembered per file."""
    side = _SIDE_CACHE.get(key) if (key is not None and names is None) else None
    if side is None:
        order = sorted(seqs) if names is None else list(names)
        texts = [RunBlast._text(seqs[n]) for n in order]
        side = dict(names=order, index={n: i for i, n in enumerate(order)}, tab=_NameTable(str(x) for x in order),
                    sorted=names is None and all(isinstance(x, str) for x in order),
                    lens=np.fromiter(map(len, texts), dtype=np.int64, count=len(texts)), packed=N._pack(texts))
        if key is not None and names is None:
            _SIDE_CACHE[key] = side
    return side


class _gpu_gate(object):
    """Worker processes of one pool share one GPU (mapworkers.py).  Their batched searches are chains of short launches with host round trips in
    between; more than a few of them in flight at once and every one of them crawls (eight at once: 2 - 3 times a search's time alone).  The pool
    therefore hands out PEPPAN_GPU_GATE="<directory>,<M>": a batch's searches run while the process holds one of M lock files - the other workers
    do their host work (groups, members) meanwhile.  Nothing set: no gate."""

    def __enter__(self):
        self.f = None
        spec = os.environ.get('PEPPAN_GPU_GATE')
        if spec:
            import fcntl
            where, m = spec.rsplit(',', 1)
            m = max(1, int(m))
            files = [open(os.path.join(where, 'gate%d' % i), 'a') for i in range(m)]
            first = os.getpid() % m
            for k in range(m):
                f = files[(first + k) % m]
                try:
                    fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
                    self.f = f
                    break
                except OSError:
                    pass
            if self.f is None:
                self.f = files[first]
                fcntl.flock(self.f, fcntl.LOCK_EX)
            for f in files:
                if f is not self.f:
                    f.close()
        return self

    def __exit__(self, *exc):
        if self.f is not None:
            self.f.close()                      # (closing gives the lock back)
        return False


def get_context(device=None):
    """one HIP context per (process, device), created lazily so that forked workers make their own
    (the reference forks pool workers before calling uberBlast, PEPPAN.py:922)"""
    if device is None:
        device = int(os.environ.get('PEPPAN_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    key = (os.getpid(), device)
    if key not in _CTX:
        _CTX[key] = N.Context(device)
    return _CTX[key]


_plain_get_context = get_context


def get_nucl_context(device=None):
    """The context the nucleotide tool works in when a run asks for both tools (PEPPAN's calls do: --blastn --diamond, uberBlast.py:597-599): a second
    HIP context - stream, work space, packed sets of its own - on the same device, so that the two searches of one run go side by side (one tool's
    memory-bound seed stage under the other's Smith-Waterman passes, one tool's table being built while the other searches) and neither packs its
    sets over the other's.  The SAME context as get_context when that function has been replaced (tests) or PEPPAN_ONE_CONTEXT=1."""
    if get_context is not _plain_get_context or os.environ.get('PEPPAN_ONE_CONTEXT') == '1':
        return get_context(device)
    if device is None:
        device = int(os.environ.get('PEPPAN_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    key = (os.getpid(), device, 'nucl')
    if key not in _CTX:
        _CTX[key] = N.Context(device)
    return _CTX[key]


def get_side_context(role, device=None):
    """one more HIP context of this process on the device, kept under `role` like the search contexts: work that runs on a thread of its own beside the
    searches (K12 inside build_groups while the next batch of genomes is being searched, mapbsn.get_map_bsn) must not share their stream and work space"""
    if device is None:
        device = get_context().device
    key = (os.getpid(), device, role)
    if key not in _CTX:
        _CTX[key] = N.Context(device)
    return _CTX[key]


# ------------------------------------------------------------------------------------------------------------
# GPU hits -> the reference's table rows (coordinate algebra of parseDiamond, uberBlast.py:25-58)
# ------------------------------------------------------------------------------------------------------------
class _NameTable(list):
    """a name table this module built: every entry is a str (tables that share one concatenate without a remap)"""
    __slots__ = ()


def _str_table(names):
    """a name table as a list of str.  Only a table built here (_prepare_side) is passed on as it is; anything a caller hands in is coerced
    entry by entry - PEPPAN's encoded gene ids are integers, and a mixed list must not reach the row builder"""
    return names if isinstance(names, _NameTable) else _NameTable(str(x) for x in names)


def hits_to_table(hits, cigar, q_meta, t_meta, q_names, r_names, q_len, r_len, min_id, min_cov, min_ratio, nt_match=None):
    """hits/cigar: output of Context.search for a translated search.  q_len / r_len: nucleotide lengths per sequence index.
    Returns the numeric HitTable of the rows parseDiamond would keep (coordinate algebra and filters of uberBlast.py:25-58)."""
    if len(hits) == 0:
        return HitTable.empty()
    c, arena = N.table_from_hits(0, hits, cigar, q_len, r_len, min_id, min_cov, min_ratio, q_meta=q_meta, t_meta=t_meta, nt_match=nt_match)      # host C++: one pass over the records
    T = HitTable(_str_table(q_names), _str_table(r_names), c['qi'], c['ri'], c['iden'], c['aln'], c['mis'], c['gap'], c['qs'], c['qe'], c['ss'], c['se'],
                 c['evalue'], c['score'], c['ql'], c['sl'], arena, c['c_off'], c['c_runs'], rid=c['rid'], score_is_int=nt_match is None)
    T.rescored = nt_match is not None        # (nt_match: the search counted K7's identical columns - the rows carry reScore mode 1's identity and score)
    return T


def hits_to_blastab(hits, cigar, q_meta, t_meta, q_names, r_names, q_len, r_len, min_id, min_cov, min_ratio):
    """the same as the reference's row format: ndarray(object)[n, 15], CIGAR as [[n, op], ...] in nucleotides"""
    return hits_to_table(hits, cigar, q_meta, t_meta, q_names, r_names, q_len, r_len, min_id, min_cov, min_ratio).to_rows(with_rid=False)


def _three_decimals(v):
    """float('%.3f' % x) for every element (blastn prints pident with three decimals, uberBlast.py:282): the decimal string is the
    correctly rounded 3-digit decimal of the double, and float() of it is the double nearest to that decimal - np.round(v, 3) gives the
    same value except when x * 1000 sits within an ulp of a tie, so those few go through the string"""
    v = np.asarray(v, dtype=np.float64)
    scaled = v * 1000.
    out = np.round(v, 3)
    frac = np.abs(scaled - np.floor(scaled) - 0.5)
    near = np.flatnonzero(frac < 1e-6)
    if len(near):
        out[near] = [float('%.3f' % x) for x in v[near].tolist()]
    return out


_NT_CODE = np.full(256, 4, dtype=np.uint8)
for _c, _v in zip('ACGTacgt', (0, 1, 2, 3, 0, 1, 2, 3)):
    _NT_CODE[ord(_c)] = _v
_NT_RC = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
_TILE_HOME = 1 << 22          # home stretch of one window when a reference strand exceeds PEP_MAX_SEQ_LEN (runBlast)


def _encode_nt(texts):
    """list of nucleotide strings -> (codes A0 C1 G2 T3 other 4 of the concatenation, uint64 offsets[n+1]); one table pass for all of them"""
    off = np.zeros(len(texts) + 1, dtype=np.uint64)
    if texts:
        off[1:] = np.cumsum(np.fromiter(map(len, texts), dtype=np.int64, count=len(texts)))
    raw = b''.join(t if isinstance(t, (bytes, bytearray)) else t.encode('ascii') for t in texts)      # (str or ASCII bytes: the mapping workers pass bytes)
    return _NT_CODE[np.frombuffer(raw, dtype=np.uint8)], off


def blast_hits_to_table(hits, cigar, q_names, r_names, q_len, r_len, min_id, min_cov, min_ratio, params, t_seq, t_rev, windows=None, nt_match=None):
    """nucleotide-search hits -> the rows parseBlast builds from blastn's outfmt 6 (uberBlast.py:275-290, 311-320), as a HitTable:
    t_seq / t_rev give the reference sequence and strand of every target (reverse strand: sstart > send); identity
    carries blastn's 3 printed decimals"""
    if len(hits) == 0:
        return HitTable.empty()
    q_len = np.asarray(q_len, dtype=np.int64)
    evalue = params.ka_k * q_len[hits['q'].astype(np.int64)] * params.dbsize * np.exp(-params.ka_lambda * hits['score'].astype(np.int64))
    c, arena = N.table_from_hits(1, hits, cigar, q_len, r_len, min_id, min_cov, min_ratio, t_seq=t_seq, t_rev=t_rev, windows=windows, evalue=evalue, nt_match=nt_match)
    T = HitTable(_str_table(q_names), _str_table(r_names), c['qi'], c['ri'], c['iden'], c['aln'], c['mis'], c['gap'], c['qs'], c['qe'], c['ss'], c['se'],
                 c['evalue'], c['score'], c['ql'], c['sl'], arena, c['c_off'], c['c_runs'], rid=c['rid'], score_is_int=nt_match is None)
    T.rescored = nt_match is not None
    return T


def blast_hits_to_blastab(hits, cigar, q_names, r_names, q_len, r_len, min_id, min_cov, min_ratio, params, t_seq, t_rev, windows=None):
    return blast_hits_to_table(hits, cigar, q_names, r_names, q_len, r_len, min_id, min_cov, min_ratio, params, t_seq, t_rev, windows).to_rows(with_rid=False)


# ------------------------------------------------------------------------------------------------------------
# rescoring (cigar2score, uberBlast.py:221-269)
# ------------------------------------------------------------------------------------------------------------
_CODON_WEIGHT = (9. / 7., 9. / 7., 3. / 7.)        # what a match is worth at the three codon positions in mode 3 (uberBlast.py:256)


def rescore_alignments(run_len, run_op, run_owner, q_cat, q_at, r_cat, r_at, first_base, mode, gap_open=6, gap_extend=1, table_id=11):
    """Identity and score of MANY alignments at once, all three modes of the reference's cigar2score (uberBlast.py:221-269).

    The alignments' CIGAR runs lie back to back - `run_len`, `run_op` (0 M, 1 I = query only, 2 D = reference only), `run_owner` = the
    alignment a run belongs to, ascending -; `q_cat` / `r_cat` hold the aligned stretches of all queries / references back to back in
    the rescoring alphabet (A 0, C 1, G 3, T 4, other 2; the reference strand already turned), alignment a starting at `q_at[a]` /
    `r_at[a]`; `first_base[a]` is its 1-based first query base (the codon phase of modes 2 / 3).
    Every column of every alignment becomes one element of flat arrays and every count is one np.bincount over them; the floating-point
    expressions keep the reference's order of operations (golden G5 holds the three modes to the last bit).
      mode 1  columns = M runs; identical bases, 3 / -1 per column, affine gaps
      mode 3  columns = M and I runs (an I column never matches) cut to whole codons from the query's phase on; matches weighted by position
      mode 2  the same columns as codons without an I column -> amino acids (table 11, or 4: TGA = W) -> identical residues, BLOSUM62
    -> (identity float64[n], score float64[n])"""
    n = len(q_at)
    run_len, run_op, run_owner = (np.asarray(x, dtype=np.int64) for x in (run_len, run_op, run_owner))
    gap = run_op != 0
    n_gap = np.bincount(run_owner, weights=gap, minlength=n)                             # float64 counts: exact, and what the formulas below take
    b_gap = np.bincount(run_owner, weights=run_len * gap, minlength=n)
    m_gap = np.bincount(run_owner, weights=run_len * (gap & (run_len > 3)), minlength=n)
    gap_cost_open, gap_cost_len = n_gap * (gap_open - gap_extend), b_gap * gap_extend
    # where every run starts inside its alignment's two stretches: exclusive sums of the bases it consumes, restarted per alignment
    first_run = np.searchsorted(run_owner, np.arange(n), side='left')

    def starts(step, base):
        upto = np.cumsum(step) - step
        return upto - upto[first_run][run_owner] + np.asarray(base, dtype=np.int64)[run_owner]
    q_run, r_run = starts(np.where(run_op != 2, run_len, 0), q_at), starts(np.where(run_op != 1, run_len, 0), r_at)
    shown = np.flatnonzero((run_op == 0) | ((run_op == 1) & (mode > 1)))                # the runs that contribute columns
    width = run_len[shown]
    col_run = np.repeat(shown, width)
    inside = np.arange(len(col_run)) - np.repeat(np.cumsum(width) - width, width)       # column number inside its run
    owner = run_owner[col_run]
    qa = np.asarray(q_cat, dtype=np.int64)[q_run[col_run] + inside]
    paired = run_op[col_run] == 0
    ra = np.full(len(col_run), -1, dtype=np.int64)
    ra[paired] = np.asarray(r_cat, dtype=np.int64)[(r_run[col_run] + inside)[paired]]
    if mode == 1:
        n_match = np.bincount(owner, weights=qa == ra, minlength=n)
        n_mis = np.bincount(owner, minlength=n) - n_match
        return n_match / (n_match + n_mis + b_gap - m_gap), n_match * 3 - n_mis * 1 - gap_cost_open - gap_cost_len
    # codon grid: column p of alignment a (counted from the phase on) is position p % 3 of codon p // 3; a trailing partial codon is dropped
    n_col = np.bincount(owner, minlength=n)
    phase = (np.asarray(first_base, dtype=np.int64) - 1) % 3
    p = np.arange(len(owner)) - (np.cumsum(n_col) - n_col)[owner] - phase[owner]
    whole = np.maximum(n_col - phase, 0) // 3
    keep = (p >= 0) & (p < 3 * whole[owner])
    owner, qa, ra, p = owner[keep], qa[keep], ra[keep], p[keep]
    if mode == 3:
        hit = np.bincount(owner * 3 + p % 3, weights=qa == ra, minlength=3 * n).reshape(n, 3)
        n_match = hit[:, 0] * _CODON_WEIGHT[0] + hit[:, 1] * _CODON_WEIGHT[1] + hit[:, 2] * _CODON_WEIGHT[2]
        n_mis = np.bincount(owner, weights=ra >= 0, minlength=n) - n_match
        return n_match / (n_match + n_mis + b_gap - m_gap), n_match * 3 - n_mis * 1 - gap_cost_open - gap_cost_len
    table = gtable
    if table_id == 4:
        table = gtable.copy()
        table[56] = 22            # TGA -> W; the reference patches its module-level table for good (uberBlast.py:223-224), here it is this call's
    codon = (np.cumsum(whole) - whole)[owner] + p // 3                                  # codons numbered through all alignments
    n_codon = int(whole.sum())
    place = np.array([25, 5, 1], dtype=np.int64)[p % 3]
    q_word = np.bincount(codon, weights=qa * place, minlength=n_codon).astype(np.int64)
    r_word = np.bincount(codon, weights=np.maximum(ra, 0) * place, minlength=n_codon).astype(np.int64)
    full = np.bincount(codon, weights=ra < 0, minlength=n_codon) == 0                    # no I column inside
    codon_owner = np.repeat(np.arange(n), whole)[full]
    q_aa, r_aa = table[q_word[full]], table[r_word[full]]
    n_match = np.bincount(codon_owner, weights=q_aa == r_aa, minlength=n) * 3.
    n_total = np.bincount(codon_owner, minlength=n) * 3. + b_gap - m_gap
    return n_match / n_total, np.bincount(codon_owner, weights=blosum62[(q_aa << 5) + r_aa], minlength=n) - gap_cost_open - gap_cost_len


def cigar2score(data):
    """(cigar, rSeq, qSeq, frame, mode, gapOpen, gapExtend, table_id) -> (identity, score): the reference's per-alignment entry point
    (uberBlast.py:221), one alignment through rescore_alignments.  RunBlast.reScore does not come through here: mode 1 is counted on the
    GPU (K7), modes 2 / 3 take all rows of the table in one call."""
    cigar, r_seq, q_seq, frame, mode, gap_open, gap_ext, table_id = data
    iden, score = rescore_alignments([n for n, op in cigar], [_OP_CODE[op] for n, op in cigar], np.zeros(len(cigar), dtype=np.int64),
                                     q_seq, [0], r_seq, [0], [frame], mode, gap_open, gap_ext, table_id)
    return iden[0], score[0]


class RunBlast(object):
    def __init__(self, device=None, sensitive=False):
        self.sensitive = bool(sensitive)    # four seed shapes in the translated search (pep_set_sensitivity level 1); default: the reference's diamond call (level 0)
        self.qrySeq = self.refSeq = None
        self._q_key = self._r_key = None    # cache keys of the files the sequences came from (None: handed over by the caller)
        self.device = device
        self._nt_loaded = {}                # id(context) -> what this instance left on it (one entry per context: the two tools run on two threads)
        self._batch = None                  # (reference names genome-major, genome id per name) in run_batch
        self._as_tables = False             # True: run() / run_batch() hand over the numeric HitTable instead of object rows
        # how the nucleotide tool treats the HSPs of one subject: 1 (default) every 64-diagonal band that reaches the threshold; 2 BLAST-like culling
        # (start / end cell shared with, or ranges inside, a better accepted HSP) and -num_alignments counted per subject (include/peppan_hip.h, hsp_mode)
        self.blast_hsp_mode = int(os.environ.get('PEPPAN_BLAST_HSP_MODE', '1'))

    # ---------------------------------------------------------------------------------------------- driver
    def run(self, ref, qry, methods, min_id, min_cov, min_ratio, table_id=11, n_thread=8, useProcess=False, re_score=0,
            filter=[False, 0.9, 0.], linear_merge=[False, 300., 1.2], return_overlap=[True, 300, 0.6], fix_end=[6., 6.]):
        self.min_id, self.min_cov, self.min_ratio = min_id, min_cov, min_ratio
        self.table_id, self.n_thread = table_id, n_thread
        self.pool = useProcess            # accepted for signature compatibility; the GPU path does not fan out
        tables = self._run_tools(methods, ref, qry, rescore=re_score)
        return self._post(tables, ref, qry, re_score, filter, linear_merge, return_overlap, fix_end, rescored=self._rescored_by_tools)

    def _tool_table(self):
        """tool name -> callable(ref, qry).  Inside run() the built-in tools hand over the numeric HitTable (no Python object per cell);
        the PUBLIC runBlast / runDiamond / runDiamondSELF keep the reference's plug-in contract - ndarray(object)[n, 15] (uberBlast.py:327,
        343-353) - and a subclass that overrides one of them is called through its override (its rows are converted on entry)."""
        def builtin(public):
            """is the tool this INSTANCE would call still the one defined here?  (the reference builds its dictionary from bound instance
            attributes, uberBlast.py:327: a subclass method and an attribute set on the instance both count)"""
            f = getattr(self, public)
            return getattr(f, '__func__', f) is _BUILTIN_TOOLS[public]
        blastn = self._runBlast_table if builtin('runBlast') else self.runBlast
        diamond = self._runDiamond_table if builtin('runDiamond') else self.runDiamond
        if not builtin('runDiamondSELF'):
            diamondself = self.runDiamondSELF
        elif builtin('runDiamond'):
            diamondself = lambda ref, qry: self._runDiamond_table(ref, qry, nhits=200, frames='F')
        else:                                   # the reference's runDiamondSELF goes through self.runDiamond (uberBlast.py:511): so does an override of it
            diamondself = lambda ref, qry: self.runDiamond(ref, qry, nhits=200, frames='F')
        return dict(blastn=blastn, diamond=diamond, diamondself=diamondself, gpu=diamond)

    def _run_tools(self, methods, ref, qry, rescore=0, nt_match=False):
        """the tools of one run in the order given.  A tool that fails is reported and the other tools' tables are kept - the reference's
        convention (uberBlast.py:347-349) - but never silently: `failed_tools` lists (tool, message) of this run, and a PEP_ERR_LIMIT
        (an input beyond a documented limit) is raised instead of costing a whole table.
        rescore=1 (-s 1): when the tools run side by side, each rescores its own table (K7 is a function of the row alone, uberBlast.py:397-415)
        in its own context while the other tool is still searching; `_rescored_by_tools` tells _post that only the identity cut is left."""
        tools = self._tool_table()
        tables, self.failed_tools = [], []
        self._rescored_by_tools = False
        self._nucl_searched = None
        self._want_nt_match = bool(nt_match) and os.environ.get('PEPPAN_NT_MATCH_IN_SEARCH', '1') != '0'       # (run_batch: every tool's table is rescored, mode 1)
        todo = [m for m in methods if m.lower() in tools]

        def attempt(method):
            """one tool -> ('ok', table) | ('limit', exception) | ('failed', method, message)"""
            try:
                T = _as_table(tools[method.lower()](ref, qry))
                if self._rescored_by_tools and not T.rescored:
                    try:
                        ctx = get_nucl_context(self.device) if method.lower() == 'blastn' else get_context(self.device)
                        T = self._rescore_table(ref, qry, T, 1, None, self.table_id, cut=False, ctx=ctx)
                    except BaseException as e:    # (a failure of the rescoring is the run's, not the tool's: it is raised once both tools are back)
                        return ('limit', e)
                return ('ok', T)
            except N.PepError as e:
                if 'PEP_ERR_LIMIT' in str(e) or '(-3)' in str(e):
                    return ('limit', e)           # an input beyond a documented limit must not silently cost a whole tool's hits
                import traceback
                traceback.print_exc()
                return ('failed', method, str(e))
            except Exception as e:
                import traceback
                traceback.print_exc()
                return ('failed', method, repr(e))
        # Both tools in one run, each in a context of its own (get_nucl_context): the nucleotide tool goes to a second thread, the translated tool(s) stay
        # on this one; the outcomes are taken in the order the reference runs them.  Built-in tools only - a caller's override is called one after the other.
        names = [m.lower() for m in todo]
        plain = all(getattr(getattr(self, p), '__func__', getattr(self, p)) is _BUILTIN_TOOLS[p] for p in ('runBlast', 'runDiamond', 'runDiamondSELF'))
        outcomes, side = {}, None
        if self._batch is None and plain and 'blastn' in names and len(set(names)) > 1 and get_nucl_context(self.device) is not get_context(self.device):
            import threading
            self._load(ref, qry)
            self._ensure_nt(get_context(self.device), 3 if set(names) <= {'blastn', 'diamondself'} else 6)       # (shared preparations first: the threads only read them)
            self._ensure_nt(get_nucl_context(self.device))
            k_side = names.index('blastn')
            self._rescored_by_tools = rescore == 1
            self._want_nt_match = rescore == 1 and os.environ.get('PEPPAN_NT_MATCH_IN_SEARCH', '1') != '0'
            # Whose turn the GPU is.  The two searches together take the GPU as long side by side as one after the other (6.0 ms of kernels at 10 000 genes either
            # way), so what counts is which tool's host chain starts first: the nucleotide tool has the longer one behind its search (table 1.6 + K7 1.1 ms against
            # 0.55 + 1.0), so its search goes first and the translated search starts when it is back - the nucleotide table is then built while the translated search
            # has the GPU to itself.  PEPPAN_TOOL_TURNS=0: both searches at once (round 5).
            turn = self._nucl_searched = threading.Event() if os.environ.get('PEPPAN_TOOL_TURNS', '1') != '0' else None

            def side_tool():
                try:
                    outcomes[k_side] = attempt(todo[k_side])
                finally:
                    if turn is not None:
                        turn.set()                # (also when the tool failed before or inside its search)
            side = threading.Thread(target=side_tool)
            # (a thread that comes back from the library waits for the interpreter lock until the other one gives it up: at the default 5 ms
            # between such requests the two tools cost more side by side than one after the other - 30.5 against 29.1 ms per call; at 0.1 ms 24.0)
            interval = sys.getswitchinterval()
        try:
            if side is not None:
                sys.setswitchinterval(1e-4)
                side.start()
            for k, method in enumerate(todo):
                if side is None or k != k_side:
                    outcomes[k] = attempt(method)
                    if side is None and outcomes[k][0] != 'ok':
                        break                     # (one after the other: nothing runs behind a failure)
        finally:
            if side is not None:
                if side.ident is not None:
                    side.join()
                sys.setswitchinterval(interval)
        for k in range(len(todo)):
            o = outcomes.get(k)
            if o is None:
                break
            if o[0] == 'ok':
                tables.append(o[1])
            elif o[0] == 'limit':
                raise o[1]
            else:
                self.failed_tools.append((o[1], o[2]))
                break                             # (the reference's try block ends with the first failure, uberBlast.py:343-349: what ran behind it does not count)
        if self.failed_tools:
            logger('WARNING: {0} of {1} search tools failed: {2}'.format(len(self.failed_tools), len(methods), ', '.join(m for m, _ in self.failed_tools)))
        return tables

    def _post(self, tables, ref, qry, re_score, filter, linear_merge, return_overlap, fix_end, rescored=False):
        """everything RunBlast.run does after the tools returned (uberBlast.py:352-376), on the numeric table; the object rows the
        caller gets are made at the very end.  rescored: the tables carry the rescored identity / score already (run_batch does K7 for all
        genomes of a batch at once) and only the identity cut is left"""
        steps = self._post_steps(tables, ref, qry, re_score, filter, linear_merge, return_overlap, fix_end, rescored)
        try:
            T = next(steps)                                  # (-O: the table whose overlaps are wanted)
            steps.send(mapfilters.overlaps_table(T, return_overlap[1], return_overlap[2], sweep=get_context(self.device).overlaps))
        except StopIteration as e:
            return e.value
        raise RuntimeError('RunBlast._post: the steps asked twice')

    def _post_steps(self, tables, ref, qry, re_score, filter, linear_merge, return_overlap, fix_end, rescored=False):
        """_post as a generator: with -O it yields the table whose overlaps are wanted ONCE and is sent them (run_batch sweeps the tables of all
        genomes of a batch in one K11 call); its return value is _post's"""
        T = HitTable.concat(tables)
        if len(T) == 0:
            none = HitTable.empty() if self._as_tables else np.empty([0, 16], dtype=object)
            return (none, np.empty([0, 3], dtype=int)) if return_overlap[0] else none
        T.rid = np.arange(len(T), dtype=np.int64)
        plain = not (filter[0] or linear_merge[0] or return_overlap[0])
        keep = None
        if re_score and not rescored:
            T = self._rescore_table(ref, qry, T, re_score, self.min_id, self.table_id, cut=not plain)
        if re_score and (rescored or plain):
            # the identity cut of reScore (uberBlast.py:414).  With nothing between it and the final sort that looks at other rows (-f, -m, -O), the
            # cut rides on the sort's gather: fixEnd is row-local, one take instead of two
            if plain:
                keep = T.iden >= self.min_id
            else:
                T = T.take(T.iden >= self.min_id)
        if filter[0]:
            T = mapfilters.ovl_filter_table(T, filter[1], filter[2])
        if linear_merge[0]:
            T = mapfilters.linear_merge_table(T, linear_merge[1], linear_merge[2])
        T.fix_end(*fix_end)
        overlap = None
        if return_overlap[0]:
            overlap = yield T
        order = T.final_order()
        T = T.take(order if keep is None else order[keep[order]])
        rows = T if self._as_tables else T.to_rows(cigar='str')
        return (rows, overlap) if return_overlap[0] else rows

    MAX_BATCH_NT = 120000000        # nucleotides per search: 6 frames -> 2 packed protein bytes per nt, under the 2^29 limit

    def run_batch(self, refs, qry, methods, min_id, min_cov, min_ratio, table_id=11, n_thread=8, useProcess=False, re_score=0,
                  filter=[False, 0.9, 0.], linear_merge=[False, 300., 1.2], return_overlap=[True, 300, 0.6], fix_end=[6., 6.]):
        """run() for MANY reference files (genomes) against one query file with ONE search per tool and sub-batch: the
        references are packed genome-major, the GPU ranks hits inside each genome (pep_set_target_groups), and the tables
        are split per genome before the post-processing chain - each result equals run(ref_i, qry, ...).  This is the
        GPU-native form of PEPPAN's per-genome fan-out (PEPPAN.py:907-922, iter_map_bsn :759-772): the query index is
        shared and no worker process ever touches the device.  Genomes are grouped into sub-batches of at most
        MAX_BATCH_NT nucleotides so that a packed reference set stays inside the library's 2^29-byte limit."""
        refs = [_reference_seqs(p) for p in refs]                  # (paths are read once, here)
        sizes = [sum(len(v) for v in rs.values()) for rs in refs]
        out, start, failed = [], 0, []
        while start < len(refs):
            stop, tot = start, 0
            while stop < len(refs) and (stop == start or tot + sizes[stop] <= self.MAX_BATCH_NT):
                tot += sizes[stop]
                stop += 1
            sub = RunBlast(self.device, sensitive=self.sensitive) if (start, stop) != (0, len(refs)) else self
            sub._as_tables = self._as_tables
            out += sub._run_one_batch(refs[start:stop], qry, methods, min_id, min_cov, min_ratio, table_id, n_thread, useProcess, re_score,
                                      filter, linear_merge, return_overlap, fix_end)
            failed += sub.failed_tools
            start = stop
        self.failed_tools = failed               # (of every sub-batch)
        return out

    def _run_one_batch(self, refs, qry, methods, min_id, min_cov, min_ratio, table_id, n_thread, useProcess, re_score,
                       filter, linear_merge, return_overlap, fix_end):
        self.min_id, self.min_cov, self.min_ratio = min_id, min_cov, min_ratio
        self.table_id, self.n_thread, self.pool = table_id, n_thread, useProcess
        self.qrySeq, self._q_key = _read_cached(qry, with_key=True)
        self._r_key = None
        combined, names, groups = {}, [], []
        for g, rs in enumerate(refs):
            if not isinstance(rs, dict):
                rs = _reference_seqs(rs)
            for n in sorted(rs):
                if n in combined:
                    raise ValueError('run_batch: reference sequence name {0} occurs in more than one file'.format(n))
                combined[n] = rs[n]
                names.append(n)
                groups.append(g)
        self.refSeq, self._batch = combined, (names, groups)
        genome_of = dict(zip(names, groups))
        with _gpu_gate():
            tables = self._run_tools(methods, None, None, nt_match=re_score == 1)
        # mode-1 rescoring is a function of the row alone: K7 once per tool over the rows of ALL genomes (a launch and a round trip per genome
        # otherwise: 84 us of GPU and a synchronisation each, sixteen times per batch), the identity cut stays with the genome's table (_post)
        batch_rescore = re_score == 1
        if batch_rescore:
            for T in tables:
                if not T.rescored:                # (a built-in tool's table comes rescored: its search counted the identical columns, _search)
                    self._rescore_table(None, None, T, 1, None, table_id, cut=False)
        # rows by genome (the reference set of a row's reference sequence): the tools' tables are put one behind the other ONCE per batch - their
        # name tables are the same objects, their arenas end up side by side - and a genome takes its rows of every tool, in tool and table order, with one
        # gather (a gather per tool and a concatenation per genome copied both tools' whole CIGAR arenas sixteen times per batch)
        T = HitTable.concat(tables)
        if len(T):
            owner = np.array([genome_of[r] for r in T.r_tab], dtype=np.int64)[T.ri]
            order = np.argsort(owner, kind='stable')
            cuts = np.searchsorted(owner[order], np.arange(len(refs) + 1))
        # every genome's chain up to its overlaps, then ONE K11 sweep over the tables of all genomes (mapfilters.overlaps_tables), then the chains' ends
        steps, out, asked = [], [None] * len(refs), []
        for g in range(len(refs)):
            part = [T.take(order[cuts[g]:cuts[g + 1]])] if len(T) else []
            st = self._post_steps(part, None, None, re_score, filter, linear_merge, return_overlap, fix_end, rescored=batch_rescore)
            steps.append(st)
            try:
                asked.append((g, next(st)))
            except StopIteration as e:
                out[g] = e.value
        if asked:
            swept = mapfilters.overlaps_tables([t for _, t in asked], return_overlap[1], return_overlap[2], get_context(self.device).overlaps)
            for (g, _), ovl in zip(asked, swept):
                try:
                    steps[g].send(ovl)
                    raise RuntimeError('RunBlast._post_steps asked twice')
                except StopIteration as e:
                    out[g] = e.value
        return out

    # ---------------------------------------------------------------------------------------------- inputs
    def _load(self, ref, qry):
        if self._batch is not None:
            return                          # run_batch filled qrySeq / refSeq itself
        if not self.qrySeq:
            (self.qrySeq, self._q_key), self.qryQual = _read_cached(qry, with_key=True), None
        if not self.refSeq:
            (self.refSeq, self._r_key), self.refQual = _read_cached(ref, with_key=True), None

    def _ensure_nt(self, ctx, frames=None):
        """this instance's nucleotide sets on the device (K1 and K7 read them there).  sorted(name) order is the order in which the
        reference writes its FASTA files (uberBlast.py:527, 537), so sequence / target indices follow it and the 5-way split
        membership is reproduced.  The context is shared by every RunBlast of the process: what this instance uploaded is only
        still there while the context's upload generation is the one it left behind (frames None = any frame count will do)."""
        gen = getattr(ctx, 'upload_generation', 0)
        done = self._nt_loaded.get(id(ctx))
        if done is not None and done[2] == gen and done[1] == self.table_id and frames in (None, done[0]):
            return
        frames = frames or 6
        q = _prepare_side(self.qrySeq, getattr(self, '_q_key', None))
        r = _prepare_side(self.refSeq, getattr(self, '_r_key', None), names=None if self._batch is None else self._batch[0])
        self.q_names, self.q_index, self._q_tab, self._q_sorted, self._q_len = q['names'], q['index'], q['tab'], q['sorted'], q['lens']
        self.r_names, self.r_index, self._r_tab, self._r_len = r['names'], r['index'], r['tab'], r['lens']
        self._r_sorted = r['sorted'] and self._batch is None
        # a side the context still holds from an earlier call on the same file (PEPPAN maps one gene file against every genome) is not sent again
        q_token = (getattr(self, '_q_key', None), self.table_id) if getattr(self, '_q_key', None) is not None else None
        r_token = (getattr(self, '_r_key', None), frames, self.table_id) if getattr(self, '_r_key', None) is not None else None
        if q_token is None or getattr(ctx, 'q_nt_token', None) != q_token:
            ctx.set_query_nt(q['packed'], self.table_id)
            ctx.q_nt_token = q_token
        if r_token is None or getattr(ctx, 'r_nt_token', None) != r_token:
            ctx.set_ref_nt(r['packed'], frames, self.table_id)
            ctx.r_nt_token = r_token
        ctx.set_target_groups(None if self._batch is None else self._batch[1])
        self._nt_loaded[id(ctx)] = (frames, self.table_id, getattr(ctx, 'upload_generation', 0))

    @staticmethod
    def _text(s):
        return s if isinstance(s, (str, bytes)) else ''.join('ACNGT'[int(x)] for x in s)

    # ---------------------------------------------------------------------------------------------- tools
    # The three public tools keep the reference's plug-in contract (uberBlast.py:327): method(ref, qry) -> ndarray(object)[n, 15], names as
    # str, CIGAR as [[n, op], ...] in nucleotides, rows in any order - what the reference's own run() loop vstacks (uberBlast.py:343-354).
    def runDiamondSELF(self, ref, qry):
        return self.runDiamond(ref, qry, nhits=200, frames='F')                 # (through self.runDiamond, like uberBlast.py:511)

    def runDiamond(self, ref, qry, nhits=10, frames='7'):
        return self._runDiamond_table(ref, qry, nhits, frames).to_rows(with_rid=False)

    def runBlast(self, ref, qry):
        return self._runBlast_table(ref, qry).to_rows(with_rid=False)

    def _search(self, ctx, params):
        """ctx.search for a tool of this run -> (hits, cigar, stats, nt_match).  With -s 1 (`_want_nt_match`, set by the run) the search also counts the identical
        nucleotide columns of every hit - K7's one count that needs the sequences, from the table on the device, inside the search's own wait - and the table
        builder turns them into the rescored identity and score: no second upload of the table, no K7 round trip, no pass of numpy expressions per tool"""
        want = bool(getattr(self, '_want_nt_match', False)) and hasattr(ctx, 'set_nt_match')       # (the CPU tests' oracle-backed context has no such switch: K7 afterwards, as before)
        if not want:
            return ctx.search(params, copy=False) + (None,)
        ctx.set_nt_match(True)
        try:
            hits, cigar, stats = ctx.search(params, copy=False)
        finally:
            ctx.set_nt_match(False)
        return hits, cigar, stats, (ctx.last_nt_match if len(hits) else np.zeros(0, np.uint32))

    def _runDiamond_table(self, ref, qry, nhits=10, frames='7'):
        """translated search on the GPU: K1 translate/pack, K2-K4 seeds, K5/K6 banded Smith-Waterman + traceback,
        K8 filters/top-k; thresholds as on the reference's diamond command line (uberBlast.py:550)"""
        logger('Run diamond starts')
        self._load(ref, qry)
        ctx = get_context(self.device)
        self._ensure_nt(ctx, 6 if frames == '7' else 3)
        params = N.default_params(min_id_pct=self.min_id * 100., min_qcov_pct=self.min_ratio * 100., top_k=nhits, n_splits=5,
                                  dbsize=5000000., max_evalue=1., sensitive=self.sensitive)
        ctx.translate()                                                # (K1; a nucleotide search before this one left base codes in the packed sets)
        if getattr(self, '_nucl_searched', None) is not None:
            self._nucl_searched.wait()                                 # (_run_tools: the nucleotide tool's search goes first)
        hits, cigar, stats, nt_match = self._search(ctx, params)       # consumed at once by the table builder below
        table = hits_to_table(hits, cigar, ctx.query_meta(), ctx.target_meta(), self._q_tab, self._r_tab, self._q_len, self._r_len,
                              self.min_id, self.min_cov, self.min_ratio, nt_match=nt_match)
        table.q_sorted, table.r_sorted = self._q_sorted, self._r_sorted
        self.last_stats = stats
        logger('Run diamond finishes. Got {0} alignments'.format(len(table)))
        return table

    def _runBlast_table(self, ref, qry):
        """nucleotide search on the same GPU engine, configured like the reference's blastn call (uberBlast.py:294):
        exact 17-mer seeds, +2/-3, gap 6+2k, e-value 1e-2 at dbsize 5e6, both strands of the reference, the
        -perc_identity / -qcov_hsp_perc cuts, then parseBlast's filters (uberBlast.py:283).  hsp_mode 1: every
        64-diagonal band of a (query, subject strand) that reaches the score threshold yields an alignment."""
        logger('Run BLASTn starts')
        self._load(ref, qry)
        ctx = get_nucl_context(self.device) if self._batch is None else get_context(self.device)        # (a batch of genomes: one context, tool after tool)
        params = N.nucleotide_params(min_id_pct=self.min_id * 100., min_qcov_pct=self.min_ratio * 100., hsp_mode=self.blast_hsp_mode)
        side = _SIDE_CACHE.get(getattr(self, '_r_key', None)) if self._batch is None else None      # (the prepared side knows the lengths: 10 000 len() calls are 0.3 ms of the hot call's start)
        longest = (int(side['lens'].max()) if len(side['lens']) else 0) if side is not None else max(map(len, self.refSeq.values()), default=0)
        if longest <= N.MAX_SEQ_LEN:
            # the usual case: the nucleotide sets that K1 and K7 read on the device are packed THERE into the base-code residue sets of this
            # search (forward strands, then reverse complements, per reference set) - no encoding, concatenation or upload on the host
            self._ensure_nt(ctx)
            ctx.use_nt_as_residues(2)
            hits, cigar, stats, nt_match = self._search(ctx, params)       # consumed at once by the table builder below
            if getattr(self, '_nucl_searched', None) is not None:
                self._nucl_searched.set()                                  # (_run_tools: the translated tool's search may start)
            tm = ctx.target_meta()
            table = blast_hits_to_table(hits, cigar, self._q_tab, self._r_tab, self._q_len, self._r_len, self.min_id, self.min_cov, self.min_ratio, params,
                                        tm['seq'].astype(np.int64), tm['frame'] > 3, nt_match=nt_match)
            table.q_sorted, table.r_sorted = self._q_sorted, self._r_sorted
            logger('Run BLASTn finishes. Got {0} alignments'.format(len(table)))
            return table
        q_names = sorted(self.qrySeq)
        r_names = sorted(self.refSeq) if self._batch is None else list(self._batch[0])
        groups = [0] * len(r_names) if self._batch is None else list(self._batch[1])
        # (the query file's codes are kept: the mapping path comes back with the same 50 000 genes for every batch of genomes, 60 ms each time)
        q_key = getattr(self, '_q_key', None)
        if q_key is not None and _Q_CODES.get('key') == q_key:
            q_codes, q_off = _Q_CODES['codes']
        else:
            q_codes, q_off = _encode_nt([self._text(self.qrySeq[n]) for n in q_names])
            if q_key is not None:
                _Q_CODES.update(key=q_key, codes=(q_codes, q_off))
        r_codes, r_off = _encode_nt([self._text(self.refSeq[n]) for n in r_names])
        q_len, r_len = np.diff(q_off.astype(np.int64)), np.diff(r_off.astype(np.int64))
        # targets: per reference set all forward strands, then all reverse strands.  The reverse complement of the WHOLE concatenation
        # holds every sequence's reverse complement (in reverse sequence order): sequence i sits at [total - off[i+1], total - off[i])
        total = int(r_off[-1])
        r_rc = _NT_RC[r_codes[::-1]]
        t_seq, t_rev, t_grp, parts = [], [], [], []
        g_arr = np.asarray(groups, dtype=np.int64)
        ro = r_off.astype(np.int64).tolist()
        # a strand longer than the engine's sequence limit (PEP_MAX_SEQ_LEN, 8.39 Mbp: Streptomyces, any eukaryote) is searched as
        # overlapping windows: window w owns the "home" stretch [w * HOME, (w + 1) * HOME) and carries a halo of one query length + band
        # width on both sides, so every alignment whose midpoint lies in the home stretch has its whole 128-diagonal band inside the
        # window; window starts are multiples of 64, which keeps the diagonal bins - and therefore the alignments - those of the
        # unsplit search.  Hits are kept by the window whose home stretch holds their midpoint (blast_hits_to_table).
        HOME = _TILE_HOME
        halo = ((int(q_len.max()) if len(q_len) else 0) + 256 + 63) // 64 * 64
        tiled = bool(len(r_len)) and int(r_len.max()) > N.MAX_SEQ_LEN
        if self.blast_hsp_mode == 2:
            # hsp_mode 2 (unpinned, an option) culls and counts per SUBJECT; on this path every strand and every window of a long sequence is a
            # target of its own, i.e. its own subject: the list would silently mean something else than on the usual path
            raise N.PepError('runBlast: PEPPAN_BLAST_HSP_MODE=2 is not defined for reference sequences beyond %d nt (searched as windows: every window '
                             'would be a subject of its own); use the default hsp_mode 1' % N.MAX_SEQ_LEN)
        if tiled and HOME + 2 * halo > N.MAX_SEQ_LEN:
            raise N.PepError('runBlast: queries of %d nt are too long to tile a %d nt reference sequence (PEP_ERR_LIMIT)' % (int(q_len.max()), int(r_len.max())))
        t_woff, t_hlo, t_hhi = [], [], []
        for g in sorted(set(groups)):
            members = np.flatnonzero(g_arr == g).tolist()
            if not tiled:
                # the members of a set are consecutive sequences: their forward strands are one slice of the concatenation
                parts.append(r_codes[ro[members[0]]:ro[members[-1] + 1]])
                parts += [r_rc[total - ro[i + 1]:total - ro[i]] for i in members]
                t_seq += members + members
                t_rev += [False] * len(members) + [True] * len(members)
                t_grp += [g] * (2 * len(members))
                continue
            for rev in (False, True):
                for i in members:
                    L = ro[i + 1] - ro[i]
                    strand = r_rc[total - ro[i + 1]:total - ro[i]] if rev else r_codes[ro[i]:ro[i + 1]]
                    if L <= N.MAX_SEQ_LEN:
                        wins = [(0, L, 0, 1 << 62)]
                    else:
                        wins = [(max(0, h - halo), min(L, h + HOME + halo), h, (h + HOME) if h + HOME < L else (1 << 62)) for h in range(0, L, HOME)]
                    for lo, hi, hlo, hhi in wins:
                        parts.append(strand[lo:hi])
                        t_seq.append(i); t_rev.append(rev); t_grp.append(g)
                        t_woff.append(lo); t_hlo.append(hlo); t_hhi.append(hhi)
        if tiled:
            t_len = np.fromiter(map(len, parts), dtype=np.int64, count=len(parts))
        else:
            t_len = r_len[np.asarray(t_seq, dtype=np.int64)] if t_seq else np.zeros(0, np.int64)
        t_off = np.zeros(len(t_len) + 1, dtype=np.uint64)
        t_off[1:] = np.cumsum(t_len)
        ctx.set_query_aa((q_codes, q_off))
        ctx.set_ref_aa((np.concatenate(parts) if parts else np.zeros(0, np.uint8), t_off))
        ctx.set_target_groups(None if self._batch is None else t_grp)
        self._nt_loaded.pop(id(ctx), None)          # the packed sets THIS context held from an earlier search are gone (only this context's record: the
        #                                             other tool's thread may be reading or writing its own entry at this moment)
        hits, cigar, stats = ctx.search(params, copy=False)            # consumed at once by the table builder below
        if getattr(self, '_nucl_searched', None) is not None:
            self._nucl_searched.set()
        table = blast_hits_to_table(hits, cigar, q_names, r_names, q_len, r_len,
                                    self.min_id, self.min_cov, self.min_ratio, params, np.array(t_seq, dtype=np.int64), np.array(t_rev, dtype=bool),
                                    windows=(np.array(t_woff, dtype=np.int64), np.array(t_hlo, dtype=np.int64), np.array(t_hhi, dtype=np.int64)) if tiled else None)
        logger('Run BLASTn finishes. Got {0} alignments'.format(len(table)))
        return table

    # ---------------------------------------------------------------------------------------------- post-processing
    def reScore(self, ref, qry, blastab, mode, min_id, table_id=11, perBatch=10000):
        """recompute identity / score of every hit from its CIGAR over the nucleotide sequences (uberBlast.py:397-415); object rows in
        and out (the numeric chain inside run() calls _rescore_table directly)"""
        if blastab.shape[0] == 0:
            return blastab
        return self._rescore_table(ref, qry, HitTable.from_rows(blastab), mode, min_id, table_id).to_rows()

    def _rescore_table(self, ref, qry, T, mode, min_id, table_id=11, cut=True, ctx=None):
        """Mode 1: integer counts on the GPU (K7), float arithmetic and np.round in float64 here.  Modes 2 / 3 (amino-acid / codon-position
        scoring, not used by PEPPAN's calls) walk the rows on the host.  cut=False: identity and score are replaced in place and every row stays"""
        self._load(ref, qry)
        if len(T) == 0:
            return T
        if mode == 1:
            ctx = ctx or get_context(self.device)
            self._ensure_nt(ctx)
            h = np.zeros(len(T), dtype=N.NT_HIT_DTYPE)
            # (a table of this instance's own tools carries the name tables the sides were prepared with: row codes ARE sequence indices)
            h['q'] = T.qi if T.q_tab is self._q_tab else np.array([self.q_index[str(x)] for x in T.q_tab], dtype=np.int64)[T.qi]
            h['r'] = T.ri if T.r_tab is self._r_tab else np.array([self.r_index[str(x)] for x in T.r_tab], dtype=np.int64)[T.ri]
            h['qs'], h['qe'], h['rs'], h['re'] = T.qs, T.qe, T.ss, T.se
            h['cigar_runs'], h['cigar_off'] = T.c_runs, T.c_off
            c = ctx.rescore_nt(h, T.arena).astype(np.int64)
            n_match, n_mis, n_gap, b_gap, m_gap = c.T
            iden = n_match.astype(np.float64) / (n_match + n_mis + b_gap - m_gap)
            score = (n_match * 3 - n_mis - n_gap * (6 - 1) - b_gap * 1).astype(np.float64)
        else:
            # the aligned stretch of every row's query and reference (reverse strand: complemented and turned), back to back, then ONE call
            enc = {}

            def codes(side, name):
                key = (side, str(name))
                if key not in enc:
                    enc[key] = nucEncoder[np.frombuffer((self.qrySeq if side == 'q' else self.refSeq)[name].encode('latin-1'), dtype=np.uint8)]
                return enc[key]
            q_parts, r_parts = [], []
            for qn, rn, qs, qe, ss, se in zip((T.q_tab[i] for i in T.qi.tolist()), (T.r_tab[i] for i in T.ri.tolist()), T.qs.tolist(), T.qe.tolist(), T.ss.tolist(), T.se.tolist()):
                q_parts.append(codes('q', qn)[qs - 1:qe])
                r = codes('r', rn)
                r_parts.append(r[ss - 1:se] if ss < se else 4 - r[se - 1:ss][::-1])
            at = lambda parts: np.concatenate([[0], np.cumsum([len(x) for x in parts])[:-1]]).astype(np.int64)
            pick = np.concatenate([np.arange(o, o + k) for o, k in zip(T.c_off.tolist(), T.c_runs.tolist())]) if len(T) else np.zeros(0, np.int64)
            iden, score = rescore_alignments(T.arena[pick] >> 2, T.arena[pick] & 3, np.repeat(np.arange(len(T)), T.c_runs), np.concatenate(q_parts), at(q_parts),
                                             np.concatenate(r_parts), at(r_parts), T.qs, mode, 6, 1, table_id)
        T.iden, T.score = np.round(iden, 3), np.round(score, 3)
        T.score_is_int = False
        return T.take(T.iden >= min_id) if cut else T

    def ovlFilter(self, blastab, params):
        return mapfilters.ovl_filter(blastab, params[1], params[2])

    def linearMerge(self, blastab, params):
        return mapfilters.linear_merge(blastab, params[1], params[2])

    def returnOverlap(self, blastab, param):
        """interval sweep on the GPU (K11); the host only sorts the intervals the way the reference does"""
        return mapfilters.overlaps(blastab, param[1], param[2], sweep=get_context(self.device).overlaps)

    def fixEnd(self, blastab, se, ee):
        """extend alignments over short unaligned ends (<= se at the query head, <= ee at its tail) and turn the CIGAR
        into its string form, in place; identity and score stay as they are (uberBlast.py:462-480)"""
        if blastab.shape[0] == 0:
            return
        T = HitTable.from_rows(blastab)
        T.fix_end(se, ee)
        for col, vals in ((6, T.qs), (7, T.qe), (8, T.ss), (9, T.se)):
            blastab[:, col] = vals.tolist()
        blastab[:, 14] = T.cigar_strings()


_BUILTIN_TOOLS = {name: getattr(RunBlast, name) for name in ('runBlast', 'runDiamond', 'runDiamondSELF')}     # as defined above: run() bypasses their object rows


def _as_table(x):
    """a tool's product as a HitTable: the GPU tools return one, a plug-in following the reference's contract returns object rows"""
    return x if isinstance(x, HitTable) else HitTable.from_rows(x)


# one flag table for both command-line front ends: (flags, keyword arguments of add_argument).  Names, defaults and meaning
# are the reference's command line (uberBlast.py:565-591) because callers pass these strings verbatim (PEPPAN.py:229, 771).
_TOOL_FLAGS = (
    (('--blastn',), dict(action='store_true', default=False, help='run the nucleotide search (17-mer seeds, +2/-3, both strands) on the GPU')),
    (('--diamond',), dict(action='store_true', default=False, help='run the translated search (6 reference frames, BLOSUM62) on the GPU')),
    (('--diamondSELF',), dict(action='store_true', default=False, help='translated search against the 3 forward frames only, 200 targets per query')),
    (('--gpu',), dict(action='store_true', default=False, help='same as --diamond')),
    (('--sensitive',), dict(action='store_true', default=False, help='translated search with four seed shapes instead of two (not a flag of the reference: recall between 0.45 and 0.7 '
                                                                      'identity 0.93 -> 0.985 at twice the seed-stage cost; the default matches the reference\'s diamond call, which runs at default sensitivity)')),
)
_SEARCH_FLAGS = (
    (('--gtable',), dict(type=int, default=11, help='translation table: 11 (bacteria) or 4 (Mycoplasma, TGA codes W) [11]')),
    (('--min_id',), dict(type=float, default=0.3, help='lowest identity of a reported alignment, before any rescoring [0.3]')),
    (('--min_cov',), dict(type=float, default=40., help='shortest reported alignment, in query nucleotides [40]')),
    (('--min_ratio',), dict(type=float, default=0.05, help='shortest reported alignment as a fraction of the query length [0.05]')),
    (('-s', '--re_score'), dict(type=int, default=0, help='recompute identity and score from the nucleotides: 0 off, 1 per base, 2 per amino acid, 3 per codon position [0]')),
    (('-f', '--filter'), dict(action='store_true', default=False, help='of two alignments of one query that overlap on the reference keep the better one')),
    (('--filter_cov',), dict(type=float, default=0.9, help='overlap (fraction of either alignment) at which -f acts [0.9]')),
    (('--filter_score',), dict(type=float, default=0., help='score margin the better alignment needs for -f to act [0]')),
    (('-m', '--linear_merge'), dict(action='store_true', default=False, help='chain collinear alignments of one query into merged groups (column 17)')),
    (('--merge_gap',), dict(type=float, default=600., help='largest distance between chained alignments [600]')),
    (('--merge_diff',), dict(type=float, default=1.5, help='largest ratio of the query and reference distances inside a chain [1.5]')),
    (('-O', '--return_overlap'), dict(action='store_true', default=False, help='also return the pairs of alignments that overlap on the reference')),
    (('--overlap_length',), dict(type=float, default=300, help='an overlap this long is always reported [300]')),
    (('--overlap_proportion',), dict(type=float, default=0.6, help='... or this fraction of either alignment [0.6]')),
    (('-e', '--fix_end'), dict(default='0,0', help='L,R: stretch an alignment to the query ends when at most L / R bases are left unaligned there [0,0]')),
    (('-t', '--n_thread'), dict(type=int, default=1, help='kept for command-line compatibility (the search runs on the GPU)')),
    (('-p', '--process'), dict(action='store_true', default=False, help='kept for command-line compatibility')),
)


_PARSERS = {}


def _parser(description, with_reference):
    """the argument parser of uberBlast / uberBlastBatch, built once per process (PEPPAN calls uberBlast once per genome: a millisecond each time)"""
    if (description, with_reference) in _PARSERS:
        return _PARSERS[description, with_reference]
    import argparse
    ap = _PARSERS[description, with_reference] = argparse.ArgumentParser(description=description)
    if with_reference:
        ap.add_argument('-r', '--reference', required=True, help='FASTA / FASTQ file searched against (gzip accepted)')
        ap.add_argument('-o', '--output', default=None, help='write the table as tab-separated text to this file, or STDOUT')
        ap.add_argument('--device', type=int, default=None, help='HIP device index [$LOCAL_RANK, else 0]')
    ap.add_argument('-q', '--query', required=True, help='FASTA / FASTQ file with the query sequences')
    for flags, kw in _TOOL_FLAGS + _SEARCH_FLAGS:
        ap.add_argument(*flags, **kw)
    return ap


def _run_arguments(a):
    """parsed flags -> (tools in the order the reference runs them, keyword arguments of RunBlast.run / run_batch)"""
    wanted = (('blastn', a.blastn), ('diamond', a.diamond or a.gpu), ('diamondSELF', a.diamondSELF))
    ends = a.fix_end.split(',')
    ends[-2:] = [float(x) for x in ends[-2:]]
    kw = dict(re_score=a.re_score, filter=[a.filter, a.filter_cov, a.filter_score], linear_merge=[a.linear_merge, a.merge_gap, a.merge_diff],
              return_overlap=[a.return_overlap, a.overlap_length, a.overlap_proportion], fix_end=ends)
    return [name for name, on in wanted if on], kw


def uberBlast(args, extPool=None, as_table=False):
    """command-line style entry point (reference: uberBlast.py:564): list of flags -> hit table, or (table, overlaps) with -O.
    as_table (not in the reference): the table comes as the numeric HitTable - same rows, same order, no Python object per cell - for
    callers inside this package that work on the columns (pipeline.get_similar_pairs)."""
    a = _parser('Similarity search of query sequences against a reference on an MI355X; table format of PEPPAN uberBlast.', True).parse_args(args)
    methods, kw = _run_arguments(a)
    rb = RunBlast(a.device, sensitive=a.sensitive)
    rb._as_tables = bool(as_table)
    data = rb.run(a.reference, a.query, methods, a.min_id, a.min_cov, a.min_ratio, a.gtable, a.n_thread,
                  extPool if extPool is not None else a.process, **kw)
    if a.output:
        rows = data[0] if a.return_overlap else data
        if as_table:
            rows = rows.to_rows(cigar='str')
        text = ''.join('\t'.join(map(str, row)) + '\n' for row in rows)
        if a.output.upper() == 'STDOUT':
            sys.stdout.write(text)
        else:
            with open(a.output, 'w') as fout:
                fout.write(text)
    return data


def uberBlastBatch(references, args, device=None, as_tables=False, strict=False):
    """uberBlast for a LIST of reference files (or in-memory references: dicts / lists of (name, sequence), see _reference_seqs) and one
    query file: `args` are uberBlast's flags without -r/-o.
    Returns one result per reference, each identical to uberBlast(['-r', ref] + args).  One GPU search per tool.
    as_tables: the hit tables come as numeric HitTables (same rows, same order) instead of object rows - for callers like
    mapbsn.build_bsn that work on the columns.  strict: a tool that fails raises instead of being reported and left out (the reference's
    convention, uberBlast.py:347-349, is the default; a caller that writes stores from the result does not want half of it)."""
    a = _parser('uberBlast over many reference files with one search per tool.', False).parse_args(args)
    methods, kw = _run_arguments(a)
    rb = RunBlast(device, sensitive=a.sensitive)
    rb._as_tables = bool(as_tables)
    out = rb.run_batch(references, a.query, methods, a.min_id, a.min_cov, a.min_ratio, a.gtable, a.n_thread, a.process, **kw)
    if strict and rb.failed_tools:
        raise RuntimeError('uberBlastBatch: %s' % '; '.join('%s failed: %s' % ft for ft in rb.failed_tools))
    return out


if __name__ == '__main__':
    uberBlast(sys.argv[1:])
