"""CPU ORACLE - test infrastructure, never imported by the product (peppan_amd/).

ctypes front end of oracle/_build/liboracle.so (align_oracle.c) plus numpy/pure-Python
restatements of the reference's small host-side algorithms that the HIP path moves
to the GPU (translation, query-frame choice, reference chunking, mode-1 rescoring).
Every function cites the reference lines it restates; each is pinned against the golden
vectors in tests/golden (tests/test_oracle_golden.py).

The aligner half (seeds / banded Smith-Waterman / traceback / filters / top-k) is
PARITY UNPINNED: the reference runs it inside the DIAMOND binary, which is not in
/root/reference and cannot be executed here; see align_oracle.c's header.
"""
import ctypes as C
import os
import re
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'liboracle.so')


class Params(C.Structure):
    _fields_ = [('gap_open', C.c_int32), ('gap_ext', C.c_int32), ('n_shapes', C.c_int32), ('base', C.c_int32),
                ('weight', C.c_int32 * 4), ('offs', (C.c_int32 * 32) * 4), ('reduce', C.c_uint8 * 32),
                ('sub', C.c_int8 * 1024), ('min_id_pct', C.c_double), ('min_qcov_pct', C.c_double),
                ('top_k', C.c_int32), ('n_splits', C.c_int32), ('ungapped_min', C.c_int32), ('xdrop', C.c_int32),
                ('ext_right', C.c_int32), ('ext_left', C.c_int32), ('hsp_mode', C.c_int32), ('t_base', C.c_int32), ('stage1_min', C.c_int32), ('pad0', C.c_int32)]


class Hit(C.Structure):
    _fields_ = [('q', C.c_uint32), ('t', C.c_uint32), ('q_start', C.c_uint32), ('q_end', C.c_uint32),
                ('t_start', C.c_uint32), ('t_end', C.c_uint32), ('score', C.c_int32), ('nm', C.c_uint32),
                ('n_ident', C.c_uint32), ('aln_len', C.c_uint32), ('cigar_runs', C.c_uint32), ('bin', C.c_int32),
                ('cigar_off', C.c_uint64), ('cells', C.c_uint64)]


HIT_DTYPE = np.dtype([('q', '<u4'), ('t', '<u4'), ('q_start', '<u4'), ('q_end', '<u4'), ('t_start', '<u4'), ('t_end', '<u4'),
                      ('score', '<i4'), ('nm', '<u4'), ('n_ident', '<u4'), ('aln_len', '<u4'), ('cigar_runs', '<u4'),
                      ('bin', '<i4'), ('cigar_off', '<u8'), ('cells', '<u8')])

_lib = None


def build():
    subprocess.check_call(['make', '-C', HERE, '-s'])


def granted_cpus():
    """CPUs the process may use: the affinity mask cut down to the control group's allowance (a container that shows 256 hardware threads
    and grants 16 CPUs runs an OpenMP team of 256 slower than one of 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_min_score.restype = C.c_int32
        _lib.oracle_min_score.argtypes = [C.c_uint32, C.c_double, C.c_double]
        _lib.oracle_set_subjects.argtypes = [C.c_void_p]
        _lib.oracle_set_subjects.restype = None
        _lib.oracle_set_threads(granted_cpus())
    return _lib


def default_params(min_id_pct=0., min_qcov_pct=0., top_k=10, n_splits=5, ungapped_min=None):
    p = Params()
    lib().oracle_default_params(C.byref(p))
    p.min_id_pct, p.min_qcov_pct, p.top_k, p.n_splits = min_id_pct, min_qcov_pct, top_k, n_splits
    if ungapped_min is not None:
        p.ungapped_min = ungapped_min
    return p


def min_score(qlen, dbsize=5e6, max_evalue=1., ka_lambda=None, ka_k=None):
    if ka_lambda is None:
        return int(lib().oracle_min_score(int(qlen), float(dbsize), float(max_evalue)))
    f = lib().oracle_min_score_ka
    f.restype = C.c_int32
    f.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double]
    return int(f(int(qlen), float(dbsize), float(max_evalue), float(ka_lambda), float(ka_k)))


def params_from(native):
    """oracle parameter block with the same algorithmic fields as a peppan_amd._native.SearchParams"""
    p = Params()
    for f in ('gap_open', 'gap_ext', 'n_shapes', 'base', 'min_id_pct', 'min_qcov_pct', 'top_k', 'n_splits', 'ungapped_min', 'xdrop', 'ext_right', 'ext_left', 'hsp_mode'):
        setattr(p, f, getattr(native, f))
    p.t_base = getattr(native, 't_index_base', 0)
    p.stage1_min = getattr(native, 'stage1_min', 0)
    for i in range(4):
        p.weight[i] = native.weight[i]
        for j in range(32):
            p.offs[i][j] = native.offs[i][j]
    for i in range(32):
        p.reduce[i] = native.reduce[i]
    for i in range(1024):
        p.sub[i] = native.sub[i]
    return p


def aa_codes(s):
    """protein letters -> residue codes (letter - 'A'); '-' and anything else -> X (23)"""
    a = np.frombuffer(s.encode('ascii'), dtype=np.uint8).astype(np.int16) - 65
    a[(a < 0) | (a > 25)] = 23
    return a.astype(np.uint8)


def pack(seqs):
    """list of uint8 arrays -> (concatenated residues, uint64 offsets[n+1])"""
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs])
    res = np.concatenate(seqs).astype(np.uint8) if len(seqs) and off[-1] else np.zeros(0, np.uint8)
    return np.ascontiguousarray(res), off


def search(q_seqs, t_seqs, params=None, min_scores=None, dbsize=5e6, max_evalue=1., subjects=None):
    """q_seqs / t_seqs: lists of uint8 residue-code arrays.  subjects (hsp_mode 2): the sequence every target is a strand / frame of.
    returns (hits structured array, cigar uint32 array (len<<2|op), stats dict)"""
    L = lib()
    p = params or default_params()
    qr, qo = pack(q_seqs)
    tr, to = pack(t_seqs)
    if min_scores is None:
        min_scores = np.array([min_score(len(s), dbsize, max_evalue) for s in q_seqs], dtype=np.int32)
    min_scores = np.ascontiguousarray(min_scores, dtype=np.int32)
    hits_p, cig_p = C.POINTER(Hit)(), C.POINTER(C.c_uint32)()
    nh, ncg = C.c_uint64(), C.c_uint64()
    stats = (C.c_uint64 * 4)()
    if len(qr) == 0:
        qr = np.zeros(1, np.uint8)
    if len(tr) == 0:
        tr = np.zeros(1, np.uint8)
    subj = None if subjects is None else np.ascontiguousarray(subjects, dtype=np.uint32)
    assert subj is None or len(subj) == len(t_seqs)
    L.oracle_set_subjects(None if subj is None else subj.ctypes.data_as(C.c_void_p))
    rc = L.oracle_search(C.byref(p), qr.ctypes.data_as(C.c_void_p), qo.ctypes.data_as(C.c_void_p), C.c_uint32(len(q_seqs)),
                         tr.ctypes.data_as(C.c_void_p), to.ctypes.data_as(C.c_void_p), C.c_uint32(len(t_seqs)),
                         min_scores.ctypes.data_as(C.c_void_p), C.byref(hits_p), C.byref(nh), C.byref(cig_p), C.byref(ncg), stats)
    L.oracle_set_subjects(None)
    assert rc == 0
    n = nh.value
    hits = np.zeros(n, dtype=HIT_DTYPE)
    if n:
        C.memmove(hits.ctypes.data, hits_p, n * C.sizeof(Hit))
    cig = np.zeros(ncg.value, dtype=np.uint32)
    if ncg.value:
        C.memmove(cig.ctypes.data, cig_p, ncg.value * 4)
    L.oracle_free(hits_p)
    L.oracle_free(cig_p)
    return hits, cig, dict(candidates=int(stats[0]), cells=int(stats[1]), pairs=int(stats[2]), tracebacks=int(stats[3]))


def align_one(q, t, bin_, params=None):
    L = lib()
    p = params or default_params()
    h = Hit()
    cig = np.zeros(2 * min(len(q), len(t)) + 4, dtype=np.uint32)
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    L.oracle_align_one(C.byref(p), q.ctypes.data_as(C.c_void_p), C.c_int32(len(q)), t.ctypes.data_as(C.c_void_p), C.c_int32(len(t)),
                       C.c_int32(bin_), C.byref(h), cig.ctypes.data_as(C.c_void_p), C.c_uint32(len(cig)))
    return h, cig[:h.cigar_runs].copy()


def phase_seconds(reset=False):
    """clocks of the searches since the last reset: wall seconds of the seed stage and of the alignment stage; thread-seconds of the score-only
    sweeps and of band_align (traceback sweeps, walks) inside the alignment stage (the protein configuration, hsp_mode 0)"""
    out = np.zeros(4, dtype=np.float64)
    lib().oracle_phase_seconds(out.ctypes.data_as(C.c_void_p), C.c_int(1 if reset else 0))
    return dict(seed_s=float(out[0]), align_s=float(out[1]), score_thread_s=float(out[2]), trace_thread_s=float(out[3]))


def set_rule5a(on):
    """test switch: False = no gapless shortcut, every alignment is traced (align_oracle.c: oracle_set_rule5a)"""
    lib().oracle_set_rule5a(C.c_int(1 if on else 0))


def trace_counts(reset=False):
    """how the reported alignments were obtained since the last reset: (traced, needed the full band, gapless shortcut - rule 5a)"""
    out = np.zeros(3, dtype=np.uint64)
    lib().oracle_trace_counts(out.ctypes.data_as(C.c_void_p), C.c_int(1 if reset else 0))
    return dict(traced=int(out[0]), full_band=int(out[1]), gapless=int(out[2]))


def rescore_counts(q_codes, r_codes, qs, rs, re_, cigar):
    """mode-1 integer counts (nMatch, nMismatch, nGap, bGap, mGap); uberBlast.py:226-249, 412"""
    out = np.zeros(5, dtype=np.int64)
    cigar = np.ascontiguousarray(cigar, np.uint32)
    q_codes = np.ascontiguousarray(q_codes, np.uint8)
    r_codes = np.ascontiguousarray(r_codes, np.uint8)
    lib().oracle_rescore_counts(q_codes.ctypes.data_as(C.c_void_p), r_codes.ctypes.data_as(C.c_void_p), C.c_int64(qs), C.c_int64(rs),
                                C.c_int64(re_), cigar.ctypes.data_as(C.c_void_p), C.c_uint32(len(cigar)), out.ctypes.data_as(C.c_void_p))
    return out


def components(n, a, b):
    a = np.ascontiguousarray(a, np.uint32)
    b = np.ascontiguousarray(b, np.uint32)
    lab = np.zeros(n, dtype=np.uint32)
    lib().oracle_components(C.c_uint32(n), C.c_uint64(len(a)), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p))
    return lab


def sha1_digests(seqs):
    """K13 restatement: the reference's own hash (hashlib.sha1, PEPPAN.py:62, 1019) as uint8[n, 20]"""
    import hashlib
    out = np.zeros((len(seqs), 20), dtype=np.uint8)
    for i, s in enumerate(seqs):
        out[i] = np.frombuffer(hashlib.sha1(s if isinstance(s, (bytes, bytearray)) else s.encode('utf-8')).digest(), dtype=np.uint8)
    return out


def dedup(lengths, digests):
    """K13 restatement of the loop of writeGenes (PEPPAN.py:1026-1038) on genes in priority order: rep[i] = first gene with the same
    digest since the last change of length; the seen-table is rebuilt whenever a different length shows up"""
    digests = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 20)
    rep = np.zeros(len(lengths), dtype=np.uint32)
    seen, open_len = {}, None
    for i, (ln, d) in enumerate(zip(np.asarray(lengths).tolist(), digests)):
        if ln != open_len:
            seen, open_len = {}, ln
        rep[i] = seen.setdefault(d.tobytes(), i)
    return rep


LOCUS_DTYPE = np.dtype([('contig', '<u4'), ('q_start', '<u4'), ('rs', '<u4'), ('re', '<u4'), ('cigar_runs', '<u4'), ('group', '<u4'),
                        ('cigar_off', '<u8')])


def alleles(contigs, rows, cigar, grp_off, grp_qlen, gtable=11):
    """K12 restatement (PEPPAN.py:812-835, 846-848): contigs = list of ASCII byte strings; rows = LOCUS_DTYPE records grouped
    by gene group -> (in_frame int64[n], orf int64[n], packed uint8[sum ceil(qlen/3)])"""
    rows = np.ascontiguousarray(rows, dtype=LOCUS_DTYPE)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    grp_off = np.ascontiguousarray(grp_off, dtype=np.uint64)
    grp_qlen = np.ascontiguousarray(grp_qlen, dtype=np.uint32)
    nt = np.frombuffer(b''.join(contigs), dtype=np.uint8)
    off = np.zeros(len(contigs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(c) for c in contigs])
    in_frame, orf = np.zeros(len(rows), np.int64), np.zeros(len(rows), np.int64)
    packed = np.zeros(int(((grp_qlen.astype(np.int64) + 2) // 3).sum()), dtype=np.uint8)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    rc_ = lib().oracle_alleles(P(nt), P(off), C.c_uint32(len(contigs)), C.c_uint64(len(rows)), P(rows), P(cigar), C.c_uint32(len(grp_qlen)),
                               P(grp_off), P(grp_qlen), C.c_int(gtable), P(in_frame), P(orf), P(packed))
    if rc_ != 0:
        raise ValueError('oracle_alleles: inconsistent rows')
    return in_frame, orf, packed


# --------------------------------------------------------------------------------------------
# numpy / Python restatements of the reference's host-side steps that the HIP path runs on the GPU
# --------------------------------------------------------------------------------------------
_CODON_AA_11 = 'KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVXYXYSSSSXCWCLFLF'   # index a<<4|b<<2|c, A0 C1 G2 T3


def nt_encode_rescore(s):
    """A0 C1 G3 T4 other 2  (uberBlast.py:270-271)"""
    lut = np.full(256, 2, dtype=np.uint8)
    for ch, v in zip('ACGT', (0, 1, 3, 4)):
        lut[ord(ch)] = v
    return lut[np.frombuffer(s.encode('ascii'), dtype=np.uint8)]


def translate_frames(nt, frames, table=11):
    """configure.transeq (configure.py:160-194) for one upper-case nt string.
    frames: iterable of 1..6.  Returns list of protein strings ('X' = stop or ambiguous, '-' = gap codon)."""
    tab = list(_CODON_AA_11)
    if table == 4:
        tab[56] = 'W'          # TGA -> W (configure.py:167-168)
    conv = {'A': 0, 'C': 1, 'G': 2, 'T': 3}
    fw = [conv.get(c, -1 if c != '-' else -2) for c in nt.upper()]
    rv = [(3 - v) if v >= 0 else v for v in reversed(fw)]
    out = []
    for f in frames:
        s = fw[f - 1:] if f <= 3 else rv[f - 4:]
        aa = []
        for k in range(0, len(s), 3):
            cod = s[k:k + 3]
            cod = cod + [-1] * (3 - len(cod))       # partial codon padded with 'ambiguous' (configure.py:186-187)
            if -2 in cod:
                aa.append('-')                      # any '-' -> index 64 (configure.py:190)
            elif -1 in cod:
                aa.append('X')                      # ambiguous -> index 50 (configure.py:191)
            else:
                aa.append(tab[(cod[0] << 4) | (cod[1] << 2) | cod[2]])
        out.append(''.join(aa))
    return out


def query_frame(nt, table=11):
    """frame choice of runDiamond (uberBlast.py:525-529): min over (number of X-separated
    segments of s[:-1], frame index); returns (frame 1..3, protein string)"""
    ss = translate_frames(nt, (1, 2, 3), table)
    best = min((s[:-1].count('X') + 1, i, s) for i, s in enumerate(ss))
    return best[1] + 1, best[2]


def ref_chunks(aa):
    """chunking of a reference frame string (uberBlast.py:539-544): cut after the first 'X'
    at or beyond 1000 residues from the chunk start; returns [(offset, chunk_string)], empty chunks dropped"""
    s = aa + 'X'
    out, c0, n = [], 0, len(s)
    while c0 < n:
        if n - c0 >= 1001:
            x = s.find('X', c0 + 1000)
            out.append((c0, s[c0:x + 1]))
            c0 = x + 1
        else:
            out.append((c0, s[c0:]))
            c0 = n
    off, last = out[-1]
    out[-1] = (off, last[:-1])
    return [(o, c) for o, c in out if len(c)]


def diamond_fasta(query, ref, frames='7', table=11):
    """the qryAA text and the 5 refAA.i texts runDiamond writes (uberBlast.py:525-549)"""
    q_txt = []
    for n in sorted(query):
        f, s = query_frame(query[n], table)
        q_txt.append('>{0}:{1}\n{2}\n'.format(n, f, s))
    fl = (1, 2, 3, 4, 5, 6) if frames == '7' else (1, 2, 3)
    recs = []
    for n in sorted(ref):
        for f, aa in zip(fl, translate_frames(ref[n], fl, table)):
            for off, cs in ref_chunks(aa):
                recs.append('>{0}:{1}:{2}\n{3}\n'.format(n, f, off, cs))
    return ''.join(q_txt), [''.join(recs[i::5]) for i in range(5)]


def parse_diamond_record(qf, rf, rx, pos, cigar_aa, nm, zs, score, ql, rl):
    """parseDiamond coordinate algebra (uberBlast.py:25-58) for one record with forward query frame.
    cigar_aa: list of (n, op) in residues.  Returns the 15-column row minus names, or None if filtered is decided by caller."""
    rs = pos + rx
    cigar = [[3 * n, op] for n, op in cigar_aa]
    qm = sum(n for n, op in cigar_aa if op in 'MI')
    cl = sum(n for n, _ in cigar)
    variation = 3. * nm
    iden = 1 - round(variation / cl, 3)
    rm = sum(n for n, op in cigar_aa if op in 'MD')
    if rf <= 3:
        rs_nt, re_nt = rs * 3 + rf - 3, (rs + rm - 1) * 3 + rf - 1
    else:
        rs_nt, re_nt = rl - (rs * 3 + rf - 6) + 1, rl - ((rs + rm - 1) * 3 + rf - 4) + 1
    qs_nt, qe_nt = zs * 3 + qf - 3, (zs + qm - 1) * 3 + qf - 1
    gaps = [n for n, op in cigar if op != 'M']
    return dict(iden=iden, cl=cl, mismatch=int(variation - sum(gaps)), gapopen=len(gaps), qs=qs_nt, qe=qe_nt, rs=rs_nt, re=re_nt,
                score=score, qm=qm, cigar=cigar)


def nt_codes(s):
    """A0 C1 G2 T3, anything else 4 (the clustering alphabet)"""
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in zip('ACGTacgt', (0, 1, 2, 3, 0, 1, 2, 3)):
        lut[ord(ch)] = v
    return lut[np.frombuffer(s.encode('ascii') if isinstance(s, str) else s, dtype=np.uint8)]


def linclust(seqs, min_id, min_cov, base=4, k=17, m=20):
    """seqs: list of uint8 code arrays -> (rep index per sequence, stats)"""
    res, off = pack(seqs)
    if len(res) == 0:
        res = np.zeros(1, np.uint8)
    rep = np.zeros(len(seqs), dtype=np.uint32)
    stats = (C.c_uint64 * 3)()
    lib().oracle_linclust(res.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), C.c_uint32(len(seqs)), C.c_int(base), C.c_int(k), C.c_int(m),
                          C.c_double(min_id), C.c_double(min_cov), rep.ctypes.data_as(C.c_void_p), stats)
    return rep, dict(selected=int(stats[0]), verified=int(stats[1]), accepted=int(stats[2]))


def overlaps_sweep(contig, start, end, row_id, ovl_l, ovl_p):
    """interval sweep of tab2overlaps (uberBlast.py:73-97) without its 1e6 batching: rows sorted by (contig, start, end);
    returns int64[m, 3] (id1, id2, overlap) in (i, j) order"""
    out = []
    n = len(contig)
    for i in range(n):
        need = min(ovl_l, ovl_p * (end[i] - start[i] + 1))
        for j in range(i + 1, n):
            if contig[j] != contig[i] or start[j] > end[i]:
                break
            ovl = min(end[i], end[j]) - start[j] + 1
            if ovl >= need or ovl >= ovl_p * (end[j] - start[j] + 1):
                out.append((row_id[i], row_id[j], ovl))
    return np.array(out, dtype=np.int64).reshape(-1, 3)


def pair_support(rows, q_len, r_len, lim):
    """get_similar of PEPPAN.py:195-224 for the forward alignments of ONE (query, reference) pair, restated with the reference's own
    dictionary (query nucleotide position -> identity of the last alignment that covered it; np.mean over the values in insertion order).
    rows: [(q_start, r_start, identity, [(n, op), ...])] in table order, op 0 = M, 1 = I, 2 = D (nucleotides);
    lim: dict(match_len=[3], match_prop=[3], identity_x1e4, any_frame).  -> None (no decision) | 0 | int(mean identity * 10000)"""
    if min(q_len, r_len) * 20 <= max(q_len, r_len):                                  # PEPPAN.py:199-200
        return None
    matched = {}
    for q_start, r_start, ident, runs in rows:
        s_i, s_j = int(q_start), int(r_start)
        for n, op in runs:
            n = int(n)
            if op == 0:
                frame_i, frame_j = s_i % 3, s_j % 3
                if frame_i == frame_j or lim['any_frame']:                           # PEPPAN.py:205-206
                    matched.update({(s_i + x): ident for x in range((3 - (frame_i - 1)) % 3, n)})
                s_i += n
                s_j += n
                if len(matched) * 3 >= min(lim['match_len']) and len(matched) * 3 >= min(lim['match_prop']) * q_len:      # PEPPAN.py:211
                    ave = int(np.mean(list(matched.values())) * 10000)
                    if ave >= lim['identity_x1e4']:
                        shorter = min(q_len, r_len)
                        need = min(max(l, p * shorter) for l, p in zip(lim['match_len'], lim['match_prop']))
                        return ave if len(matched) * 3 >= need else 0                # PEPPAN.py:214-219
            elif op == 1:
                s_i += n
            else:
                s_j += n
    return None
