"""ctypes binding of libpeppan_hip.so (include/peppan_hip.h).  No torch, no CPU fallback:
if the HIP library is missing or no MI355X is visible every entry point raises."""
import ctypes as C
import os
import threading
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PEPPAN_HIP_LIB') or os.path.join(_HERE, 'libpeppan_hip.so')      # (PEPPAN_HIP_LIB: a measurement build of the library, tools/ab only)

ABI_VERSION = 16
MAX_SEQ_LEN = (1 << 23) - 256          # PEP_MAX_SEQ_LEN: longest single sequence of a packed set
EXPORTS = ['pep_version', 'pep_device_count', 'pep_ctx_create', 'pep_ctx_destroy', 'pep_last_error', 'pep_default_params', 'pep_set_sensitivity',
           'pep_min_score', 'pep_min_score_ka', 'pep_set_query_nt', 'pep_set_ref_nt', 'pep_set_query_aa', 'pep_set_ref_aa', 'pep_translate', 'pep_use_nt_as_residues',
           'pep_query_count', 'pep_target_count', 'pep_get_query_meta', 'pep_get_target_meta', 'pep_get_query_aa',
           'pep_get_target_aa', 'pep_set_target_groups', 'pep_set_result_mode', 'pep_set_timing', 'pep_set_grouping', 'pep_result_labels', 'pep_invalidate_translation', 'pep_search', 'pep_result_size', 'pep_result_copy', 'pep_result_data', 'pep_result_device', 'pep_result_stats', 'pep_components_of_result', 'pep_result_free',
           'pep_merge_hits', 'pep_rescore_nt', 'pep_components', 'pep_components_of_hits', 'pep_linclust', 'pep_overlaps', 'pep_alleles', 'pep_ovl_filter', 'pep_known_order', 'pep_linear_merge', 'pep_sha1', 'pep_dedup',
           'pep_similar_classify', 'pep_similar_scan', 'pep_pair_support', 'pep_similar_resolve', 'pep_fasta_keep', 'pep_fasta_scan', 'pep_fasta_records', 'pep_store_mat_member', 'pep_store_seq_member', 'pep_store_tab_members', 'pep_store_tab_archive', 'pep_deflate_literals', 'pep_deflate_fast', 'pep_crc32', 'pep_pack_member', 'pep_argsort_object_order',
           'pep_set_nt_match', 'pep_result_nt_match', 'pep_table_from_hits', 'pep_cols_fix_end', 'pep_cols_order', 'pep_cols_gather', 'pep_lex_order', 'pep_set_host_threads']


class PepError(RuntimeError):
    pass


class SearchParams(C.Structure):
    _fields_ = [('gap_open', C.c_int32), ('gap_ext', C.c_int32), ('n_shapes', C.c_int32), ('base', C.c_int32),
                ('weight', C.c_int32 * 4), ('offs', (C.c_int32 * 32) * 4), ('reduce', C.c_uint8 * 32),
                ('sub', C.c_int8 * 1024), ('min_id_pct', C.c_double), ('min_qcov_pct', C.c_double),
                ('top_k', C.c_int32), ('n_splits', C.c_int32), ('dbsize', C.c_double), ('max_evalue', C.c_double),
                ('use_lds', C.c_int32), ('ungapped_min', C.c_int32), ('xdrop', C.c_int32), ('ext_right', C.c_int32),
                ('ext_left', C.c_int32), ('reserved', C.c_int32 * 3), ('ka_lambda', C.c_double), ('ka_k', C.c_double), ('hsp_mode', C.c_int32), ('t_index_base', C.c_int32), ('stage1_min', C.c_int32), ('reserved2', C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ('query_residues', 'target_residues', 'query_seeds', 'target_seeds', 'seed_hits',
                                           'seed_hits_passed', 'candidates', 'pairs', 'tracebacks', 'hits', 'cells', 'cells_swept', 'dir_bytes',
                                           'sw_launches', 'cells_trace', 'cells_swept_trace', 'tracebacks_gapless', 'candidates_settled', 'cells_settled')] + \
               [(n, C.c_double) for n in ('ms_seed', 'ms_sw', 'ms_trace', 'ms_total', 'ms_k1', 'ms_sw_trace', 'ms_seed_match', 'ms_reserved0', 'ms_reserved1', 'ms_reserved2')]


HIT_DTYPE = np.dtype([('q', '<u4'), ('t', '<u4'), ('q_start', '<u4'), ('q_end', '<u4'), ('t_start', '<u4'), ('t_end', '<u4'),
                      ('score', '<i4'), ('nm', '<u4'), ('n_ident', '<u4'), ('aln_len', '<u4'), ('cigar_runs', '<u4'),
                      ('bin', '<i4'), ('cigar_off', '<u8'), ('cells', '<u8')])
QUERY_META_DTYPE = np.dtype([('seq', '<u4'), ('frame', '<u4'), ('aa_len', '<u4'), ('nt_len', '<u4')])
TARGET_META_DTYPE = np.dtype([('seq', '<u4'), ('frame', '<u4'), ('chunk_off', '<u4'), ('aa_len', '<u4')])
NT_HIT_DTYPE = np.dtype([('q', '<u4'), ('r', '<u4'), ('qs', '<u4'), ('qe', '<u4'), ('rs', '<u4'), ('re', '<u4'),
                         ('cigar_runs', '<u4'), ('pad', '<u4'), ('cigar_off', '<u8')])

LOCUS_DTYPE = np.dtype([('contig', '<u4'), ('q_start', '<u4'), ('rs', '<u4'), ('re', '<u4'), ('cigar_runs', '<u4'), ('group', '<u4'),
                        ('cigar_off', '<u8')])

SUPPORT_ROW_DTYPE = np.dtype([('q_start', '<u4'), ('r_start', '<u4'), ('cigar_runs', '<u4'), ('pad', '<u4'), ('cigar_off', '<u8'), ('identity', '<f8')])
SUPPORT_NONE = -2 ** 31
ROW_ORDINARY, ROW_CONFLICT, ROW_ABSORB_QUERY, ROW_ABSORB_REF = 0, 1, 2, 3
EVENT_CONFLICT, EVENT_SUPPORT = 0, 1


class SupportLimits(C.Structure):
    _fields_ = [('match_len', C.c_double * 3), ('match_prop', C.c_double * 3), ('identity_x1e4', C.c_double), ('any_frame', C.c_int32), ('pad', C.c_int32)]


def support_limits(params):
    """the thresholds of get_similar (PEPPAN.py:205-216) from PEPPAN's parameter dictionary"""
    lim = SupportLimits()
    for k, (l, p) in enumerate((('match_len', 'match_prop'), ('match_len1', 'match_prop1'), ('match_len2', 'match_prop2'))):
        lim.match_len[k], lim.match_prop[k] = float(params[l]), float(params[p])
    lim.identity_x1e4 = params['match_identity'] * 10000
    lim.any_frame = 1 if 'f' in params['incompleteCDS'] else 0
    return lim


_lib = None


def _tune_malloc():
    """glibc hands every block of 128 KiB and more straight to mmap and back: the host chain's columns - a few MB per table, a dozen tables per call - arrive as fresh
    zero pages every time, and the page faults are a sixth of the hot call's wall time (17.7 -> 14.9 ms with object rows, 30.5 -> 28.4 ms for get_similar_pairs on
    one box).  Blocks up to 32 MiB (the largest threshold glibc accepts) are therefore kept in the heap and reused; at most 1 GiB of free heap top is held back.
    A process-wide setting, made when the library is first loaded; PEPPAN_MALLOC_TUNE=0 leaves the allocator alone."""
    if os.environ.get('PEPPAN_MALLOC_TUNE', '1') == '0':
        return
    try:
        libc = C.CDLL(None)
        libc.mallopt.argtypes, libc.mallopt.restype = [C.c_int, C.c_int], C.c_int
        libc.mallopt(-3, 32 << 20)               # M_MMAP_THRESHOLD
        libc.mallopt(-1, 1 << 30)                # M_TRIM_THRESHOLD
    except (OSError, AttributeError):
        pass                                     # (not glibc: nothing to tune)


def load_library():
    """dlopen the in-tree library; raises PepError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _tune_malloc()
    if not os.path.exists(LIB_PATH):
        raise PepError('libpeppan_hip.so is not built (run `python -c "import __graft_entry__ as g; g.build()"` '
                       'or `make -C peppan_amd/csrc`); there is no CPU fallback')
    if 'PEPPAN_HOST_THREADS' not in os.environ:
        # threads a pass of the host chain (csrc/hostchain.hip) may use on a large table: a quarter of the CPUs the container grants, four at most
        from .configure import effective_cpus
        os.environ['PEPPAN_HOST_THREADS'] = str(max(1, min(4, effective_cpus() // 4)))
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise PepError('libpeppan_hip.so does not export ' + name)
    lib.pep_last_error.restype = C.c_char_p
    lib.pep_last_error.argtypes = [C.c_void_p]
    lib.pep_min_score.restype = C.c_int32
    lib.pep_min_score.argtypes = [C.c_uint32, C.c_double, C.c_double]
    lib.pep_min_score_ka.restype = C.c_int32
    lib.pep_min_score_ka.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.pep_ctx_destroy.argtypes = [C.c_void_p]
    lib.pep_ctx_destroy.restype = None
    lib.pep_result_free.argtypes = [C.c_void_p]
    lib.pep_result_free.restype = None
    if lib.pep_version() != ABI_VERSION:
        raise PepError('libpeppan_hip.so ABI version mismatch')
    _lib = lib
    return lib


DEFAULT_SHAPES = ('111101110111', '111011010010111')          # DIAMOND's two default-sensitivity shapes (weight 10)
SENSITIVE_SHAPES = ('110010011111011', '10111110011011')      # two more weight-10 shapes for the sensitive mode (see default_params)


def set_shapes(p, shapes):
    """seed shapes as '1101...' strings -> the weight / offsets fields of a parameter block (native or oracle)"""
    p.n_shapes = len(shapes)
    for s, sh in enumerate(shapes):
        ones = [k for k, ch in enumerate(sh) if ch == '1']
        p.weight[s] = len(ones)
        for i in range(32):
            p.offs[s][i] = ones[i] if i < len(ones) else 0
    for s in range(len(shapes), 4):
        p.weight[s] = 0


def default_params(min_id_pct=0., min_qcov_pct=0., top_k=10, n_splits=5, dbsize=5e6, max_evalue=1., use_lds=1, ungapped_min=None, sensitive=False):
    """the protein search configured like the reference's diamond call (uberBlast.py:550).  sensitive: four seed shapes instead of DIAMOND's
    two default-mode shapes - against exhaustive Smith-Waterman the recall between 0.45 and 0.7 identity rises from 0.93 to 0.985 on the
    1 000-gene configuration (above 0.7 it is 1.0 either way) at twice the seed-stage cost; the reference itself runs diamond at its default
    sensitivity, so this is an option (RunBlast(sensitive=True) / uberBlast's --sensitive flag / pep_set_sensitivity in the C ABI), not the default"""
    p = SearchParams()
    load_library().pep_default_params(C.byref(p))
    if ungapped_min is not None:
        p.ungapped_min = int(ungapped_min)
    p.min_id_pct, p.min_qcov_pct, p.top_k, p.n_splits = float(min_id_pct), float(min_qcov_pct), int(top_k), int(n_splits)
    p.dbsize, p.max_evalue, p.use_lds = float(dbsize), float(max_evalue), int(use_lds)
    if sensitive:
        if load_library().pep_set_sensitivity(C.byref(p), 1) != 0:
            raise PepError('pep_set_sensitivity failed')
    return p


def min_score(qlen, dbsize=5e6, max_evalue=1.):
    return int(load_library().pep_min_score(int(qlen), float(dbsize), float(max_evalue)))


_NUCL_PARAMS = {}


def nucleotide_params(min_id_pct=0., min_qcov_pct=0., top_k=1000, dbsize=5e6, max_evalue=1e-2, hsp_mode=1):
    """the search engine configured like the reference's blastn call (uberBlast.py:294): residues A0 C1 G2 T3 (other 4),
    exact 17-mers (-word_size 17), reward 2 / penalty -3, gap 6 + 2k, e-value 1e-2 at dbsize 5e6, 1000 targets per query.
    Karlin-Altschul lambda 0.625 / K 0.41 are NCBI's published values for 2/-3 with gap costs 5/2 (closest tabulated)."""
    if hsp_mode not in (1, 2):
        raise ValueError('nucleotide_params: hsp_mode 1 (every band that reaches the threshold) or 2 (BLAST-like culling, top_k counts subjects)')
    key = (float(min_id_pct), float(min_qcov_pct), int(top_k), float(dbsize), float(max_evalue), int(hsp_mode))
    made = _NUCL_PARAMS.get(key)
    if made is not None:                 # (the block is 1.4 KB of fields set one by one below - half a millisecond of every nucleotide search: a copy of the first one)
        return SearchParams.from_buffer_copy(made)
    p = default_params(min_id_pct, min_qcov_pct, top_k, 1, dbsize, max_evalue)
    p.gap_open, p.gap_ext = 6, 2
    p.n_shapes, p.base = 1, 4
    for s in range(4):
        p.weight[s] = 0
    p.weight[0] = 17
    for i in range(32):
        p.offs[0][i] = i if i < 17 else 0
        p.reduce[i] = i if i < 4 else 0xFF
    for a in range(32):
        for b in range(32):
            p.sub[a * 32 + b] = (2 if a == b else -3) if (a < 4 and b < 4) else (-3 if (a < 5 and b < 5) else -64)
    p.ungapped_min, p.xdrop, p.ext_right, p.ext_left = 40, 16, 40, 24
    p.stage1_min = 0                     # (an exact 17-mer scores 32 over its first 16 bases: the first stage has nothing to reject here)
    p.ka_lambda, p.ka_k = 0.625, 0.41
    p.hsp_mode = hsp_mode                # blastn reports every HSP of a subject; a contig can carry several copies of a gene.  2: include/peppan_hip.h
    if len(_NUCL_PARAMS) < 64:
        _NUCL_PARAMS[key] = bytes(p)
    return p


def _pack(seqs):
    """list of str / bytes / uint8 arrays -> (uint8 concatenation, uint64 offsets[n+1]); a ready-made (codes, offsets) tuple passes through"""
    if isinstance(seqs, tuple) and len(seqs) == 2:
        res, off = seqs
        res = np.ascontiguousarray(res, dtype=np.uint8)
        return (res if res.size else np.zeros(1, dtype=np.uint8)), np.ascontiguousarray(off, dtype=np.uint64)
    n = len(seqs)
    off = np.zeros(n + 1, dtype=np.uint64)
    if n >= 4096 and isinstance(seqs, list):                         # big lists of str / bytes: lengths and ONE copy of the bytes in two C loops
        from .hittable import _pyrows
        lens = np.zeros(n, dtype=np.int64)
        total = _pyrows().pep_strs_measure(seqs, lens.ctypes.data)
        if total >= 0:
            res = np.empty(max(total, 1), dtype=np.uint8)
            if _pyrows().pep_strs_pack(seqs, res.ctypes.data, total) == 0:
                off[1:] = np.cumsum(lens)
                return res, off
    if n and all(isinstance(s, str) for s in seqs):                  # one join + one buffer view instead of one array per sequence
        off[1:] = np.cumsum(np.fromiter(map(len, seqs), dtype=np.int64, count=n))
        res = np.frombuffer(''.join(seqs).encode('ascii'), dtype=np.uint8)
    elif n and all(isinstance(s, (bytes, bytearray)) for s in seqs):
        off[1:] = np.cumsum(np.fromiter(map(len, seqs), dtype=np.int64, count=n))
        res = np.frombuffer(b''.join(seqs), dtype=np.uint8)
    else:
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else
                (np.frombuffer(s.encode('ascii'), dtype=np.uint8) if isinstance(s, str) else np.asarray(s, dtype=np.uint8)) for s in seqs]
        if arrs:
            off[1:] = np.cumsum([a.size for a in arrs])
        res = np.concatenate(arrs) if arrs and off[-1] else np.zeros(1, dtype=np.uint8)
    if res.size == 0:
        res = np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(res, dtype=np.uint8), off


def _count(seqs):
    return len(seqs[1]) - 1 if isinstance(seqs, tuple) and len(seqs) == 2 else len(seqs)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def ovl_filter(q, r, qs, qe, ss, se, score, iden, coverage, delta):
    """pep_ovl_filter on sorted numeric columns; `iden` (float64) is updated in place: dropped rows get -1"""
    lib = load_library()
    n = len(q)
    arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (q, r, qs, qe, ss, se)]
    score = np.ascontiguousarray(score, dtype=np.float64)
    assert iden.dtype == np.float64 and iden.flags['C_CONTIGUOUS']
    rc_ = lib.pep_ovl_filter(C.c_uint64(n), *[_ptr(a) for a in arrs], _ptr(score), _ptr(iden), C.c_double(coverage), C.c_double(delta))
    if rc_ != 0:
        raise PepError('pep_ovl_filter failed (%d)' % rc_)


def known_order(T, genes_of):
    """pep_known_order: compare_prediction over a HitTable's columns -> (order int64[n], known float64[n]) - the rows in the order (query, contig, score) and their
    column 10.  genes_of(c): the original genes of the contig with row code c as (start int64[], end int64[], plus bool[]) in the store's order, or None"""
    lib = load_library()
    n = len(T)
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    ri, r_code, q_code = i64(T.ri), i64(T.r_codes()), i64(T.q_codes())
    n_contigs = len(T.r_tab)
    g_off = np.zeros(n_contigs + 1, dtype=np.uint64)
    parts, is_sorted = {}, np.ones(max(n_contigs, 1), dtype=np.uint8)
    for c in (np.unique(ri).tolist() if n else []):
        g = genes_of(c)
        if g is not None and len(g[0]):
            parts[c] = g
            g_off[c + 1] = len(g[0])
            is_sorted[c] = 0 if (np.diff(g[0]) < 0).any() else 1
    np.cumsum(g_off, out=g_off)
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([parts[c][k] for c in sorted(parts)]) if parts else np.zeros(1), dtype=dt)
    g1, g2, plus = cat(0, np.int64), cat(1, np.int64), cat(2, np.uint8)
    cols = [i64(a) for a in (T.ss, T.se, T.qs, T.qe, T.ql)]
    score = np.ascontiguousarray(T.score, dtype=np.float64)
    order, known = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.float64)
    rc_ = lib.pep_known_order(C.c_uint64(n), _ptr(ri), _ptr(r_code), _ptr(q_code), *([_ptr(c) for c in cols] + [_ptr(score), C.c_uint64(n_contigs), _ptr(g_off), _ptr(g1), _ptr(g2),
                              _ptr(plus), _ptr(is_sorted), _ptr(order), _ptr(known)]))
    if rc_ != 0:
        raise PepError('pep_known_order failed (%d)' % rc_)
    return order, known


def linear_merge(q, r, iden, qs, qe, ss, se, score, ql, sl, rid, gap_dist, len_diff):
    """pep_linear_merge on sorted numeric columns -> (keep_seq, query_off, query_ascending, grp_score, grp_iden, grp_span, grp_ids_off, grp_ids)"""
    lib = load_library()
    n = len(q)
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    q, r, qs, qe, ss, se, ql, sl, rid = (i64(a) for a in (q, r, qs, qe, ss, se, ql, sl, rid))
    iden, score = f64(iden), f64(score)
    keep_cap, ids_cap = 2 * n + 16, 2 * n + 16
    for _ in range(2):
        keep = np.zeros(keep_cap, dtype=np.int64)
        q_off = np.zeros(n + 2, dtype=np.uint64)
        asc = np.zeros(n + 1, dtype=np.uint8)
        g_score, g_iden, g_span = np.zeros(n + 1), np.zeros(n + 1), np.zeros(n + 1, dtype=np.int64)
        ids_off = np.zeros(n + 2, dtype=np.uint64)
        ids = np.zeros(ids_cap, dtype=np.int64)
        n_keep, n_query, n_ids = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc_ = lib.pep_linear_merge(C.c_uint64(n), _ptr(q), _ptr(r), _ptr(iden), _ptr(qs), _ptr(qe), _ptr(ss), _ptr(se), _ptr(score), _ptr(ql), _ptr(sl),
                                   _ptr(rid), C.c_double(gap_dist), C.c_double(len_diff), _ptr(keep), C.c_uint64(keep_cap), C.byref(n_keep), _ptr(q_off),
                                   _ptr(asc), C.byref(n_query), _ptr(g_score), _ptr(g_iden), _ptr(g_span), _ptr(ids_off), _ptr(ids), C.c_uint64(ids_cap),
                                   C.byref(n_ids))
        if rc_ != 0:
            raise PepError('pep_linear_merge failed (%d)' % rc_)
        if n_keep.value <= keep_cap and n_ids.value <= ids_cap:
            nq = n_query.value
            return (keep[:n_keep.value], q_off[:nq + 1].astype(np.int64), asc[:nq], g_score[:n], g_iden[:n], g_span[:n], ids_off[:n + 1].astype(np.int64),
                    ids[:n_ids.value])
        keep_cap, ids_cap = n_keep.value + 16, n_ids.value + 16
    raise PepError('pep_linear_merge: inconsistent sizes')


class MatCols(C.Structure):
    """pep_mat_cols (include/peppan_hip.h): the sixteen stored columns of the hit table as pointers"""
    _fields_ = [(n, C.c_void_p) for n in ('q', 'r', 'iden', 'aln', 'mis', 'gap', 'qs', 'qe', 'ss', 'se', 'evalue', 'score', 'ql', 'sl', 'arena', 'c_off', 'c_runs', 'rid')] + \
               [('score_is_int', C.c_int32), ('reserved', C.c_int32)]


_RECON_MODULE = None


def _recon_module():
    """the module numpy's own pickles name for `_reconstruct` (numpy._core.multiarray since numpy 2, numpy.core.multiarray before)"""
    global _RECON_MODULE
    if _RECON_MODULE is None:
        _RECON_MODULE = np.empty(0).__reduce__()[0].__module__.encode()
    return _RECON_MODULE


def _npy_object_header(n):
    """the .npy header of a 1-D object array of n elements (what np.lib.format.write_array puts in front of the pickle)"""
    import io
    buf = io.BytesIO()
    np.lib.format.write_array_header_1_0(buf, {'descr': '|O', 'fortran_order': False, 'shape': (int(n),)})
    return buf.getvalue()


def store_mat_member(cols, row_off, score_is_int):
    """pep_store_mat_member: the complete .npy member (header + pickle stream) of one chunk of the .mat store (PEPPAN.py:959-966).
    cols: the 18 arrays of MatCols in field order (q and r as int64 names per row); row_off int64[n_groups + 1]."""
    lib = load_library()
    lib.pep_store_mat_member.restype = C.c_int64
    keep = [np.ascontiguousarray(a, dtype=dt) for a, dt in zip(cols, (np.int64, np.int64, np.float64) + (np.int64,) * 7 + (np.float64, np.float64, np.int64, np.int64,
                                                                                                                       np.uint32, np.int64, np.int64, np.int64))]
    mc = MatCols(*[a.ctypes.data for a in keep], int(bool(score_is_int)), 0)
    row_off = np.ascontiguousarray(row_off, dtype=np.int64)
    n_groups = len(row_off) - 1
    n_rows = int(row_off[-1] - row_off[0]) if n_groups else 0
    head = _npy_object_header(n_groups)
    cap = 256 + 64 * n_groups + 200 * n_rows + 12 * int(keep[16][row_off[0]:row_off[-1]].sum() if n_rows else 0)
    for _ in range(2):
        buf = np.empty(len(head) + cap, dtype=np.uint8)
        need = lib.pep_store_mat_member(C.byref(mc), _ptr(row_off), C.c_int64(n_groups), C.c_char_p(_recon_module()), C.c_void_p(buf.ctypes.data + len(head)), C.c_int64(cap))
        if need < 0:
            raise PepError('pep_store_mat_member failed (%d)' % need)
        if need <= cap:
            buf[:len(head)] = np.frombuffer(head, dtype=np.uint8)
            return buf[:len(head) + need].tobytes()
        cap = int(need)
    raise PepError('pep_store_mat_member: inconsistent sizes')


def store_seq_member(packed, pack_off):
    """pep_store_seq_member: the complete .npy member of one chunk of the .seq store (PEPPAN.py:950-957): object array of uint8 arrays"""
    lib = load_library()
    lib.pep_store_seq_member.restype = C.c_int64
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    pack_off = np.ascontiguousarray(pack_off, dtype=np.int64)
    n_groups = len(pack_off) - 1
    head = _npy_object_header(n_groups)
    cap = 256 + 64 * n_groups + (int(pack_off[-1] - pack_off[0]) if n_groups else 0)
    buf = np.empty(len(head) + cap, dtype=np.uint8)
    need = lib.pep_store_seq_member(_ptr(packed), _ptr(pack_off), C.c_int64(n_groups), C.c_char_p(_recon_module()), C.c_void_p(buf.ctypes.data + len(head)), C.c_int64(cap))
    if need < 0 or need > cap:
        raise PepError('pep_store_seq_member failed (%d)' % need)
    buf[:len(head)] = np.frombuffer(head, dtype=np.uint8)
    return buf[:len(head) + need].tobytes()


def deflate_literals(data):
    """pep_deflate_literals: bytes -> raw DEFLATE stream (zlib.decompress(x, -15) gives them back), Huffman coding only"""
    lib = load_library()
    lib.pep_deflate_literals.restype = C.c_int64
    src = np.frombuffer(data, dtype=np.uint8)
    cap = len(src) + len(src) // 64 + 512
    out = np.empty(cap, dtype=np.uint8)
    n = lib.pep_deflate_literals(_ptr(src) if len(src) else None, C.c_int64(len(src)), _ptr(out), C.c_int64(cap))
    if n < 0 or n > cap:
        raise PepError('pep_deflate_literals failed (%d)' % n)
    return out[:n].tobytes()


def crc32(data, crc=0):
    """pep_crc32: zlib.crc32 of a bytes-like object, by carry-less multiplication where the CPU has it"""
    lib = load_library()
    lib.pep_crc32.restype = C.c_uint32
    src = np.frombuffer(data, dtype=np.uint8)
    return int(lib.pep_crc32(_ptr(src) if len(src) else None, C.c_int64(len(src)), C.c_uint32(crc)))


def pack_member(data, coder):
    """pep_pack_member: bytes -> (raw DEFLATE stream, crc32 of the bytes); coder 0 = literals only (deflate_literals), 1 = single-probe matcher (deflate_fast)"""
    lib = load_library()
    lib.pep_pack_member.restype = C.c_int64
    src = np.frombuffer(data, dtype=np.uint8)
    cap = len(src) + len(src) // 8 + 1024
    out = np.empty(cap, dtype=np.uint8)
    crc = C.c_uint32(0)
    n = lib.pep_pack_member(_ptr(src) if len(src) else None, C.c_int64(len(src)), C.c_int32(coder), _ptr(out), C.c_int64(cap), C.byref(crc))
    if n < 0 or n > cap:
        raise PepError('pep_pack_member failed (%d)' % n)
    return out[:n].tobytes(), int(crc.value)


def deflate_fast(data):
    """pep_deflate_fast: bytes -> raw DEFLATE stream (zlib.decompress(x, -15) gives them back): single-probe matcher + dynamic Huffman blocks"""
    lib = load_library()
    lib.pep_deflate_fast.restype = C.c_int64
    src = np.frombuffer(data, dtype=np.uint8)
    cap = len(src) + len(src) // 8 + 1024
    out = np.empty(cap, dtype=np.uint8)
    n = lib.pep_deflate_fast(_ptr(src) if len(src) else None, C.c_int64(len(src)), _ptr(out), C.c_int64(cap))
    if n < 0 or n > cap:
        raise PepError('pep_deflate_fast failed (%d)' % n)
    return out[:n].tobytes()


class HitCols(C.Structure):
    """pep_hit_cols: pointers to the columns of a hit table"""
    FIELDS = ('qi', 'ri', 'iden', 'aln', 'mis', 'gap', 'qs', 'qe', 'ss', 'se', 'evalue', 'score', 'ql', 'sl', 'c_off', 'c_runs', 'rid')
    FLOATS = ('iden', 'evalue', 'score')
    _fields_ = [(f, C.c_void_p) for f in FIELDS]

    @classmethod
    def over(cls, arrays):
        """the struct over a dict of contiguous int64 / float64 arrays (field -> array; a missing 'rid' is NULL)"""
        c = cls()
        for f in cls.FIELDS:
            a = arrays.get(f)
            if a is not None:
                assert a.flags['C_CONTIGUOUS'] and a.dtype == (np.float64 if f in cls.FLOATS else np.int64), f
                setattr(c, f, a.ctypes.data)
        return c

    @classmethod
    def blank(cls, n):
        """n uninitialised rows: field -> array"""
        block = np.empty([len(cls.FIELDS), max(n, 1)], dtype=np.int64)
        return {f: (block[k].view(np.float64) if f in cls.FLOATS else block[k]) for k, f in enumerate(cls.FIELDS)}


def table_from_hits(tool, hits, cigar, q_len, r_len, min_id, min_cov, min_ratio, q_meta=None, t_meta=None, t_seq=None, t_rev=None, windows=None, evalue=None, nt_match=None):
    """pep_table_from_hits: hit records -> ({field: column[m]}, CIGAR arena in nucleotides).  tool 0 = translated search (q_meta / t_meta), tool 1 =
    nucleotide search (t_seq / t_rev [, windows = (offset, home_lo, home_hi) per target], evalue per hit).  nt_match (uint32 per hit, Context.last_nt_match):
    identity and score come out rescored (reScore mode 1)"""
    lib = load_library()
    lib.pep_table_from_hits.restype = C.c_int64
    n = len(hits)
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    arena = np.empty(max(len(cigar), 1), dtype=np.uint32)
    cols = HitCols.blank(n)
    i64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int64)
    q_len, r_len, t_seq = i64(q_len), i64(r_len), i64(t_seq)
    t_rev = None if t_rev is None else np.ascontiguousarray(t_rev, dtype=np.uint8)
    w = [i64(a) for a in windows] if windows is not None else [None, None, None]
    evalue = None if evalue is None else np.ascontiguousarray(evalue, dtype=np.float64)
    qm = None if q_meta is None else np.ascontiguousarray(q_meta, dtype=QUERY_META_DTYPE)
    tm = None if t_meta is None else np.ascontiguousarray(t_meta, dtype=TARGET_META_DTYPE)
    if nt_match is not None:
        nt_match = np.ascontiguousarray(nt_match, dtype=np.uint32)
        if len(nt_match) != n:
            raise ValueError('table_from_hits: one nt_match count per hit')
    p = lambda a: None if a is None else _ptr(a)
    hc = HitCols.over(cols)
    m = lib.pep_table_from_hits(C.c_int32(tool), C.c_uint64(n), p(hits), p(cigar), C.c_uint64(len(cigar)), p(qm), p(tm), p(q_len), p(r_len), p(t_seq), p(t_rev),
                                p(w[0]), p(w[1]), p(w[2]), p(evalue), C.c_double(min_id), C.c_double(min_cov), C.c_double(min_ratio), C.byref(hc), _ptr(arena), p(nt_match) if n else None)
    if m < 0:
        raise PepError('pep_table_from_hits failed (%d)' % m)
    return {f: a[:m] for f, a in cols.items()}, arena[:len(cigar)]


def cols_fix_end(cols, arena, se_lim, ee_lim):
    """pep_cols_fix_end over {field: column} in place -> the rows' private arena (c_off rewritten)"""
    lib = load_library()
    lib.pep_cols_fix_end.restype = C.c_int64
    n = len(cols['qs'])
    out = np.empty(max(int(cols['c_runs'].sum()) if n else 0, 1), dtype=np.uint32)
    hc = HitCols.over(cols)
    arena = np.ascontiguousarray(arena, dtype=np.uint32)
    rc_ = lib.pep_cols_fix_end(C.c_uint64(n), C.byref(hc), _ptr(arena) if len(arena) else None, C.c_uint64(len(arena)), _ptr(out), C.c_double(se_lim), C.c_double(ee_lim))
    if rc_ < 0:
        raise IndexError('fix_end: a row without CIGAR runs cannot be extended (the reference fails on cigar[0] here, uberBlast.py:468), or runs outside the arena')
    return out[:int(cols['c_runs'].sum()) if n else 0]


def lex_order(keys):
    """pep_lex_order: numpy.lexsort(keys) for int64 key columns (the last key is the primary one), by radix passes; numpy's own for keys it does not take"""
    lib = load_library()
    keys = [np.ascontiguousarray(k) for k in keys]
    n = len(keys[0]) if keys else 0
    if not keys or n < 64 or any(k.dtype != np.int64 or len(k) != n for k in keys):
        return np.lexsort(keys)
    order = np.empty(n, dtype=np.int64)
    ptrs = (C.c_void_p * len(keys))(*[k.ctypes.data for k in keys])
    rc_ = lib.pep_lex_order(C.c_uint64(n), C.c_int32(len(keys)), ptrs, _ptr(order))
    if rc_ != 0:
        return np.lexsort(keys)
    return order


def set_host_threads(n):
    """pep_set_host_threads: the most threads a pass of the host chain may use (0: the default again); returns the value before"""
    return int(load_library().pep_set_host_threads(C.c_int(int(n))))


def cols_order(q_code, r_code, score):
    """pep_cols_order: the row order of a stable sort by (q_code, r_code, score); codes non-negative"""
    lib = load_library()
    n = len(q_code)
    q_code, r_code = np.ascontiguousarray(q_code, dtype=np.int64), np.ascontiguousarray(r_code, dtype=np.int64)
    score = np.ascontiguousarray(score, dtype=np.float64)
    order = np.empty(n, dtype=np.int64)
    if n and lib.pep_cols_order(C.c_uint64(n), _ptr(q_code), _ptr(r_code), _ptr(score), _ptr(order)) != 0:
        raise PepError('pep_cols_order failed (negative name codes?)')
    return order


def cols_gather(columns, idx):
    """pep_cols_gather: [column[idx] for column in columns] for contiguous 8-byte columns of one length, in one call"""
    lib = load_library()
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    k, n = len(columns), len(idx)
    out = np.empty([max(k, 1), max(n, 1)], dtype=np.int64)
    src = (C.c_void_p * max(k, 1))(*[a.ctypes.data for a in columns])
    dst = (C.c_void_p * max(k, 1))(*[out[c].ctypes.data for c in range(k)])
    for a in columns:
        assert a.flags['C_CONTIGUOUS'] and a.dtype.itemsize == 8
    n_src = len(columns[0]) if k else 0
    if lib.pep_cols_gather(C.c_int32(k), src, dst, _ptr(idx) if n else None, C.c_uint64(n), C.c_uint64(n_src)) != 0:
        raise IndexError('take: row index outside the table')
    return [out[c][:n].view(a.dtype) for c, a in enumerate(columns)]


def store_tab_members(rows, off, keys, date_time, threads=None, order=None):
    """pep_store_tab_members: the finished zip entries of all members of the .tab store (PEPPAN.py:91-113, 972-975) ->
    (bytes of all entries, crc uint32[m], compressed size int64[m], size int64[m], offset of the entry int64[m])"""
    lib = load_library()
    lib.pep_store_tab_members.restype = C.c_int64
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    off, keys = np.ascontiguousarray(off, dtype=np.int64), np.ascontiguousarray(keys, dtype=np.int64)
    order = None if order is None else np.ascontiguousarray(order, dtype=np.int64)
    m = len(keys)
    if rows.ndim != 2 or len(off) != m + 1:
        raise ValueError('store_tab_members: rows int64[n, c], off int64[m + 1], keys int64[m]')
    crc, csize, usize, at = np.empty(m, np.uint32), np.empty(m, np.int64), np.empty(m, np.int64), np.empty(m, np.int64)
    y, mo, d, h, mi, sec = date_time
    dos_date, dos_time = (y - 1980) << 9 | mo << 5 | d, h << 11 | mi << 5 | (sec // 2)
    if threads is None:
        from .configure import effective_cpus
        threads = max(1, min(16, effective_cpus()))
    cap = rows.nbytes + rows.nbytes // 512 + 512 * m + 4096         # an upper bound (deflateBound + .npy header + entry header per member): one call; untouched pages cost nothing
    for _ in range(2):
        buf = np.empty(cap, dtype=np.uint8)
        need = lib.pep_store_tab_members(_ptr(rows), C.c_int64(rows.shape[1]), None if order is None else _ptr(order), _ptr(off), _ptr(keys), C.c_int64(m), C.c_uint32(dos_time), C.c_uint32(dos_date), C.c_int32(threads),
                                         _ptr(buf), C.c_int64(cap), _ptr(crc), _ptr(csize), _ptr(usize), _ptr(at))
        if need < 0:
            raise PepError('pep_store_tab_members failed (%d)' % need)
        if need <= cap:
            return buf[:need], crc, csize, usize, at
        cap = int(need)
    raise PepError('pep_store_tab_members: the entries did not fit the size it had asked for')


_ARGSORT_OK = None


def _argsort_matches_numpy():
    """once per process: does pep_argsort_object_order still leave EQUAL elements where the installed numpy's generic index sort leaves them?  (it restates a private
    routine of numpy - npy_aquicksort - step by step; a release that changes that routine would change the row order of the .tab store silently.)  A few tie-heavy arrays
    of the sizes a genome's rows have; on a difference numpy itself is asked from then on - as _fast_append_ok guards the zip fast path."""
    global _ARGSORT_OK
    if _ARGSORT_OK is None:
        rng = np.random.default_rng(20261003)
        ok = True
        lib = load_library()
        for n, k in ((17, 3), (200, 5), (1500, 30)):
            v = np.ascontiguousarray(rng.integers(0, k, size=n).astype(np.float64))
            out = np.empty(n, dtype=np.int64)
            ok = ok and lib.pep_argsort_object_order(_ptr(v), C.c_int64(n), _ptr(out)) == 0 and np.array_equal(out, np.argsort(v.astype(object)))
        _ARGSORT_OK = bool(ok)
    return _ARGSORT_OK


def argsort_object_order(values):
    """np.argsort(values.astype(object)) for float64 values without NaN - the same steps as numpy's generic index sort, on the doubles (pep_argsort_object_order)"""
    v = np.ascontiguousarray(values, dtype=np.float64)
    if (len(v) and np.isnan(v).any()) or not _argsort_matches_numpy():
        return np.argsort(v.astype(object))
    out = np.empty(len(v), dtype=np.int64)
    rc_ = load_library().pep_argsort_object_order(_ptr(v), C.c_int64(len(v)), _ptr(out))
    if rc_ == -3:                                    # the sort's depth limit: numpy goes on with heapsort there
        return np.argsort(v.astype(object))
    if rc_ != 0:
        raise PepError('pep_argsort_object_order failed (%d)' % rc_)
    return out


def store_tab_archive(rows, off, keys, date_time, threads=None, order=None):
    """pep_store_tab_archive: the .tab store as one complete zip archive (uint8 array), or None when it would need zip64 (>= 65 535 members, >= 4 GiB)"""
    lib = load_library()
    lib.pep_store_tab_archive.restype = C.c_int64
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    off, keys = np.ascontiguousarray(off, dtype=np.int64), np.ascontiguousarray(keys, dtype=np.int64)
    order = None if order is None else np.ascontiguousarray(order, dtype=np.int64)
    m = len(keys)
    y, mo, d, h, mi, sec = date_time
    dos_date, dos_time = (y - 1980) << 9 | mo << 5 | d, h << 11 | mi << 5 | (sec // 2)
    if threads is None:
        from .configure import effective_cpus
        threads = max(1, min(16, effective_cpus()))
    cap = rows.nbytes + rows.nbytes // 512 + 640 * m + 4096         # an upper bound, as above, plus the directory: one call
    for _ in range(2):
        buf = np.empty(cap, dtype=np.uint8)
        need = lib.pep_store_tab_archive(_ptr(rows), C.c_int64(rows.shape[1]), None if order is None else _ptr(order), _ptr(off), _ptr(keys), C.c_int64(m), C.c_uint32(dos_time),
                                         C.c_uint32(dos_date), C.c_int32(threads), _ptr(buf), C.c_int64(cap))
        if need == -3:
            return None
        if need < 0:
            raise PepError('pep_store_tab_archive failed (%d)' % need)
        if need <= cap:
            return buf[:need]
        cap = int(need)
    raise PepError('pep_store_tab_archive: the archive did not fit the size it had asked for')


def similar_classify(T, q, r, rank_ge, rank_le, near_identity, cover):
    """pep_similar_classify: the row-local tests of PEPPAN.py:244-263 over a HitTable's columns -> (action uint8[n], forward uint8[n], iden4 int32[n])"""
    lib = load_library()
    n = len(T)
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    q, r = i64(q), i64(r)
    cols = [i64(T.qs), i64(T.qe), i64(T.ss), i64(T.se), i64(T.ql), i64(T.sl)]
    iden = np.ascontiguousarray(T.iden, dtype=np.float64)
    ge, le = np.ascontiguousarray(rank_ge, dtype=np.uint8), np.ascontiguousarray(rank_le, dtype=np.uint8)
    action, forward, iden4 = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.int32)
    rc_ = lib.pep_similar_classify(C.c_uint64(n), _ptr(q), _ptr(r), _ptr(iden), *([_ptr(c) for c in cols] + [_ptr(ge), _ptr(le), C.c_double(near_identity), C.c_double(cover),
                                   _ptr(action), _ptr(forward), _ptr(iden4)]))
    if rc_ != 0:
        raise PepError('pep_similar_classify failed (%d)' % rc_)
    return action, forward, iden4


def similar_scan(q, r, action, forward, iden4, n_genes):
    """pep_similar_scan: the ordered pass of get_similar_pairs (PEPPAN.py:231-276) over numeric columns -> dict(alive, seen_as_query,
    absorbed int64[m, 3], ev_kind, ev_a, ev_b, ev_row_off, ev_rows)"""
    lib = load_library()
    n = len(q)
    q, r = np.ascontiguousarray(q, dtype=np.int64), np.ascontiguousarray(r, dtype=np.int64)
    action, forward = np.ascontiguousarray(action, dtype=np.uint8), np.ascontiguousarray(forward, dtype=np.uint8)
    iden4 = np.ascontiguousarray(iden4, dtype=np.int32)
    alive, seen = np.zeros(max(n_genes, 1), dtype=np.uint8), np.zeros(max(n_genes, 1), dtype=np.uint8)
    absorbed = np.zeros((n + 1, 3), dtype=np.int64)
    ev_kind, ev_a, ev_b = np.zeros(n + 1, dtype=np.uint8), np.zeros(n + 1, dtype=np.int64), np.zeros(n + 1, dtype=np.int64)
    ev_row_off, ev_rows = np.zeros(n + 2, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint64)
    na, ne = C.c_uint64(), C.c_uint64()
    rc_ = lib.pep_similar_scan(C.c_uint64(n), _ptr(q), _ptr(r), _ptr(action), _ptr(forward), _ptr(iden4), C.c_uint64(n_genes), _ptr(alive), _ptr(seen),
                               _ptr(absorbed), C.byref(na), _ptr(ev_kind), _ptr(ev_a), _ptr(ev_b), _ptr(ev_row_off), _ptr(ev_rows), C.byref(ne))
    if rc_ != 0:
        raise PepError('pep_similar_scan failed (%d)' % rc_)
    ne, na = ne.value, na.value
    off = ev_row_off[:ne + 1].astype(np.int64)
    return dict(alive=alive[:n_genes], seen_as_query=seen[:n_genes], absorbed=absorbed[:na], ev_kind=ev_kind[:ne], ev_a=ev_a[:ne], ev_b=ev_b[:ne],
                ev_row_off=off, ev_rows=ev_rows[:int(off[-1])].astype(np.int64))


def fasta_keep(path, ids):
    """pep_fasta_keep: rewrite the FASTA file so that only records named by one of the integers `ids` stay -> (records, kept), or None when
    a record's name is not a plain decimal integer (file untouched: the caller goes its own way)"""
    lib = load_library()
    ids = np.unique(np.asarray(ids, dtype=np.int64))
    nr, nk = C.c_uint64(), C.c_uint64()
    rc_ = lib.pep_fasta_keep(os.fsencode(path), _ptr(ids if len(ids) else np.zeros(1, np.int64)), C.c_uint64(len(ids)), C.byref(nr), C.byref(nk))
    if rc_ == -2:                                   # PEP_ERR_ARG: a name that is not a plain integer, or no such file
        return None
    if rc_ != 0:
        raise PepError('pep_fasta_keep failed (%d)' % rc_)
    return nr.value, nk.value


def fasta_scan(data, table, n_records):
    """pep_fasta_scan: the sequences of FASTA text `data` (bytes) as (codes uint8[total], off uint64[n + 1]) with codes = table[byte], or None when
    the text does not hold exactly n_records records or a sequence holds non-ASCII bytes (the caller then goes its own way)"""
    lib = load_library()
    table = np.ascontiguousarray(table, dtype=np.uint8)
    assert len(table) == 256
    codes = np.empty(max(len(data), 1), dtype=np.uint8)
    off = np.zeros(n_records + 2, dtype=np.uint64)
    nr, high = C.c_uint64(), C.c_int32()
    rc_ = lib.pep_fasta_scan(C.c_char_p(data), C.c_uint64(len(data)), _ptr(table), _ptr(codes), _ptr(off), C.c_uint64(n_records), C.byref(nr), C.byref(high))
    if rc_ == -3 or (rc_ == 0 and (nr.value != n_records or high.value)):          # PEP_ERR_LIMIT: more records than the caller counted
        return None
    if rc_ != 0:
        raise PepError('pep_fasta_scan failed (%d)' % rc_)
    return codes[:int(off[n_records])], off[:n_records + 1]


_UPPER = np.frombuffer(bytes(range(256)).upper(), dtype=np.uint8)


_SCRATCH = threading.local()


def fasta_records(data, as_dict=False):
    """pep_fasta_records: the records of FASTA text `data` (ASCII bytes without carriage returns) as (names, text, off): names = list of str (first word
    of every header line), text = all sequences upper-cased and without blanks in one str, off = int64[n + 1] where each record's sequence starts in it.
    None when a header has no name or a sequence holds non-ASCII bytes: the caller then goes its own way.  as_dict: the records as {name: sequence}
    (of two records with one name the later one) instead."""
    lib = load_library()
    # the cleaned sequences land in a scratch buffer this thread keeps (a fresh 10 MB array costs a page fault per 4 KiB - a third of the scan)
    codes = getattr(_SCRATCH, 'codes', None)
    if codes is None or len(codes) < len(data) or not as_dict:
        codes = np.empty(max(len(data), 1), dtype=np.uint8)
        if as_dict and len(data) <= (64 << 20):
            _SCRATCH.codes = codes
    nr, high = C.c_uint64(), C.c_int32()
    cap = max(1024, len(data) // 128)                # (a guess; a file of shorter records is counted and scanned again)
    while True:
        off, name_off, name_len = np.empty(cap + 1, dtype=np.uint64), np.empty(cap, dtype=np.uint64), np.empty(cap, dtype=np.uint32)
        rc_ = lib.pep_fasta_records(C.c_char_p(data), C.c_uint64(len(data)), _ptr(_UPPER), _ptr(codes), _ptr(off), _ptr(name_off), _ptr(name_len), C.c_uint64(cap),
                                    C.byref(nr), C.byref(high))
        if rc_ != -3 or cap >= len(data):            # PEP_ERR_LIMIT: more records than guessed
            break
        cap = data.count(b'>') + 1
    if rc_ != 0:
        raise PepError('pep_fasta_records failed (%d)' % rc_)
    n = nr.value
    if high.value or (n and int(name_len[:n].min()) == 0):
        return None
    if as_dict:                                      # {name: sequence} made by one C loop over the buffers (csrc/pyrows.c)
        from .hittable import _pyrows
        return _pyrows().pep_records_dict(data, name_off.ctypes.data, name_len.ctypes.data, codes.ctypes.data, off.ctypes.data, n)
    a = name_off[:n].astype(np.int64)
    b = a + name_len[:n]
    names = [data[x:y].decode('ascii') for x, y in zip(a.tolist(), b.tolist())]
    off = off[:n + 1].astype(np.int64)
    return names, str(codes[:int(off[n])].data, 'ascii'), off


def similar_resolve(ev_kind, ev_a, ev_b, ev_value):
    """pep_similar_resolve: ortho_pairs as the reference's dictionary builds it -> int64[m, 3] (a, b, value), value != 0, insertion order"""
    lib = load_library()
    n = len(ev_kind)
    ev_kind = np.ascontiguousarray(ev_kind, dtype=np.uint8)
    ev_a, ev_b = np.ascontiguousarray(ev_a, dtype=np.int64), np.ascontiguousarray(ev_b, dtype=np.int64)
    ev_value = np.ascontiguousarray(ev_value, dtype=np.int32)
    out = np.zeros((n + 1, 3), dtype=np.int64)
    no = C.c_uint64()
    rc_ = lib.pep_similar_resolve(C.c_uint64(n), _ptr(ev_kind), _ptr(ev_a), _ptr(ev_b), _ptr(ev_value), _ptr(out), C.byref(no))
    if rc_ != 0:
        raise PepError('pep_similar_resolve failed (%d)' % rc_)
    return out[:no.value]


def merge_hits(hits, cigar, top_k, n_splits, out=None):
    """pep_merge_hits: the union of the hit tables of several TARGET shards (global q / t indices) -> the table of the unsharded
    search: top-k per (q, t mod n_splits) re-applied, rows ordered by (q, t, bin), CIGAR arena compacted.
    `out`: optional dict that keeps the output arrays between calls (the result is then only valid until the next call)."""
    lib = load_library()
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    if out is None:
        out_h = np.empty(max(len(hits), 1), dtype=HIT_DTYPE)
        out_c = np.empty(max(len(cigar), 1), dtype=np.uint32)
    else:
        if len(out.get('h', ())) < max(len(hits), 1):
            out['h'] = np.empty(int(len(hits) * 1.5) + 64, dtype=HIT_DTYPE)
        if len(out.get('c', ())) < max(len(cigar), 1):
            out['c'] = np.empty(int(len(cigar) * 1.5) + 64, dtype=np.uint32)
        out_h, out_c = out['h'], out['c']
    nh, nc = C.c_uint64(), C.c_uint64()
    rc_ = lib.pep_merge_hits(C.c_uint64(len(hits)), _ptr(hits) if len(hits) else None, _ptr(cigar) if len(cigar) else None, C.c_uint64(len(cigar)),
                             C.c_int32(int(top_k)), C.c_int32(int(n_splits)), _ptr(out_h), _ptr(out_c), C.byref(nh), C.byref(nc))
    if rc_ != 0:
        raise PepError('pep_merge_hits failed (%d)' % rc_)
    return out_h[:nh.value], out_c[:nc.value]


class Context(object):
    """one GPU context (one per process per device; create it AFTER any fork)"""

    def __init__(self, device=0):
        self._lib = load_library()
        if self._lib.pep_device_count() <= 0:
            raise PepError('no HIP device visible: peppan_amd needs an MI355X (there is no CPU fallback)')
        h = C.c_void_p()
        rc = self._lib.pep_ctx_create(int(device), C.byref(h))
        self._h = h
        self._view = None
        self._nt_match_on, self.last_nt_match = False, None
        self._grouping, self.labels = 0, None           # set_grouping: K10 as the tail of every search
        self.upload_generation = 0           # bumped by every call that replaces a device-resident sequence set (see RunBlast._ensure_nt)
        self.q_nt_token = self.r_nt_token = None     # what the nucleotide sets on the device were made from (set by RunBlast._ensure_nt, cleared by any set_*)
        if rc != 0:
            msg = self._lib.pep_last_error(h).decode() if h else 'pep_ctx_create failed'
            if h:
                self._lib.pep_ctx_destroy(h)
                self._h = None
            raise PepError('pep_ctx_create(%d): %d %s' % (device, rc, msg))
        self.device = device

    def close(self):
        if getattr(self, '_h', None):
            if getattr(self, '_view', None) is not None:
                self._lib.pep_result_free(self._view)
                self._view = None
            self._lib.pep_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise PepError('%s failed (%d): %s' % (what, rc, self._lib.pep_last_error(self._h).decode()))

    # ---- inputs
    def set_query_nt(self, seqs, gtable=11):
        self.upload_generation += 1
        self.q_nt_token = None
        nt, off = _pack(seqs)
        self._check(self._lib.pep_set_query_nt(self._h, _ptr(nt), _ptr(off), C.c_uint32(_count(seqs)), C.c_int(gtable)), 'pep_set_query_nt')

    def set_ref_nt(self, seqs, frames=6, gtable=11):
        self.upload_generation += 1
        self.r_nt_token = None
        nt, off = _pack(seqs)
        self._check(self._lib.pep_set_ref_nt(self._h, _ptr(nt), _ptr(off), C.c_uint32(_count(seqs)), C.c_int(frames), C.c_int(gtable)), 'pep_set_ref_nt')

    def set_query_aa(self, seqs):
        self.upload_generation += 1
        self.q_nt_token = None
        aa, off = _pack(seqs)
        self._check(self._lib.pep_set_query_aa(self._h, _ptr(aa), _ptr(off), C.c_uint32(_count(seqs))), 'pep_set_query_aa')

    def set_ref_aa(self, seqs):
        self.upload_generation += 1
        self.r_nt_token = None
        aa, off = _pack(seqs)
        self._check(self._lib.pep_set_ref_aa(self._h, _ptr(aa), _ptr(off), C.c_uint32(_count(seqs))), 'pep_set_ref_aa')

    def set_target_groups(self, groups):
        """groups: one non-decreasing id per reference sequence (None / empty clears): batch of reference sets in one search"""
        g = np.ascontiguousarray(groups if groups is not None else [], dtype=np.uint32)
        self._check(self._lib.pep_set_target_groups(self._h, _ptr(g) if len(g) else None, C.c_uint32(len(g))), 'pep_set_target_groups')

    def use_nt_as_residues(self, strands=2):
        """the device-resident nucleotide sets themselves become the residue sets (base codes; reference: forward strands, then reverse
        complements, per target group) - the inputs of the nucleotide search.  Until the next translate() / set_*."""
        self._drop_view()
        self._check(self._lib.pep_use_nt_as_residues(self._h, C.c_int(strands)), 'pep_use_nt_as_residues')

    def set_timing(self, level):
        """phase timers of the searches (the ms_* statistics): 0 none (default), 1 the score pass only, 2 every phase - each HIP event costs
        the GPU about 6 us of idle time between two kernels"""
        self._check(self._lib.pep_set_timing(self._h, C.c_int(level)), 'pep_set_timing')

    def set_grouping(self, n_nodes, node_of_target=None, q_base=0):
        """every search of this context ends with single linkage (K10) over its own hit table: edges (hit.q + q_base, node_of_target[hit.t]);
        the labels of the newest search are in self.labels afterwards.  n_nodes = 0 switches it off."""
        nn = np.ascontiguousarray(node_of_target if node_of_target is not None else [], dtype=np.uint32)
        self._check(self._lib.pep_set_grouping(self._h, C.c_uint32(n_nodes), C.c_uint32(q_base), _ptr(nn) if len(nn) else None, C.c_uint64(len(nn))), 'pep_set_grouping')
        self._grouping = int(n_nodes)
        self.labels = None

    def _take_labels(self, r):
        if getattr(self, '_grouping', 0):
            lab = np.empty(self._grouping, dtype=np.uint32)
            self._check(self._lib.pep_result_labels(r, _ptr(lab), C.c_uint32(self._grouping)), 'pep_result_labels')
            self.labels = lab

    def invalidate_translation(self):
        """the next search() runs K1 again, inside the search (cheaper than translate(force=True) in front of it: no host wait between K1 and the search)"""
        self._check(self._lib.pep_invalidate_translation(self._h), 'pep_invalidate_translation')

    def translate(self, force=False):
        self._check(self._lib.pep_translate(self._h, C.c_int(1 if force else 0)), 'pep_translate')

    # ---- K1 products
    def query_meta(self):
        n, r = C.c_uint32(), C.c_uint64()
        self._check(self._lib.pep_query_count(self._h, C.byref(n), C.byref(r)), 'pep_query_count')
        out = np.zeros(n.value, dtype=QUERY_META_DTYPE)
        self._check(self._lib.pep_get_query_meta(self._h, _ptr(out), C.c_uint32(n.value)), 'pep_get_query_meta')
        return out

    def target_meta(self):
        n, r = C.c_uint32(), C.c_uint64()
        self._check(self._lib.pep_target_count(self._h, C.byref(n), C.byref(r)), 'pep_target_count')
        out = np.zeros(n.value, dtype=TARGET_META_DTYPE)
        self._check(self._lib.pep_get_target_meta(self._h, _ptr(out), C.c_uint32(n.value)), 'pep_get_target_meta')
        return out

    def _get_aa(self, count_fn, get_fn, what):
        n, r = C.c_uint32(), C.c_uint64()
        self._check(count_fn(self._h, C.byref(n), C.byref(r)), what)
        codes = np.zeros(max(1, r.value), dtype=np.uint8)
        off = np.zeros(n.value + 1, dtype=np.uint64)
        self._check(get_fn(self._h, _ptr(codes), C.c_uint64(r.value), _ptr(off)), what)
        return codes[:r.value], off

    def query_aa(self):
        return self._get_aa(self._lib.pep_query_count, self._lib.pep_get_query_aa, 'pep_get_query_aa')

    def target_aa(self):
        return self._get_aa(self._lib.pep_target_count, self._lib.pep_get_target_aa, 'pep_get_target_aa')

    def _drop_view(self):
        if self._view is not None:                      # the handle behind the previous zero-copy views: released first, so that
            self._lib.pep_result_free(self._view)       # the library does not preserve a table nobody may look at any more
            self._view = None

    # ---- search
    def search(self, params=None, copy=True):
        """returns (hits [HIT_DTYPE], cigar uint32 [len<<2|op], stats dict).
        copy=False: the arrays are views of the library's pinned staging memory - no 2 MB copy and no fresh pages - valid only
        until the next search on this context (for callers that consume the table at once, like hits_to_blastab)."""
        self._drop_view()
        r = C.c_void_p()
        self._check(self._lib.pep_search(self._h, C.byref(params) if params is not None else None, C.byref(r)), 'pep_search')
        try:
            nh, nc = C.c_uint64(), C.c_uint64()
            self._check(self._lib.pep_result_size(r, C.byref(nh), C.byref(nc)), 'pep_result_size')
            st = Stats()
            self._check(self._lib.pep_result_stats(r, C.byref(st)), 'pep_result_stats')
            self._take_labels(r)
            self.last_nt_match = None
            if self._nt_match_on and nh.value:
                pm = C.c_void_p()
                self._check(self._lib.pep_result_nt_match(r, C.byref(pm)), 'pep_result_nt_match')
                if pm.value:
                    self.last_nt_match = np.frombuffer((C.c_char * (nh.value * 4)).from_address(pm.value), dtype=np.uint32).copy()
            if copy or nh.value == 0:
                hits = np.empty(nh.value, dtype=HIT_DTYPE)
                cig = np.empty(nc.value, dtype=np.uint32)
                self._check(self._lib.pep_result_copy(r, _ptr(hits), _ptr(cig)), 'pep_result_copy')
            else:
                ph, pc = C.c_void_p(), C.c_void_p()
                self._check(self._lib.pep_result_data(r, C.byref(ph), C.byref(pc)), 'pep_result_data')
                hits = np.frombuffer((C.c_char * (nh.value * HIT_DTYPE.itemsize)).from_address(ph.value), dtype=HIT_DTYPE)
                cig = (np.frombuffer((C.c_char * (nc.value * 4)).from_address(pc.value), dtype=np.uint32) if nc.value else np.empty(0, np.uint32))
                self._view, r = r, None                 # keep the handle alive while the views may be in use
        finally:
            if r is not None:
                self._lib.pep_result_free(r)
        return hits, cig, {n: getattr(st, n) for n, _ in Stats._fields_}

    def search_on_device(self, params=None):
        """the search with its hit table LEFT ON THE DEVICE: -> (n_hits, n_cigar, stats dict, (address of the hit records, address of the
        CIGAR arena)).  The addresses point into the context's workspace and stay valid until the next search / linclust on this context;
        result_to_host() fetches the table afterwards if somebody wants it after all."""
        if self._view is not None:
            self._lib.pep_result_free(self._view)
            self._view = None
        self._check(self._lib.pep_set_result_mode(self._h, C.c_int(1)), 'pep_set_result_mode')
        r = C.c_void_p()
        try:
            self._check(self._lib.pep_search(self._h, C.byref(params) if params is not None else None, C.byref(r)), 'pep_search')
        finally:
            self._lib.pep_set_result_mode(self._h, C.c_int(0))
        nh, nc = C.c_uint64(), C.c_uint64()
        st = Stats()
        ph, pc = C.c_void_p(), C.c_void_p()
        self._view = r
        self._check(self._lib.pep_result_size(r, C.byref(nh), C.byref(nc)), 'pep_result_size')
        self._check(self._lib.pep_result_stats(r, C.byref(st)), 'pep_result_stats')
        if nh.value:
            self._check(self._lib.pep_result_device(r, C.byref(ph), C.byref(pc)), 'pep_result_device')
        return nh.value, nc.value, {n: getattr(st, n) for n, _ in Stats._fields_}, (ph.value or 0, pc.value or 0)

    def result_to_host(self):
        """(hits, cigar) of the result search_on_device() is holding"""
        nh, nc = C.c_uint64(), C.c_uint64()
        self._check(self._lib.pep_result_size(self._view, C.byref(nh), C.byref(nc)), 'pep_result_size')
        hits, cig = np.empty(nh.value, dtype=HIT_DTYPE), np.empty(nc.value, dtype=np.uint32)
        self._check(self._lib.pep_result_copy(self._view, _ptr(hits), _ptr(cig)), 'pep_result_copy')
        return hits, cig

    # ---- K7
    def set_nt_match(self, on):
        """pep_set_nt_match: the searches of this context also count the identical nucleotide columns of every hit (K7's n_match) -> last_nt_match after search()"""
        self._check(self._lib.pep_set_nt_match(self._h, C.c_int(1 if on else 0)), 'pep_set_nt_match')
        self._nt_match_on = bool(on)

    def rescore_nt(self, nt_hits, cigar):
        nt_hits = np.ascontiguousarray(nt_hits, dtype=NT_HIT_DTYPE)
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        out = np.zeros((len(nt_hits), 5), dtype=np.int64)
        if len(nt_hits):
            cg = cigar if len(cigar) else np.zeros(1, np.uint32)
            self._check(self._lib.pep_rescore_nt(self._h, C.c_uint64(len(nt_hits)), _ptr(nt_hits), _ptr(cg), C.c_uint64(len(cigar)), _ptr(out)), 'pep_rescore_nt')
        return out

    # ---- K14
    def pair_support(self, rows, cigar, grp_off, grp_qlen, grp_rlen, limits):
        """get_similar (PEPPAN.py:195-224) for groups of forward alignments: rows SUPPORT_ROW_DTYPE, group g = rows [grp_off[g], grp_off[g+1])
        -> int32 per group: SUPPORT_NONE | 0 | int(mean identity * 10000)"""
        rows = np.ascontiguousarray(rows, dtype=SUPPORT_ROW_DTYPE)
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        grp_off = np.ascontiguousarray(grp_off, dtype=np.uint64)
        grp_qlen, grp_rlen = np.ascontiguousarray(grp_qlen, dtype=np.uint32), np.ascontiguousarray(grp_rlen, dtype=np.uint32)
        ng = len(grp_qlen)
        value = np.full(max(ng, 1), SUPPORT_NONE, dtype=np.int32)
        if ng:
            cg = cigar if len(cigar) else np.zeros(1, np.uint32)
            rr = rows if len(rows) else np.zeros(1, SUPPORT_ROW_DTYPE)
            self._check(self._lib.pep_pair_support(self._h, C.c_uint64(len(rows)), _ptr(rr), _ptr(cg), C.c_uint64(len(cigar)), C.c_uint64(ng), _ptr(grp_off),
                                                   _ptr(grp_qlen), _ptr(grp_rlen), C.byref(limits), _ptr(value)), 'pep_pair_support')
        return value[:ng]

    # ---- K9
    def linclust(self, seqs, min_id, min_cov, base=4, k=17, m=20):
        """seqs: list of uint8 code arrays, or their (concatenation, uint64 offsets[n + 1]) -> (uint32 representative index per sequence, stats dict)"""
        codes, off = _pack(seqs)
        n = _count(seqs)
        self.upload_generation += 1                     # the gapped stage takes over the packed sequence sets
        rep = np.zeros(n, dtype=np.uint32)
        stats = np.zeros(3, dtype=np.uint64)
        if n:
            self._check(self._lib.pep_linclust(self._h, _ptr(codes), _ptr(off), C.c_uint32(n), C.c_int(base), C.c_int(k), C.c_int(m),
                                               C.c_double(min_id), C.c_double(min_cov), _ptr(rep), _ptr(stats)), 'pep_linclust')
        return rep, dict(selected=int(stats[0]), verified=int(stats[1]), accepted=int(stats[2]))

    # ---- K11
    def overlaps(self, contig, start, end, row_id, ovl_l, ovl_p):
        """rows sorted by (contig, start, end) -> int64[m, 3] (id1, id2, overlap) in sweep order"""
        contig = np.ascontiguousarray(contig, dtype=np.int32)
        start, end, row_id = (np.ascontiguousarray(x, dtype=np.int64) for x in (start, end, row_id))
        n = len(contig)
        if n == 0:
            return np.zeros((0, 3), dtype=np.int64)
        m = C.c_uint64()
        cap = max(1024, 4 * n)
        for _ in range(2):
            out = np.zeros((cap, 3), dtype=np.int64)
            self._check(self._lib.pep_overlaps(self._h, C.c_uint64(n), _ptr(contig), _ptr(start), _ptr(end), _ptr(row_id), C.c_double(ovl_l),
                                               C.c_double(ovl_p), _ptr(out), C.c_uint64(cap), C.byref(m)), 'pep_overlaps')
            if m.value <= cap:
                return out[:m.value]
            cap = m.value
        raise PepError('pep_overlaps: inconsistent pair count')

    # ---- K12
    def alleles(self, contigs, rows, cigar, grp_off, grp_qlen, gtable=11):
        """contigs: list of ASCII contig strings/bytes; rows: LOCUS_DTYPE records, the rows of group g at [grp_off[g], grp_off[g+1]);
        cigar: uint32 runs len<<2|op -> (in_frame int64[n], orf int64[n], packed uint8[sum ceil(q_len/3)])"""
        rows = np.ascontiguousarray(rows, dtype=LOCUS_DTYPE)
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        grp_off = np.ascontiguousarray(grp_off, dtype=np.uint64)
        grp_qlen = np.ascontiguousarray(grp_qlen, dtype=np.uint32)
        nt, off = _pack(contigs)
        n, ng = len(rows), len(grp_qlen)
        in_frame, orf = np.zeros(max(n, 1), dtype=np.int64), np.zeros(max(n, 1), dtype=np.int64)
        total = int(((grp_qlen.astype(np.int64) + 2) // 3).sum())
        packed = np.zeros(max(total, 1), dtype=np.uint8)
        cg = cigar if len(cigar) else np.zeros(1, np.uint32)
        self._check(self._lib.pep_alleles(self._h, _ptr(nt), _ptr(off), C.c_uint32(len(contigs)), C.c_uint64(n), _ptr(rows), _ptr(cg), C.c_uint64(len(cigar)),
                                          C.c_uint32(ng), _ptr(grp_off), _ptr(grp_qlen) if ng else None, C.c_int(gtable), _ptr(in_frame), _ptr(orf),
                                          _ptr(packed), C.c_uint64(total)), 'pep_alleles')
        return in_frame[:n], orf[:n], packed[:total]

    # ---- K13
    def sha1(self, seqs):
        """list of str / bytes -> uint8[n, 20] SHA-1 digests (hashlib.sha1(s).digest() of each)"""
        data, off = _pack(seqs)
        out = np.zeros((max(len(seqs), 1), 20), dtype=np.uint8)
        if len(seqs):
            self._check(self._lib.pep_sha1(self._h, _ptr(data), _ptr(off), C.c_uint32(len(seqs)), _ptr(out)), 'pep_sha1')
        return out[:len(seqs)]

    def dedup(self, lengths, digests):
        """genes in priority order -> uint32 rep[i] = first gene of the same (length run, digest); rep[i] == i for the ones kept"""
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        digests = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 20)
        assert len(lengths) == len(digests)
        rep = np.zeros(max(len(lengths), 1), dtype=np.uint32)
        if len(lengths):
            self._check(self._lib.pep_dedup(self._h, C.c_uint32(len(lengths)), _ptr(lengths), _ptr(digests), _ptr(rep)), 'pep_dedup')
        return rep[:len(lengths)]

    # ---- K10
    def components_of_hits(self, n_nodes, hits, node_of_target, q_base=0):
        """labels of the graph with one edge (hit.q + q_base, node_of_target[hit.t]) per hit"""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        node_of_target = np.ascontiguousarray(node_of_target, dtype=np.uint32)
        lab = np.zeros(n_nodes, dtype=np.uint32)
        if n_nodes:
            hh = hits if len(hits) else np.zeros(1, HIT_DTYPE)
            nn = node_of_target if len(node_of_target) else np.zeros(1, np.uint32)
            self._check(self._lib.pep_components_of_hits(self._h, C.c_uint32(n_nodes), C.c_uint64(len(hits)), _ptr(hh), C.c_uint32(q_base), _ptr(nn),
                                                         C.c_uint64(len(node_of_target)), _ptr(lab)), 'pep_components_of_hits')
        return lab

    def components_of_search(self, n_nodes, node_of_target, q_base=0):
        """labels of the graph with one edge (hit.q + q_base, node_of_target[hit.t]) per hit of the NEWEST search(copy=False) on this context,
        read from the table's device copy (pep_components_of_result)"""
        if self._view is None:
            raise PepError('components_of_search: no search result is held (call search(copy=False) first)')
        node_of_target = np.ascontiguousarray(node_of_target, dtype=np.uint32)
        lab = np.zeros(n_nodes, dtype=np.uint32)
        if n_nodes:
            nn = node_of_target if len(node_of_target) else np.zeros(1, np.uint32)
            self._check(self._lib.pep_components_of_result(self._h, self._view, C.c_uint32(n_nodes), C.c_uint32(q_base), _ptr(nn), C.c_uint64(len(node_of_target)),
                                                           _ptr(lab)), 'pep_components_of_result')
        return lab

    def components(self, n_nodes, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        lab = np.zeros(n_nodes, dtype=np.uint32)
        if n_nodes:
            aa = a if len(a) else np.zeros(1, np.uint32)
            bb = b if len(b) else np.zeros(1, np.uint32)
            self._check(self._lib.pep_components(self._h, C.c_uint32(n_nodes), C.c_uint64(len(a)), _ptr(aa), _ptr(bb), _ptr(lab)), 'pep_components')
        return lab
