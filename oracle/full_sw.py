"""ctypes front end of oracle/_build/libfullsw.so (full_sw.c): exhaustive full-matrix Gotoh local alignment.

TEST INFRASTRUCTURE ONLY - the independent ground truth the seed-and-band heuristic (align_oracle.c == the HIP
kernels) is measured against: optimality of every reported score and recall per identity bin
(tests/test_recall_full_sw.py, tools/recall_report.py).  Never imported by the product (peppan_amd/)."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libfullsw.so')
_lib = None


class Aln(C.Structure):
    _fields_ = [('score', C.c_int32), ('q_start', C.c_int32), ('q_end', C.c_int32), ('t_start', C.c_int32), ('t_end', C.c_int32),
                ('n_ident', C.c_uint32), ('aln_len', C.c_uint32), ('n_runs', C.c_uint32)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(['make', '-C', HERE, '-s'])
        _lib = C.CDLL(LIB)
    return _lib


def _pack(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs])
    res = np.concatenate(seqs).astype(np.uint8) if len(seqs) and off[-1] else np.zeros(1, np.uint8)
    return np.ascontiguousarray(res), off


def _sub(params):
    return np.ascontiguousarray(np.frombuffer(bytes(bytearray(C.string_at(C.addressof(params.sub), 1024))), dtype=np.int8))


def score_matrix(q_seqs, t_seqs, params, threads=0):
    """best local score of every pair: int32[len(q_seqs), len(t_seqs)].  params: any block with sub / gap_open / gap_ext"""
    L = lib()
    from .oracle import granted_cpus
    L.fullsw_set_threads(C.c_int(threads if threads > 0 else granted_cpus()))
    qr, qo = _pack(q_seqs)
    tr, to = _pack(t_seqs)
    out = np.zeros((len(q_seqs), len(t_seqs)), dtype=np.int32)
    sub = _sub(params)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.fullsw_score_matrix(P(sub), C.c_int(params.gap_open), C.c_int(params.gap_ext), P(qr), P(qo), C.c_uint32(len(q_seqs)),
                               P(tr), P(to), C.c_uint32(len(t_seqs)), P(out))
    assert rc == 0
    return out


def align(q, t, params):
    """one pair with traceback -> (Aln, cigar uint32[len<<2|op])"""
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    a = Aln()
    cig = np.zeros(len(q) + len(t) + 2, dtype=np.uint32)
    sub = _sub(params)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = lib().fullsw_align(P(sub), C.c_int(params.gap_open), C.c_int(params.gap_ext), P(q), C.c_int32(len(q)), P(t), C.c_int32(len(t)),
                            C.byref(a), P(cig), C.c_uint32(len(cig)))
    assert rc == 0, rc
    return a, cig[:a.n_runs].copy()
