"""CPU-side checks of the drop-in boundary: the library builds, loads and exports every symbol of include/peppan_hip.h."""
import ctypes as C
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    G.build()
    from peppan_amd import _native as N
    lib = N.load_library()
    hdr = open(os.path.join(ROOT, 'include', 'peppan_hip.h')).read()
    declared = set(re.findall(r'\b(pep_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(N.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pep_version() == N.ABI_VERSION


def test_struct_layouts_match_header(tmp_path):
    """sizes and field offsets of every struct that crosses the boundary, taken from the header by a C compiler, equal the ctypes / numpy
    mirrors in peppan_amd/_native.py"""
    import subprocess
    from peppan_amd import _native as N
    probes = [('pep_search_params', 'min_id_pct'), ('pep_search_params', 'dbsize'), ('pep_search_params', 'ka_lambda'), ('pep_search_params', 'hsp_mode'),
              ('pep_search_params', 't_index_base'), ('pep_stats', 'cells_swept_trace'), ('pep_stats', 'candidates_settled'), ('pep_stats', 'cells_settled'), ('pep_stats', 'ms_seed'), ('pep_stats', 'ms_seed_match'),
              ('pep_hit', 'cigar_off'), ('pep_hit', 'cells'), ('pep_nt_hit', 'cigar_off'), ('pep_locus', 'cigar_off'),
              ('pep_mat_cols', 'iden'), ('pep_mat_cols', 'arena'), ('pep_mat_cols', 'rid'), ('pep_mat_cols', 'score_is_int')]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "peppan_hip.h"\nint main(void) {\n'
    for st in ('pep_search_params', 'pep_stats', 'pep_hit', 'pep_nt_hit', 'pep_locus', 'pep_query_meta', 'pep_target_meta', 'pep_mat_cols'):
        src += '  printf("%s %%zu\\n", sizeof(%s));\n' % (st, st)
    for st, f in probes:
        src += '  printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (st, f, st, f)
    src += '  return 0;\n}\n'
    (tmp_path / 'probe.c').write_text(src)
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', str(tmp_path / 'probe'), str(tmp_path / 'probe.c')])
    got = dict(line.split() for line in subprocess.check_output([str(tmp_path / 'probe')]).decode().splitlines())
    got = {k: int(v) for k, v in got.items()}
    assert got['pep_search_params'] == C.sizeof(N.SearchParams) and got['pep_stats'] == C.sizeof(N.Stats)
    assert got['pep_hit'] == N.HIT_DTYPE.itemsize == 64 and got['pep_nt_hit'] == N.NT_HIT_DTYPE.itemsize == 40
    assert got['pep_locus'] == N.LOCUS_DTYPE.itemsize == 32 and got['pep_mat_cols'] == C.sizeof(N.MatCols)
    assert got['pep_query_meta'] == N.QUERY_META_DTYPE.itemsize == 16 and got['pep_target_meta'] == N.TARGET_META_DTYPE.itemsize == 16
    for st, f in probes:
        mirror = {'pep_search_params': N.SearchParams, 'pep_stats': N.Stats, 'pep_mat_cols': N.MatCols}.get(st)
        if mirror is not None:
            assert got[st + '.' + f] == getattr(mirror, f).offset, (st, f)
        else:
            dt = {'pep_hit': N.HIT_DTYPE, 'pep_nt_hit': N.NT_HIT_DTYPE, 'pep_locus': N.LOCUS_DTYPE}[st]
            assert got[st + '.' + f] == dt.fields[f][1], (st, f)


def test_merge_hits_is_the_unsharded_topk():
    """pep_merge_hits (host C++): union of per-shard tables -> top-k per (q, t mod splits) by (score desc, t asc, bin asc), rows by (q, t, bin)"""
    import numpy as np
    from peppan_amd import _native as N
    rng = np.random.default_rng(5)
    rows, cig = [], []
    for q in range(40):
        for t in rng.choice(400, size=int(rng.integers(0, 60)), replace=False).tolist():
            runs = [(int(rng.integers(5, 90)) << 2) | int(k % 3) for k in range(int(rng.integers(1, 5)))]
            rows.append((q, t, 1, 10, 1, 10, int(rng.integers(60, 70)), 0, 10, 10, len(runs), int(rng.integers(0, 3)), len(cig), 7))
            cig += runs
    hits = np.array(rows, dtype=N.HIT_DTYPE)
    cig = np.array(cig, dtype=np.uint32)
    perm = rng.permutation(len(hits))                              # shards arrive in any order
    out_h, out_c = N.merge_hits(hits[perm], cig, 3, 5)
    want = []
    for q in range(40):
        for sp in range(5):
            grp = [h for h in hits if h['q'] == q and h['t'] % 5 == sp]
            grp.sort(key=lambda h: (-int(h['score']), int(h['t']), int(h['bin'])))
            want += grp[:3]
    want.sort(key=lambda h: (int(h['q']), int(h['t']), int(h['bin'])))
    assert len(out_h) == len(want) > 200
    off = 0
    for g, w in zip(out_h, want):
        for f in N.HIT_DTYPE.names:
            if f != 'cigar_off':
                assert g[f] == w[f]
        assert g['cigar_off'] == off
        assert out_c[off:off + int(g['cigar_runs'])].tolist() == cig[int(w['cigar_off']):int(w['cigar_off']) + int(w['cigar_runs'])].tolist()
        off += int(g['cigar_runs'])
    assert off == len(out_c)
    e_h, e_c = N.merge_hits(hits[:0], cig[:0], 3, 5)
    assert len(e_h) == 0 and len(e_c) == 0


def test_no_gpu_is_a_loud_error():
    from peppan_amd import _native as N
    lib = N.load_library()
    if lib.pep_device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(N.PepError):
        N.Context(0)


def test_default_params_and_min_score_match_oracle():
    from peppan_amd import _native as N
    from oracle import oracle as O
    p, o = N.default_params(), O.default_params()
    assert list(p.sub) == list(o.sub) and list(p.reduce) == list(o.reduce)
    assert p.base == o.base and p.n_shapes == o.n_shapes and list(p.weight) == list(o.weight)
    assert [list(x) for x in p.offs] == [list(x) for x in o.offs]
    assert (p.gap_open, p.gap_ext, p.top_k, p.n_splits) == (o.gap_open, o.gap_ext, o.top_k, o.n_splits)
    assert (p.ungapped_min, p.xdrop, p.ext_right, p.ext_left) == (o.ungapped_min, o.xdrop, o.ext_right, o.ext_left) == (55, 12, 40, 24)
    assert p.stage1_min == o.stage1_min == 24
    for L in (1, 30, 100, 334, 1000, 3164, 50000):
        assert N.min_score(L) == O.min_score(L)
    assert N.min_score(334) == 68          # ~63/68/72 for 100/334/1000-aa queries
