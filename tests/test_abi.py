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


def test_struct_sizes_match_header():
    from peppan_amd import _native as N
    assert N.HIT_DTYPE.itemsize == 64 and N.NT_HIT_DTYPE.itemsize == 40
    assert N.QUERY_META_DTYPE.itemsize == 16 and N.TARGET_META_DTYPE.itemsize == 16
    assert N.LOCUS_DTYPE.itemsize == 32
    assert C.sizeof(N.SearchParams) == 4 * 4 + 16 + 512 + 32 + 1024 + 16 + 8 + 16 + 32 + 16 + 8
    assert C.sizeof(N.Stats) == 16 * 8 + 6 * 8


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
    assert (p.ungapped_min, p.xdrop, p.ext_right, p.ext_left) == (o.ungapped_min, o.xdrop, o.ext_right, o.ext_left) == (45, 12, 40, 24)
    for L in (1, 30, 100, 334, 1000, 3164, 50000):
        assert N.min_score(L) == O.min_score(L)
    assert N.min_score(334) == 68          # ~63/68/72 for 100/334/1000-aa queries
