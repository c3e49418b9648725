"""Genome-mapping callers and store formats (peppan_amd/mapbsn.py) against vectors produced by the reference's own
iter_map_bsn / compare_prediction / get_map_bsn / MapBsn (tests/golden/make_golden.py g14_g15; PEPPAN.py:27-114, 759-989)."""
import copy
import os

import numpy as np
import pytest

from conftest import load_golden
from peppan_amd import mapbsn
from oracle_context import OracleContext


def obj_table(rows):
    tab = np.empty([len(rows), len(rows[0]) if rows else 17], dtype=object)
    for i, r in enumerate(rows):
        for j, v in enumerate(r):
            tab[i, j] = copy.deepcopy(v)
    return tab


def plain(x):
    if isinstance(x, np.ndarray):
        return [plain(v) for v in x.tolist()]
    if isinstance(x, (list, tuple)):
        return [plain(v) for v in x]
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    return x


@pytest.fixture()
def world(tmp_path):
    g = load_golden('g14_mapbsn.json')
    old_fn = str(tmp_path / 'm.old_prediction.npz')
    with mapbsn.MapBsn(old_fn, 'w') as op:
        for contig, rows in g['old_prediction'].items():
            op.save(contig, np.array(rows, dtype=object))
    bsn_fn = str(tmp_path / 'm.self_bsn.npy')
    np.save(bsn_fn, np.array(g['self_bsn'], dtype=int))
    return g, old_fn, bsn_fn


def test_store_roundtrip(tmp_path):
    fn = str(tmp_path / 's.npz')
    with mapbsn.MapBsn(fn, 'w') as c:
        c.save(5, np.arange(6).reshape(2, 3))
        c.save('x', np.array([[1, 'a'], [2, 'b']], dtype=object))
        assert c.exists(5) and c.size() == 2 and c.get(7) == [] and c[5].shape == (2, 3)
        c.save(5, np.array([[5, 1, 1]]))                       # overwrite keeps the other members
        assert c.get('x').tolist() == [[1, 'a'], [2, 'b']] and c.get(5).tolist() == [[5, 1, 1]]
        c.update([np.array([[5, 2, 2], [5, 3, 3]]), np.array([[9, 0, 0]])])
        assert c.get(5).tolist() == [[5, 1, 1], [5, 2, 2], [5, 3, 3]] and c.get(9).tolist() == [[9, 0, 0]]
        assert sorted(c.keys()) == ['5', '9', 'x']
        assert c.pop(9).tolist() == [[9, 0, 0]] and not c.exists(9)
    z = dict(np.load(fn, allow_pickle=True))                   # the reference reads stores this way (PEPPAN.py:1931)
    assert z['5'].tolist() == [[5, 1, 1], [5, 2, 2], [5, 3, 3]]


def test_decode_encode():
    g = load_golden('g14_mapbsn.json')
    seqs = np.array(g['decodeSeq_in'], dtype=np.uint8)
    dec = mapbsn.decodeSeq(seqs)
    assert dec.tolist() == g['decodeSeq_out']
    for row, packed in zip(dec, seqs):
        assert mapbsn.encodeSeq(row).tolist() == packed.tolist()
    assert mapbsn.encodeSeq(np.array([1, 2, 3, 4], dtype=np.uint8)).tolist() == [1 * 25 + 3 * 5, 2 * 25 + 4 * 5]


def test_compare_prediction(world):
    g, old_fn, _ = world
    for case in g['cases']:
        got = mapbsn.compare_prediction(obj_table(case['table']), old_fn)       # names still strings here, like the fixture call
        assert plain(got) == case['compare_prediction']


def test_build_bsn(world):
    g, old_fn, bsn_fn = world
    for case in g['cases']:
        seq = [[int(c), s] for c, s in case['contigs'].items()]
        bsn, ovl = mapbsn.build_bsn(obj_table(case['table']), np.array(case['overlap'], dtype=int).reshape(-1, 3), seq, bsn_fn, old_fn,
                                    dict(g['params']), ctx=OracleContext())
        assert plain(ovl) == case['ovl']
        assert plain(bsn) == case['bsn']
    empty, eo = mapbsn.build_bsn(np.empty([0, 17], dtype=object), np.zeros([0, 3], dtype=int), [], bsn_fn, old_fn, dict(g['params']), ctx=OracleContext())
    assert empty.shape == (0, 7) and eo.shape == (0, 3)


def test_map_argv():
    g = load_golden('g14_mapbsn.json')
    argv = mapbsn._map_argv('CL', dict(g['params']))
    assert argv[2:] == g['cases'][0]['argv']                  # everything after "-r GENOME -q CL"
    assert '--diamond' not in mapbsn._map_argv('CL', dict(g['params'], noDiamond=True))


@pytest.mark.parametrize('save_seq', [True, False])
def test_get_map_bsn(world, tmp_path, save_seq):
    g, old_fn, bsn_fn = world
    want = load_golden('g15_getmapbsn.json')
    genomes, canned = {}, []
    for case in g['cases']:
        for c, s in case['contigs'].items():
            genomes[int(c)] = [want['genomes'][c], s]
        canned.append((obj_table(case['table']), np.array(case['overlap'], dtype=int).reshape(-1, 3)))
    names = [str(tmp_path / ('mm.%s.npz' % x)) for x in ('tab', 'seq', 'mat', 'conflicts')]

    def search(prefix, clust, jobs, params):
        assert [j[0] for j in jobs] == [0, 1, 2]
        return iter(canned)
    with mapbsn.MapBsn(names[0], 'w') as c0, mapbsn.MapBsn(names[1], 'w') as c1, mapbsn.MapBsn(names[2], 'w') as c2, \
            mapbsn.MapBsn(names[3], 'w') as c3:
        mapbsn.get_map_bsn(str(tmp_path / 'm'), 'CL', genomes, bsn_fn, old_fn, c0, c1, c2, c3, save_seq, dict(g['params']), search=search, ctx=OracleContext())
    exp = want['stores']['saveSeq_%d' % save_seq]
    for x, fn in zip(('tab', 'seq', 'mat', 'conflicts'), names):
        with mapbsn.MapBsn(fn) as c:
            got = {k: plain(c.get(k)) for k in sorted(c.keys())}
        assert got == exp[x], x
