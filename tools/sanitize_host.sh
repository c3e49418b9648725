#!/bin/bash
# AddressSanitizer + UBSan over the host-only code (GPU sanitizers are not available on the pool): the oracle, the C++ -f / -m filters and the
# store-member emitters (csrc/stores.hip).
# usage: tools/sanitize_host.sh      (CPU only; prints one line per check)
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
g++ -x c++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -shared -fPIC \
    peppan_amd/csrc/mapfilters.hip peppan_amd/csrc/stores.hip -o /tmp/libmf_asan.so -lz -lpthread
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python - <<'PY'
import sys, os, ctypes as C, copy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle as O
O.LIB = os.path.join(os.path.dirname(O.__file__), '_build', 'liboracle_asan.so'); O._lib = None
from peppan_amd import _native as N, mapfilters, synth
class Host:                       # only the two host entry points are needed
    def __init__(self): self.a = C.CDLL('/tmp/libmf_asan.so')
    def __getattr__(self, k): return getattr(self.a, k)
N._lib = Host()
import gzip, json, io
from conftest import GOLDEN
g = json.loads(gzip.open(os.path.join(GOLDEN, 'g18_filters_random.json.gz')).read().decode())
import test_host_golden as T
ok, n_tab = True, 0
for case in g['cases'][::4]:
    rows = [[q, r, iden, qe - qs + 1, 3, 0, qs, qe, ss, se, 0.0, score, ql, sl, [[qe - qs + 1, 'M']], i] for i, (q, r, iden, qs, qe, ss, se, score, ql, sl) in enumerate(case['cols'])]
    tab = T._table(rows)
    for key, want in case['ovl'].items():
        cov, delta = (float(x) for x in key.split('_'))
        ok &= [int(r[15]) for r in mapfilters.ovl_filter(copy.deepcopy(tab), cov, delta)] == want
    for key, want in case['merge'].items():
        gap, diff = (float(x) for x in key.split('_'))
        ok &= [int(r[15]) for r in mapfilters.linear_merge(copy.deepcopy(tab), gap, diff)] == [w[0] for w in want]
    n_tab += 1
print('host C++ filters == the reference on %d random tables of G18 under ASAN/UBSAN:' % n_tab, ok)
# the store-member emitters: pickle streams from columns, read back by numpy
rng = np.random.default_rng(7)
row_off = np.concatenate([[0], np.cumsum(rng.integers(0, 4, size=500))]); n = int(row_off[-1])
runs = rng.integers(0, 5, size=n); c_off = np.concatenate([[0], np.cumsum(runs)[:-1]])
arena = ((rng.integers(1, 70000, size=max(1, int(runs.sum()))) << 2) | rng.integers(0, 3, size=max(1, int(runs.sum())))).astype(np.uint32)
cols = [rng.integers(-2 ** 40, 2 ** 40, size=n) for _ in range(2)] + [rng.random(n)] + [rng.integers(0, 70000, size=n) for _ in range(7)] + [rng.random(n), rng.integers(0, 5000, size=n).astype(float)] + \
       [rng.integers(0, 300, size=n), rng.integers(0, 10 ** 7, size=n), arena, c_off, runs, rng.integers(0, 20000, size=n)]
lib = N._lib.a
lib.pep_store_mat_member.restype = lib.pep_store_seq_member.restype = C.c_int64
member = N.store_mat_member(cols, row_off, True)
back = np.lib.format.read_array(io.BytesIO(member), allow_pickle=True)
print('pep_store_mat_member under ASAN/UBSAN: %d bytes, %d groups read back,' % (len(member), len(back)), all(back[k].shape == (row_off[k + 1] - row_off[k], 16) for k in range(500)))
po = np.concatenate([[0], np.cumsum(rng.integers(0, 700, size=300))]); pk = rng.integers(0, 125, size=int(po[-1])).astype(np.uint8)
sb = np.lib.format.read_array(io.BytesIO(N.store_seq_member(pk, po)), allow_pickle=True)
print('pep_store_seq_member under ASAN/UBSAN:', all(np.array_equal(sb[k], pk[po[k]:po[k + 1]]) for k in range(300)))
# the gene table's members as finished zip entries (threads, zlib), sorted and through an order
import zipfile, tempfile
sizes = np.concatenate([rng.integers(1, 40, size=600), [900, 1, 3000]]); keys = np.sort(rng.choice(10 ** 6, size=len(sizes), replace=False))
tab = np.concatenate([np.column_stack([np.full(k, key), rng.integers(-10 ** 12, 10 ** 12, size=[k, 6])]) for key, k in zip(keys, sizes)]).astype(np.int64)
shuffle = rng.permutation(len(tab)); order = np.argsort(tab[shuffle, 0], kind='stable')
from peppan_amd import mapbsn
d = tempfile.mkdtemp()
with mapbsn.MapBsn(d + '/a.npz', 'w') as a, mapbsn.MapBsn(d + '/b.npz', 'w') as b:
    a.update_table(tab); b.update_table(tab[shuffle], order=order)
za, zb = dict(np.load(d + '/a.npz')), dict(np.load(d + '/b.npz'))
print('pep_store_tab_members under ASAN/UBSAN: %d members,' % len(za), sorted(za) == sorted(zb) == sorted(str(k) for k in keys) and all(np.array_equal(np.sort(za[k], axis=0), np.sort(zb[k], axis=0)) and len(za[k]) == n for k, n in zip(map(str, keys), sizes)),
      zipfile.ZipFile(d + '/a.npz').testzip() is None)
import zlib
lib.pep_deflate_literals.restype = C.c_int64
ok = True
for d in (b'', b'x', bytes(9000), os.urandom(300000), bytes(rng.integers(0, 125, size=400001).astype(np.uint8)), bytes(np.minimum(rng.geometric(0.002, size=300000), 255).astype(np.uint8))):
    ok &= zlib.decompress(N.deflate_literals(d), -15) == d
print('pep_deflate_literals under ASAN/UBSAN: inflates to the input,', ok)
prots = synth.make_proteins(120, length=(40, 400), seed=5, family=3, sub=0.2)
for mode in (0, 1):
    p = O.default_params(30., 20., 3, 5); p.hsp_mode = mode
    print('oracle search, hsp_mode', mode, len(O.search(prots[:50], prots, p)[0]), 'hits')
fam = []
for f in range(40):
    c = rng.integers(0, 4, int(rng.integers(300, 2500))).astype(np.uint8); fam.append(c)
    for v in range(3):
        x = c.copy(); cut = int(rng.integers(20, len(x) - 40))
        fam.append(np.concatenate([x[:cut], x[cut + int(rng.integers(1, 13)):]]))
print('oracle linclust (gapped verification)', O.linclust(fam + [np.zeros(0, np.uint8)], 0.95, 0.9)[1])
names, seqs = synth.make_genes(60, 0, seed=3)
contig = b'ACGT' * 50 + b''.join(seqs[:20])
rows = np.zeros(3, dtype=O.LOCUS_DTYPE); rows['q_start'] = 1; rows['rs'] = [201, 201 + len(seqs[0]), 300]; rows['re'] = [200 + len(seqs[0]), 202 + len(seqs[0]) - 1 + len(seqs[1]) - 1, 201]
rows['re'][1] = rows['rs'][1] + len(seqs[1]) - 1; rows['cigar_runs'] = 1; rows['cigar_off'] = [0, 1, 2]; rows['group'] = [0, 1, 2]
cig = np.array([(len(seqs[0]) << 2), (len(seqs[1]) << 2), (100 << 2)], dtype=np.uint32)
print('oracle alleles', [len(x) for x in O.alleles([contig], rows, cig, [0, 1, 2, 3], [len(seqs[0]), len(seqs[1]), 100])])
print('sanitizer run finished without reports')
PY
