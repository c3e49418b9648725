#!/bin/bash
# AddressSanitizer + UBSan over the host-only code (GPU sanitizers are not available on the pool): the oracle and the C++ -f / -m filters.
# usage: tools/sanitize_host.sh      (CPU only; prints one line per check)
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
g++ -x c++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -shared -fPIC \
    peppan_amd/csrc/mapfilters.hip -o /tmp/libmf_asan.so
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
import test_host_golden as T
rng = np.random.default_rng(7)
ok = True
for tab in [T._dense_table(rng, 40, 3, k) for k in (2, 6, 15, 25)]:
    for cov, delta in ((0.9, 0.), (0.5, 10.)):
        ok &= mapfilters.ovl_filter(copy.deepcopy(tab), cov, delta).tolist() == mapfilters.ovl_filter_py(copy.deepcopy(tab), cov, delta).tolist()
    for gap, diff in ((600., 1.5), (2000., 3.0)):
        ok &= mapfilters.linear_merge(copy.deepcopy(tab), gap, diff).tolist() == mapfilters.linear_merge_py(copy.deepcopy(tab), gap, diff).tolist()
print('host C++ filters == Python statement under ASAN/UBSAN:', ok)
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
