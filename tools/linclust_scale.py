import sys, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
names, seqs = synth.make_genes(8000, 1002, seed=3)
base = np.stack([np.frombuffer(s[:1002], dtype=np.uint8) for s in seqs if len(s) >= 1002][:6000])
lut = np.full(256, 4, np.uint8); lut[[65, 67, 71, 84]] = (0, 1, 2, 3)
base = lut[base]
rng = np.random.default_rng(0)
idx = rng.integers(0, len(base), n)
arr = base[idx].copy()
mut = rng.random(arr.shape) < rng.choice([0.0, 0.005, 0.02, 0.06], size=(n, 1))
arr[mut] = rng.integers(0, 4, int(mut.sum()))
seqs = list(arr)
ctx = N.Context(0)
for D in (1.0, 0.95, 0.9):
    t0 = time.perf_counter()
    rep, st = ctx.linclust(seqs, D, 0.8)
    dt = time.perf_counter() - t0
    print('n=%d min_id=%.2f reps=%d %s  %.3f s  (%.2f M seq/s, %.1f Mnt/s)' % (n, D, len(np.unique(rep)), st, dt, n / dt / 1e6, n * 1002 / dt / 1e6))
