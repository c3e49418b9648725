"""When and where the blocks of the seed matcher run (a PROBES build of the library: make PROBES=1, or PEPPAN_HIP_LIB=<that build>): start and end of every one of its 2 048
persistent blocks on the device's wall clock (100 MHz) and the compute unit it ran on.  python tools/ab/block_times.py [n_genes]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
names, seqs = synth.make_genes(n, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
seqs = [seqs[i] for i in order]
ctx = N.Context(0)
ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
p = N.default_params(45., 25., 10, 5)
p.n_shapes = 1                                    # (one launch of the matcher per search: the probe holds the last one)
lib = N.load_library()
for rep in range(4):
    ctx.invalidate_translation(); h, c, st = ctx.search(p, copy=False)
out = np.zeros(4 * 2048, dtype=np.uint64)
assert lib.pep_probe_block_times(out.ctypes.data_as(C.c_void_p)) == 0
t = out.reshape(2048, 4)
start, end, hw = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2]
t0 = start.min()
start, end = (start - t0) / 100., (end - t0) / 100.                 # us
xcc, hw_id = (hw >> np.uint64(32)).astype(np.int64) & 15, hw.astype(np.int64) & 0xFFFFFFFF
cu, sh, se = (hw_id >> 8) & 15, (hw_id >> 12) & 1, (hw_id >> 13) & 7
where = xcc * 10000 + se * 1000 + sh * 100 + cu
print('%d blocks: kernel %.0f us from the first start to the last end; starts: median %.1f us, 90 %% by %.1f us, last %.1f us; a block lives %.0f us (median; shortest %.0f, longest %.0f)' % (
    len(t), end.max(), np.median(start), np.percentile(start, 90), start.max(), np.median(end - start), (end - start).min(), (end - start).max()))
u, cnt = np.unique(where, return_counts=True)
print('compute units used: %d (by XCC: %s); blocks per unit: %s' % (len(u), np.bincount(xcc, minlength=8).tolist(), dict(zip(*np.unique(cnt, return_counts=True)))))
late = start > 0.25 * end.max()
print('blocks that start after a quarter of the kernel: %d' % late.sum())
for x in range(8):
    m = xcc == x
    if m.any():
        print('  XCC %d: %4d blocks, starts %.1f .. %.1f us, ends %.0f .. %.0f us, median life %.0f us' % (x, m.sum(), start[m].min(), start[m].max(), end[m].min(), end[m].max(), np.median((end - start)[m])))
