"""Device and host memory over many searches of changing size (sets re-uploaded, K1 inside, results kept and dropped): nothing may grow once
every buffer has seen its largest use.  python3 tools/soak_memory.py [rounds]"""
import os, sys, resource
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from peppan_amd import _native as N, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sets = []
for n, seed in ((3000, 1), (12000, 2), (500, 3), (8000, 4)):
    names, seqs = synth.make_genes(n, 0, seed=seed)
    sets.append(seqs)
ctx = N.Context(0)
p = N.default_params(45., 25., 10, 5)
free = []
keep = []
for r in range(rounds):
    s = sets[r % len(sets)]
    q = s[:len(s) // (1 + r % 3)]
    ctx.set_query_nt(q, 11); ctx.set_ref_nt(s, 6 if r % 2 else 3, 11)
    h, c, st = ctx.search(p, copy=bool(r % 2))
    if r % 5 == 0:
        keep = [h, c]
    ctx.invalidate_translation()
    ctx.search(p, copy=False)
    if r % 20 == 19:
        f, t = torch.cuda.mem_get_info(0)
        free.append(f)
        print('round %4d: device memory in use %.1f MiB, host RSS %.1f MiB, %d hits' % (r + 1, (t - f) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, len(h)), flush=True)
grow = (free[2] - free[-1]) / 2**20 if len(free) > 3 else 0.
print('device memory growth after the warm-up rounds: %.1f MiB' % grow)
sys.exit(1 if grow > 64 else 0)
