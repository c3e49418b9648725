"""A/B of the nucleotide tool's matcher on one box: seed_match_stride (default) against the plain 17-mer matcher (params.reserved[0] = 8).
    python tools/ab/stride_ab.py [n_genes] [genomes]
(1) the headline's genes against themselves, both strands; (2) the genes against `genomes` synthetic genomes (one mapping launch).
Prints phase times (HIP events, pep_set_timing 2) per variant, the raw hit counts, and whether the tables are byte-identical."""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import _native as N, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_genomes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
names, seqs = synth.make_genes(n, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
seqs = [seqs[i] for i in order]
ctx = N.Context(0)


def run(label, setup, reps=8):
    out = {}
    for flag in (0, 8, 0, 8):
        p = N.nucleotide_params(45., 25.)
        p.reserved[0] = flag
        setup()
        ctx.search(p)
        ctx.set_timing(2)
        acc, wall = {}, 0.
        for _ in range(reps):
            setup()
            t0 = time.perf_counter()
            h, c, st = ctx.search(p)
            wall += time.perf_counter() - t0
            for k in ('ms_seed_match', 'ms_seed', 'ms_sw', 'ms_sw_trace', 'ms_total'):
                acc[k] = acc.get(k, 0.) + st[k] / reps
        ctx.set_timing(0)
        print('%s  matcher=%s  seed_match %.3f ms  seed stage %.3f  sw %.3f  trace %.3f  total(gpu) %.3f  wall %.3f ms   raw hits %d  target seeds %d  candidates %d  hits %d' % (
            label, 'plain ' if flag else 'stride', acc['ms_seed_match'], acc['ms_seed'], acc['ms_sw'], acc['ms_sw_trace'], acc['ms_total'], wall / reps * 1e3,
            st['seed_hits'], st['target_seeds'], st['candidates'], len(h)))
        sys.stdout.flush()
        key = (np.array(h).tobytes(), np.array(c).tobytes(), st['seed_hits'], st['target_seeds'], st['candidates'])
        out.setdefault(flag, key)
        assert out[flag] == key
    print('%s  identical tables / raw hit counts / target seeds / candidates: %s' % (label, out[0] == out[8]))


ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11)
run('self %d genes' % n, lambda: ctx.use_nt_as_residues(2))
if n_genomes:
    contigs = [g[1] for g in synth.make_genomes(seqs, n_genomes, seed=11)]
    ctx.set_ref_nt(contigs, 6, 11)
    run('%d genes x %d genomes (%.1f Mnt)' % (n, n_genomes, sum(map(len, contigs)) / 1e6), lambda: ctx.use_nt_as_residues(2))
