"""Config-size inputs for the kernels outside the all-vs-all step, for rocprofv3 (tools/profile_other.sh):
   K7 k7_rescore + the nucleotide search : the reference's hot call on the 10k-gene FASTA (about 70 k hits rescored)
   K9 lc_*                               : linclust on 200 000 genes x ~1 kb
   K13 k13_sha1 / k13_table              : sha1 + duplicate collapse of 300 000 genes
   K11 ovl_sweep, K12 k12_*              : get_map_bsn, 10 000 exemplars x 8 genomes
Prints the algorithmic byte counts of each stage (SURVEY.md 8d style) so that the profile can be turned into HBM fractions."""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import synth, uberBlast as UB, mapbsn

_start_dir = os.getcwd()
os.chdir(tempfile.mkdtemp())
names, seqs = synth.make_genes(10000, 1002, seed=355)
with open('ex.fa', 'w') as f:
    for a, b in zip(names, seqs):
        f.write('>%s\n%s\n' % (a, b.decode()))
argv = '-r ex.fa -q ex.fa --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 1 --min_ratio 0.25 -e 3,3 -p --gtable 11'.split()
with contextlib.redirect_stderr(io.StringIO()):
    for rep in range(2):
        t0 = time.perf_counter()
        tab = UB.uberBlast(argv)
        dt = time.perf_counter() - t0
aln = sum(int(r[3]) for r in tab)
print('K7: uberBlast %d rows in %.0f ms; rescoring reads 2 x aligned nt + 40 B per hit = %.1f MB (before the identity cut: a few %% more)' % (len(tab), dt * 1e3, (2 * aln + 40 * len(tab)) / 1e6))

ctx = UB.get_context()
g200, s200 = synth.make_genes(200000, 0, seed=5)
from peppan_amd import linclust as LC
enc = [LC.encode(s.decode()) for s in s200]
for rep in range(2):
    t0 = time.perf_counter()
    rep_, st = ctx.linclust(enc, 0.9, 0.8)
    dt = time.perf_counter() - t0
tot = sum(len(x) for x in enc)
print('K9: linclust %d genes, %.1f Mnt, %.2f s; %d representatives; selected %d k-mers, verified %d pairs: L + 20 x 16 B per sequence = %.1f MB'
      % (len(enc), tot / 1e6, dt, len(set(rep_.tolist())), st['selected'], st['verified'], (tot + 320 * len(enc)) / 1e6))

inst = synth.make_instances(1200, 250, seed=8)
for rep in range(2):
    t0 = time.perf_counter()
    dig = ctx.sha1(inst)
    t1 = time.perf_counter()
    rep_ = ctx.dedup(np.array([len(s) for s in inst], dtype=np.uint32), dig)
    t2 = time.perf_counter()
print('K13: sha1 of %d genes (%.1f MB) %.0f ms incl. upload, dedup %.0f ms; 1 B per base + 20 B digest = %.1f MB; 28 B per gene in the collapse'
      % (len(inst), sum(map(len, inst)) / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (sum(map(len, inst)) + 20 * len(inst)) / 1e6))

gn, gs = synth.make_genes(10000, 0, seed=11)
with open('m.clust.exemplar', 'w') as f:
    for i, s in enumerate(gs):
        f.write('>%d\n%s\n' % (i, s.decode()))
worlds = synth.make_genomes(gs, 8, seed=5)
genomes = {}
with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
    for g, (gname, contig, ann) in enumerate(worlds):
        genomes[100000 + g] = [900000 + g, contig.decode()]
        op.save(100000 + g, np.array([[k, s, e, st, 1] for k, s, e, st in ann], dtype=object))
np.save('m.self_bsn.npy', np.array([[0, 1, 9000]], dtype=int))
params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
              match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
for rep in range(2):
    t0 = time.perf_counter()
    with contextlib.redirect_stderr(io.StringIO()):
        with mapbsn.MapBsn('t.npz', 'w') as c0, mapbsn.MapBsn('s.npz', 'w') as c1, mapbsn.MapBsn('m.npz', 'w') as c2, mapbsn.MapBsn('c.npz', 'w') as c3:
            mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', c0, c1, c2, c3, True, params)
    dt = time.perf_counter() - t0
print('K11/K12: get_map_bsn 8 genomes in %.2f s (%.1f genomes/s)' % (dt, 8 / dt))
os.chdir(_start_dir)          # rocprofv3 resolves its (relative) output directory when the process ends
