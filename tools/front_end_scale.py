"""The front end at BASELINE configs[2] scale: writeGenes (K13) + iterClust (K9, 11 identity levels) on N gene instances of a synthetic
pan-genome (default 5 M = 10 000 genes x 500 genomes).  usage: python tools/front_end_scale.py [n_genes] [copies]"""
import contextlib, io, os, resource, sys, tempfile, time
sys.path.insert(0, '.')
from peppan_amd import pipeline as PL, synth, _native as N
n_base = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 500
os.chdir(tempfile.mkdtemp())
t0 = time.perf_counter()
seqs = synth.make_instances(n_base, copies, seed=8)
n = len(seqs)
t1 = time.perf_counter()
with N.Context(0) as ctx:
    hashes = PL.gene_hashes(seqs, ctx=ctx)
    t2 = time.perf_counter()
    genes = {i: ['f', '', 0, 0, '+', hashes[i], seqs[i]] for i in range(n)}
    prio = {i: [i % 7, -len(seqs[i]), hashes[i]] for i in range(n)}
    t2b = time.perf_counter()              # (the two dictionaries exist already in a PEPPAN run: building them is not writeGenes' time)
    prof = None
    if os.environ.get('FRONT_END_PROFILE'):
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    fn, groups = PL.writeGenes('big.genes', genes, prio, ctx=ctx)
    t3 = time.perf_counter()
    n_unique = sum(1 for line in open(fn) if line.startswith('>'))
    with contextlib.redirect_stderr(io.StringIO()):
        ex = PL.iterClust('big', fn, groups, dict(identity=0.9, coverage=0.8, n_thread=1, translate=False))
    t4 = time.perf_counter()
    if prof is not None:
        import pstats
        prof.disable()
        sio = io.StringIO()
        pstats.Stats(prof, stream=sio).sort_stats('tottime').print_stats(28)
        print(sio.getvalue())
n_ex = sum(1 for line in open(ex) if line.startswith('>'))
print('%d instances (%.2f Gnt): generate %.1f s, sha1 on the GPU %.1f s, building the genes / priority dictionaries %.1f s, writeGenes %.1f s, iterClust %.1f s; %d unique, %d exemplars; peak RSS %.1f GB'
      % (n, sum(map(len, seqs)) / 1e9, t1 - t0, t2 - t1, t2b - t2, t3 - t2b, t4 - t3, n_unique, n_ex, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))
