"""BASELINE configs[4] as ONE recorded run on one GPU: 50 000 exemplar genes x 2 000 genomes of a 50 000-gene pan-genome through
get_map_bsn(workers=8) with the four stores written (PEPPAN.py:759-772, 907-989).
    python tools/config4_map.py [n_genes] [n_genomes] [workers] [sampled genome ids ...]
The checks are the at-size test's own (tests/test_gpu_config_size.py::_map_bsn_at_size): stores consistent (one row per group, ids dense, one
hit-row block per group), every planted allele found in its genome, the sampled genomes' tables equal to the same host code over the CPU
oracle row for row.  Printed beside them: seconds, genomes/s (the call includes the pool's start-up), peak resident memory of this process and
of its children, store sizes.  profiles/r06_map_50k_x_2000.txt is this tool's output.  A tool, not a test: the run holds 2 x 15 GB of genomes."""
import os, resource, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('PEPPAN_LOG', '0')
n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
n_genomes = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sample = tuple(int(x) for x in sys.argv[4:]) or (5, n_genomes - 3)


class Patch(object):
    """what the helper asks of pytest's monkeypatch"""
    def __init__(self):
        self.undo = []

    def chdir(self, path):
        os.chdir(str(path))

    def setattr(self, obj, name, value):
        self.undo.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)


import test_gpu_config_size as TS                     # noqa: E402
from peppan_amd import synth                          # noqa: E402
base = '/dev/shm' if os.path.isdir('/dev/shm') else None
tmp = tempfile.mkdtemp(prefix='config4_', dir=base)
t0 = time.perf_counter()
try:
    print('configs[4] mapping stage on one GPU: %d exemplar genes (log-normal lengths) x %d genomes of a %d-gene pan-genome, get_map_bsn(workers=%d), four stores, in %s' % (
        n_genes, n_genomes, n_genes, workers, tmp), flush=True)
    TS._map_bsn_at_size(tmp, Patch(), n_genes, 0, n_genomes, synth.PAN_GENOME_PRESENCE if n_genes >= 20000 else None, sample,
                        5000 if n_genes >= 20000 else 1500, min_iden4=9000, workers=workers or None)
    print('checks passed: stores consistent, every planted allele found (identity >= 0.90), genomes %s equal to the oracle-driven host code row for row' % (sample,))
    for f in sorted(os.listdir(tmp)):
        if f.startswith('s.'):
            print('  store %-20s %8.1f MB' % (f, os.path.getsize(os.path.join(tmp, f)) / 1e6))
    print('peak resident memory: this process %.1f GB, largest child %.1f GB; whole tool %.0f s (genomes made, mapped, stores read back, samples through the CPU oracle)' % (
        resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6, time.perf_counter() - t0))
finally:
    os.chdir(ROOT)
    shutil.rmtree(tmp, ignore_errors=True)
