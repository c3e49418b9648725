"""round 5: what a get_map_bsn call through the worker pool spends outside its rounds (2 000 genomes: a quarter of the call): python tools/r5_tail.py [genomes] [workers]"""
import sys, time, argparse, zipfile
sys.path.insert(0, '.')
import bench
from peppan_amd import mapbsn, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spent = {}


def timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            spent.setdefault(label, []).append(round(time.perf_counter() - t, 3))
    setattr(obj, name, g)


timed(mapbsn.MapBsn, 'close', 'MapBsn.close')
timed(mapbsn.MapBsn, '_flush', 'MapBsn._flush')
timed(zipfile.ZipFile, 'close', 'ZipFile.close')
timed(zipfile.ZipFile, '_write_end_record', 'ZipFile._write_end_record')
timed(mapbsn._StoreWriter, 'close', '_StoreWriter.close')
timed(mapbsn.MapBsn, 'update_table', 'MapBsn.update_table')
args = argparse.Namespace(genes=10000, warmup=0, presence=None)
r = bench.map_strong(args, 0, 1, 0, n, workers=w, warm=True)
print('%d genomes, %d workers: %.2f s = %.1f genomes/s' % (n, w, r['seconds'], n / r['seconds']), {k: round(v, 2) for k, v in r['phase_s_rank0'].items()})
for k, v in spent.items():
    print('  %-28s %s' % (k, v[-12:]))
