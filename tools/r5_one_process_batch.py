import sys
sys.path.insert(0, '.')
import os, argparse
import bench
from peppan_amd import mapbsn, synth
for b in (4, 8, 16, 32):
    mapbsn.ONE_PROCESS_BATCH = b
    args = argparse.Namespace(genes=10000, warmup=0, presence=None)
    r = bench.map_strong(args, 0, 1, 0, 256, workers=0, warm=True)
    print('one process, batches of %2d: 256 genomes in %.2f s = %.1f genomes/s' % (b, r['seconds'], 256 / r['seconds']), flush=True)
