"""get_map_bsn with worker processes on one GPU: genomes per second of a started pool (second pass over the same set) by number of workers.
usage: [GENES=50000 [PRESENCE=pan]] python tools/map_pool_rate.py [genomes] [workers ...]"""
import sys, argparse
sys.path.insert(0, '.')
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ws = [int(x) for x in sys.argv[2:]] or [4, 8, 16]          # (0 = one process, no pool: a single cold pass)
import os
from peppan_amd import synth
args = argparse.Namespace(genes=int(os.environ.get('GENES', 10000)), warmup=0,       # GENES=50000: BASELINE configs[4]'s exemplar count
                          presence=synth.PAN_GENOME_PRESENCE if os.environ.get('PRESENCE') == 'pan' else None)      # PRESENCE=pan: genomes of ~6 500 of the 50 000 genes (7.7 Mb)
for w in ws:
    r = bench.map_strong(args, 0, 1, 0, n, workers=w, warm=True)
    if 'container_cpu_s' in r:
        print('   container CPU over the timed pass: %.1f s = %.0f ms per mapped genome, %.1f CPUs busy on average, throttled in %d periods of 100 ms; the keeping process alone %.2f s' % (
            r['container_cpu_s'], r['container_cpu_s'] / n * 1e3, r['container_cpu_s'] / r['seconds'], r['throttled_periods'], r['keeper_process_cpu_s']), r.get('keeper_feeders_s'))
    if r.get('gpu_busy_frac') is not None:
        print('   GPU busy over the timed pass: %.0f %% (%d samples of gpu_busy_percent)' % (100 * r['gpu_busy_frac'], r['gpu_busy_samples']))
    print('workers %2d: %d genomes in %.2f s = %.1f genomes/s (first pass %.2f s, start-up %.2f s)  %s' % (
        w, n, r['seconds'], n / r['seconds'], r.get('first_pass_s', float('nan')), r.get('workers_startup_s', 0.), {k: round(v, 2) for k, v in r['phase_s_rank0'].items()}), flush=True)
