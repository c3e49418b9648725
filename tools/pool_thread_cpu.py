"""Where the CPU seconds of the mapping pool go, thread by thread: every worker prints user + system seconds of each of its threads (its three Python
threads by role, the HIP runtime's by their kernel names) when it stops, this process the same for itself.  usage: python tools/pool_thread_cpu.py [genomes] [workers]"""
import sys, os, argparse
sys.path.insert(0, '.')
os.environ['PEPPAN_WORKERS_TIMING'] = '1'
os.environ.setdefault('PEPPAN_LOG', '0')
import bench
from peppan_amd.mapworkers import _thread_cpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = int(sys.argv[2]) if len(sys.argv) > 2 else 8
args = argparse.Namespace(genes=int(os.environ.get('GENES', 10000)), warmup=0, presence=None)
r = bench.map_strong(args, 0, 1, 0, n, workers=w, warm=True)
print('workers %d: %d genomes in %.2f s = %.1f genomes/s; container CPU over the timed pass %.1f s = %.1f ms per genome (both passes of every worker are in the lines above: %d genomes each pass)' % (
    w, n, r['seconds'], n / r['seconds'], r.get('container_cpu_s', float('nan')), r.get('container_cpu_s', float('nan')) / n * 1e3, n))
print('keeping process, CPU seconds by thread (both passes + the synthetic genomes):', ' '.join('%s %.2f' % kv for kv in _thread_cpu() if kv[1] >= 0.02))
