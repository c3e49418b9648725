"""cProfile of one step of `bench.py --workload map` (16 genomes x 10 000 exemplars: batched search of both tools, -f / -m / -O, K7, K12, build_bsn)"""
import os, sys, io, contextlib, cProfile, pstats, argparse, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import bench
from peppan_amd import mapbsn
args = argparse.Namespace(genes=10000, map_genomes=16)
calls = []
orig = mapbsn.build_bsn
pr = cProfile.Profile()
state = {'n': 0}
real_step_time = []
def run():
    return bench.map_workload(args, 0, 1, 0, 16, 2, 1)
# profile only the timed steps: enable after the warm-up step by wrapping _gpu_search
orig_search = mapbsn._gpu_search
def wrapped(*a, **k):
    state['n'] += 1
    if state['n'] == 2:
        pr.enable()
    return orig_search(*a, **k)
mapbsn._gpu_search = wrapped
r = run()
pr.disable()
print('genomes/s %.1f' % (r['genomes'] / r['seconds']))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(32)
print(s.getvalue())
