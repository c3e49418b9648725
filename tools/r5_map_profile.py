"""round 5: cProfile of the one-process mapping step of bench.py (10 000 exemplars x 16 genomes: search of both tools, filters, K7, K12, build_groups; no stores)
python tools/r5_map_profile.py [genomes] [genes]"""
import cProfile, io, pstats, sys
sys.path.insert(0, '.')
import bench
from peppan_amd import synth
n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 10000


class A(object):
    pass


A.genes = genes
pr = cProfile.Profile()
orig = bench.time.perf_counter
mr = bench.map_workload(A, 0, 1, 0, n_genomes, 1, 2)
pr.enable()
mr = bench.map_workload(A, 0, 1, 0, n_genomes, 3, 1)
pr.disable()
print('%.1f genomes/s (%d groups, %d rows per step)' % (mr['genomes'] / mr['seconds'], mr['groups_per_step'], mr['hit_rows_per_step']))
for key in ('cumulative', 'tottime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(34)
    print('\n'.join(l[:170] for l in s.getvalue().splitlines()[6:]))
for what in ('prod', 'shape_base.py:380', "'reduce' of 'numpy.ufunc'"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_callers(what)
    print('\n'.join(l[:170] for l in s.getvalue().splitlines()[:40]))
