"""mapping steps of bench.py's map_50k leg for the profiler: 50 000 exemplar genes mapped onto 8 genomes of a 50 000-gene pan-genome, search + filters + K7 +
K12 + build_groups, no stores.  python tools/one_map_step.py [steps] [genes] [genomes]"""
import sys
sys.path.insert(0, '.')
import bench
from peppan_amd import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
n_genomes = int(sys.argv[3]) if len(sys.argv) > 3 else 8


class A(object):
    pass


A.genes = genes
names, seqs = synth.make_genes(genes, 1002, seed=355)
mr = bench.map_workload(A, 0, 1, 0, n_genomes, steps, 0, with_stores=False, gene_set=(names, seqs), presence=synth.PAN_GENOME_PRESENCE if genes >= 50000 else None)
print('steps %d: %.1f genomes/s, %d groups and %d hit rows per step' % (steps, mr['genomes'] / mr['seconds'], mr['groups_per_step'], mr['hit_rows_per_step']))
