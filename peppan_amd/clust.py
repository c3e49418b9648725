"""Drop-in for PEPPAN's modules/clust.py: exemplar selection by iterated linear-time clustering.

    clust(argv)                      clust.py:21-33    same flags -i -p -d -c -t -a
    getClust(prefix, genes, params)  clust.py:34-111   same arguments, same two output files:
        <prefix>.clust.exemplar   FASTA of the exemplars, original header lines, input order
        <prefix>.clust.tab        "gene<TAB>exemplar" per input gene, sorted by gene name (string order)

The three `mmseqs` calls of the reference (createdb / linclust / createtsv, clust.py:62-66) produce one thing the
rest consumes: the relation "representative, member" over the round's input.  Here it comes from
`cluster_relation` (GPU, peppan_amd.linclust) or from any callable given as params['cluster_fn'] - which is how the
golden tests replay scripted mmseqs output through this file's logic.
"""
import argparse
import os
import shutil
import sys
import tempfile

from .configure import logger, transeq, uopen


def readFasta(fasta):
    """[[name, SEQUENCE], ...] in file order (the dict version lives in configure.py)"""
    out = []
    with uopen(fasta) as fin:
        for line in fin:
            if line.startswith('>'):
                out.append([line[1:].strip().split()[0], []])
            elif len(line) > 0 and not line.startswith('#'):
                out[-1][1].extend(line.strip().split())
    for rec in out:
        rec[1] = ''.join(rec[1]).upper()
    return out


def cluster_relation(fasta, identity, coverage, n_thread=1):
    """default clusterer: linear-time k-mer grouping + verification on the MI355X; returns [(rep, member), ...]"""
    from . import linclust
    return linclust.linclust_file(fasta, identity, coverage)


def _first_of_each_group(gene_file, groups):
    """exemplar = first sequence of each group in FILE order (clust.py:71-85); returns (kept lines, {group: exemplar})"""
    kept, chosen = [], {None: 1}
    with open(gene_file) as fin:
        writing = False
        for line in fin:
            if line.startswith('>'):
                name = line[1:].strip().split()[0]
                grp = groups.get(name, None)
                writing = grp not in chosen
                if writing:
                    chosen[grp] = name
            if writing:
                kept.append(line)
    return kept, chosen


def getClust(prefix, genes, params):
    cluster_fn = params.get('cluster_fn') or cluster_relation
    groups = {}
    work = tempfile.mkdtemp(prefix='NS_', dir='.')
    try:
        if not params['translate']:
            gene_file = genes
        else:
            na_seqs = readFasta(genes)
            gene_file = os.path.join(work, 'seq.aa')
            with open(gene_file, 'w') as fout:
                for n, s in transeq(na_seqs, frame='1', transl_table='starts'):
                    fout.write('>{0}\n{1}\n'.format(n, s[0]))
        ref_file = os.path.join(work, 'seq.ref')
        n_ref = 999999999999999
        for _ in range(3):
            for rep, member in cluster_fn(gene_file, params['identity'], params['coverage'], params['n_thread']):
                groups[str(member)] = str(rep)
            kept, chosen = _first_of_each_group(gene_file, groups)
            for gene, grp in groups.items():
                if grp in chosen:
                    groups[gene] = chosen[grp]
            with open(ref_file, 'w') as fout:          # `kept` is complete, so overwriting the round's own input is safe
                fout.writelines(kept)
            if n_ref <= len(chosen):
                break
            n_ref = len(chosen)
            gene_file = ref_file
        if not params['translate']:
            shutil.copy2(ref_file, '{0}.clust.exemplar'.format(prefix))
        else:
            na = dict(na_seqs)
            with open('{0}.clust.exemplar'.format(prefix), 'w') as fout:
                for n, _ in readFasta(ref_file):
                    fout.write('>{0}\n{1}\n'.format(n, na[n]))
    finally:
        shutil.rmtree(work)
    with open('{0}.clust.tab'.format(prefix), 'w') as fout:
        for gene, grp in sorted(groups.items()):
            g = gene
            while g != grp:                      # follow exemplar -> exemplar chains of later rounds
                g, grp = grp, groups[grp]
            groups[gene] = grp
            fout.write('{0}\t{1}\n'.format(gene, grp))
    return '{0}.clust.exemplar'.format(prefix), '{0}.clust.tab'.format(prefix)


def clust(argv):
    parser = argparse.ArgumentParser(description='Get clusters and exemplars of clusters from gene sequences (MI355X linear-time clustering).')
    parser.add_argument('-i', '--input', help='[INPUT; REQUIRED] name of the file containing gene sequneces in FASTA format.', required=True)
    parser.add_argument('-p', '--prefix', help='[OUTPUT; REQUIRED] prefix of the outputs.', required=True)
    parser.add_argument('-d', '--identity', help='[PARAM; DEFAULT: 0.9] minimum intra-cluster identity.', default=0.9, type=float)
    parser.add_argument('-c', '--coverage', help='[PARAM; DEFAULT: 0.9] minimum intra-cluster coverage.', default=0.9, type=float)
    parser.add_argument('-t', '--n_thread', help='[PARAM; DEFAULT: 8]   accepted for compatibility.', default=8, type=int)
    parser.add_argument('-a', '--translate', help='[PARAM; DEFAULT: False] activate to cluster in translated sequence.', default=False, action='store_true')
    args = parser.parse_args(argv)
    exemplar, tab = getClust(args.prefix, args.input, args.__dict__)
    logger('Exemplar sequences in {0}'.format(exemplar))
    logger('Clusters in {0}'.format(tab))
    return exemplar, tab


if __name__ == '__main__':
    clust(sys.argv[1:])
