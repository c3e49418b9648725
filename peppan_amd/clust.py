"""Exemplar selection by iterated linear-time clustering - the interface of PEPPAN's modules/clust.py on the MI355X.

    clust(argv)                      (reference entry point clust.py:21)   flags -i -p -d -c -t -a
    getClust(prefix, genes, params)  (reference entry point clust.py:34)   params: identity, coverage, n_thread, translate
        -> <prefix>.clust.exemplar   FASTA of the exemplars: the input's own text blocks, input order
           <prefix>.clust.tab        one "gene<TAB>exemplar" line per clustered gene, ordered by gene name as a string

What the reference's three `mmseqs` calls (clust.py:62-66) hand to the rest of that function is a relation
"representative, member" over the sequences of one FASTA file.  Here that relation comes from `cluster_relation`
(the GPU clusterer, peppan_amd.linclust / pep_linclust) or from any callable passed as params['cluster_fn'] - the
golden tests replay scripted mmseqs output through this module that way (tests/golden G9).

The module works on an in-memory model of the FASTA file (`Block` = one record's name and its verbatim text) and on a
forest of "is represented by" links; files are written once per round only because the clusterer reads a file.
Behaviour kept from the reference because downstream results depend on it:
  * at most three rounds, each re-clustering the previous round's exemplars; a round that does not shrink the
    exemplar set ends the loop (its links are still applied);
  * the exemplar of a cluster is its FIRST member in file order, not the clusterer's representative;
  * a record the clusterer did not mention is dropped from the next round;
  * the final table resolves chains gene -> exemplar -> exemplar of a later round.
"""
import argparse
import collections
import os
import shutil
import sys
import tempfile

from .configure import logger, transeq, uopen

Block = collections.namedtuple('Block', 'name text')          # text: the header line and every following line, verbatim
MAX_ROUNDS = 3


def _read_text(path):
    with uopen(path) as fin:
        return fin.read()


def blocks_of(data):
    """the records of FASTA text as verbatim text blocks, file order; lines before the first header belong to nobody.
    (One split at the header marks: the per-line loop this replaces was a third of iterClust's time at 300 k genes.)"""
    if data[:1] != '>':
        at = data.find('\n>')
        if at < 0:
            return []
        data = data[at + 1:]
    parts = data[1:].split('\n>')
    last = len(parts) - 1
    blocks = []
    for k, part in enumerate(parts):
        nl = part.find('\n')
        head = part if nl < 0 else part[:nl]
        blocks.append(Block(head.strip().split()[0], '>' + part + ('\n' if k < last else '')))
    return blocks


def read_blocks(path):
    return blocks_of(_read_text(path))


def sequence_of(blk):
    """a record's sequence: its non-comment lines without blanks, upper case"""
    nl = blk.text.find('\n')
    body = '' if nl < 0 else blk.text[nl + 1:]
    if '#' in body:
        body = ' '.join(line for line in body.split('\n') if not line.startswith('#'))
    return ''.join(body.split()).upper()


def readFasta(fasta):
    """[[name, SEQUENCE], ...] in file order"""
    return [[blk.name, sequence_of(blk)] for blk in read_blocks(fasta)]


def cluster_relation(fasta, identity, coverage, n_thread=1):
    """default clusterer: k-mer centres + ungapped / gapped verification on the GPU (K9); -> [(representative, member), ...]"""
    from . import linclust
    return linclust.linclust_file(fasta, identity, coverage)


def _one_round(blocks, link):
    """exemplar choice of one round.  link: gene -> cluster label for every gene seen so far.
    Returns (surviving blocks, {cluster label: exemplar name}): the survivor of a cluster is its first block in file order;
    blocks without a label do not survive."""
    first_of = {}
    survivors = []
    for blk in blocks:
        label = link.get(blk.name)
        if label is not None and label not in first_of:
            first_of[label] = blk.name
            survivors.append(blk)
    return survivors, first_of


def _resolve(link):
    """every gene's final exemplar: follow the links until a gene that represents itself (iterative, with memo)"""
    final = {}
    for gene in link:
        trail, g = [], gene
        while g not in final and link[g] != g:
            trail.append(g)
            g = link[g]
        root = final.get(g, g)
        for t in trail:
            final[t] = root
        final.setdefault(g, root)
    return final


def getClust(prefix, genes, params):
    cluster_fn = params.get('cluster_fn') or cluster_relation
    exemplar_path, tab_path = '{0}.clust.exemplar'.format(prefix), '{0}.clust.tab'.format(prefix)
    link = {}                                              # gene -> label, then -> exemplar of its cluster
    work = tempfile.mkdtemp(prefix='NS_', dir='.')
    try:
        nucleotides = None
        round_input = genes
        if params['translate']:                            # -a: cluster the first-frame proteins, report the nucleotide records
            nucleotides = readFasta(genes)
            round_input = os.path.join(work, 'seq.aa')
            with open(round_input, 'w') as fout:
                fout.writelines('>{0}\n{1}\n'.format(n, frames[0]) for n, frames in transeq(nucleotides, frame='1', transl_table='starts'))
        survivors_path = os.path.join(work, 'seq.ref')
        n_before = None
        for _ in range(MAX_ROUNDS):
            if cluster_fn is cluster_relation:             # the default clusterer takes the text this function reads anyway: one read per round
                from . import linclust
                text = _read_text(round_input)
                blocks = blocks_of(text)
                relation = linclust.linclust_text(text, blocks, params['identity'], params['coverage'])
                del text
            else:
                relation = cluster_fn(round_input, params['identity'], params['coverage'], params['n_thread'])
                blocks = read_blocks(round_input)
            link.update((str(member), str(rep)) for rep, member in relation)
            survivors, first_of = _one_round(blocks, link)
            for gene, label in link.items():               # every gene seen so far follows its cluster's exemplar
                link[gene] = first_of.get(label, label)
            with open(survivors_path, 'w') as fout:
                fout.writelines(blk.text for blk in survivors)
            if n_before is not None and len(first_of) >= n_before:
                break
            n_before, round_input = len(first_of), survivors_path
        if nucleotides is None:
            shutil.copy2(survivors_path, exemplar_path)
        else:
            by_name = dict(nucleotides)
            with open(exemplar_path, 'w') as fout:
                fout.writelines('>{0}\n{1}\n'.format(blk.name, by_name[blk.name]) for blk in read_blocks(survivors_path))
    finally:
        shutil.rmtree(work)
    final = _resolve(link)
    with open(tab_path, 'w') as fout:
        fout.writelines('{0}\t{1}\n'.format(gene, final[gene]) for gene in sorted(link))
    return exemplar_path, tab_path


def clust(argv):
    ap = argparse.ArgumentParser(description='Cluster gene sequences on an MI355X and pick one exemplar per cluster.')
    ap.add_argument('-i', '--input', required=True, help='FASTA file with the gene sequences')
    ap.add_argument('-p', '--prefix', required=True, help='the two outputs are written to <prefix>.clust.exemplar and <prefix>.clust.tab')
    ap.add_argument('-d', '--identity', type=float, default=0.9, help='lowest identity of a member to its cluster centre (0.9)')
    ap.add_argument('-c', '--coverage', type=float, default=0.9, help='lowest alignment coverage of member and centre (0.9)')
    ap.add_argument('-t', '--n_thread', type=int, default=8, help='kept for command-line compatibility; the clustering runs on the GPU')
    ap.add_argument('-a', '--translate', action='store_true', default=False, help='cluster the translated sequences instead of the nucleotides')
    args = ap.parse_args(argv)
    exemplar, tab = getClust(args.prefix, args.input, vars(args))
    logger('Exemplar sequences in {0}'.format(exemplar))
    logger('Clusters in {0}'.format(tab))
    return exemplar, tab


if __name__ == '__main__':
    clust(sys.argv[1:])
