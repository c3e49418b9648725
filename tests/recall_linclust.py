"""What the clusterer delivers against an exhaustive pairwise pass (test infrastructure; the counterpart of tests/recall_report.py for K9).

`mmseqs linclust --min-seq-id D -c C` (clust.py:63) promises clusters whose members match their representative with identity >= D over
>= C of both sequences; PEPPAN runs it at 11 falling identity levels, each on the exemplars of the one before (iterClust, PEPPAN.py:1777-1792).
The build's clusterer is the exactly reproducible linclust variant oracle_linclust defines (oracle/align_oracle.c; the HIP kernels of K9
reproduce it bit for bit): 20 min-hash k-mers per sequence, ungapped then gapped verification against the k-mer's centre, greedy assignment.
A linear-time method that only compares sequences sharing a selected k-mer can miss pairs.  This module measures how many:

  truth(level)  = pairs of sequences that ENTER the level whose full-matrix optimal local alignment (oracle/full_sw.c: +2 / -3, gap 6 + 2k -
                  the scoring of the clusterer's gapped stage) has identity >= D and covers >= C of both sequences
  same cluster  = truth pairs whose two sequences leave the level under one exemplar                       -> recall
  left          = truth pairs whose two sequences BOTH leave the level as exemplars (redundancy the level did not remove)
  final recall  = pairs of INPUT sequences within the LAST level's thresholds whose members stand under one exemplar after the whole schedule
                  (what PEPPAN consumes: a level that leaves a pair apart - its k-mers' centre is a longer sequence just below the level's
                  identity, and only centres are compared - hands both to the next level, where that centre absorbs them)

    python tests/recall_linclust.py [synth|real] ...     one JSON object per workload (tracked: profiles/r05_recall_linclust.jsonl)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pair_table(codes, coverage, floor_identity, threads=0):
    """every pair (i < j) that can reach identity >= floor_identity over >= coverage of both sequences, with the identity and the two
    coverages of its full-matrix optimal alignment: int64[m, 2], float64[m, 3].  A qualifying alignment has at least coverage * max(Li, Lj)
    columns, 2 per identical column and at most 8 off per other column: its score is at least (2 f - 8 (1 - f)) * columns - only pairs whose
    full-matrix optimum reaches that are aligned."""
    from oracle import oracle as O, full_sw as F
    from peppan_amd import _native as N
    P = O.params_from(N.nucleotide_params(0., 0.))
    M = F.score_matrix(codes, codes, P, threads)
    L = np.array([len(c) for c in codes], dtype=np.float64)
    per_col = 2. * floor_identity - 8. * (1. - floor_identity)
    need = np.floor(per_col * coverage * np.maximum(L[:, None], L[None, :]))
    ii, jj = np.nonzero(np.triu(M >= need, 1))
    pairs, vals = [], []
    for i, j in zip(ii.tolist(), jj.tolist()):
        a, _ = F.align(codes[i], codes[j], P)
        pairs.append((i, j))
        vals.append((a.n_ident / float(a.aln_len), (a.q_end - a.q_start + 1) / L[i], (a.t_end - a.t_start + 1) / L[j]))
    return np.array(pairs, dtype=np.int64).reshape(-1, 2), np.array(vals, dtype=np.float64).reshape(-1, 3), float(L.sum()) ** 2


def report(seqs, identity=0.9, coverage=0.8, threads=0, cluster=None):
    """seqs: nucleotide strings in file order.  cluster(codes, identity, coverage) -> representative index per sequence (default: the oracle)."""
    from oracle import oracle as O
    from peppan_amd.pipeline import identity_schedule
    codes = [O.nt_codes(s) for s in seqs]
    cluster = cluster or (lambda c, d, cov: O.linclust(c, d, cov)[0])
    t0 = time.perf_counter()
    pairs, vals, cells = pair_table(codes, coverage, identity, threads)
    t_full = time.perf_counter() - t0
    current = np.arange(len(seqs))
    owner = np.arange(len(seqs))                              # the exemplar every input sequence stands under so far
    levels = []
    for d in identity_schedule(identity):
        rep_local = np.asarray(cluster([codes[i] for i in current.tolist()], d, coverage), dtype=np.int64)
        rep = np.full(len(seqs), -1, dtype=np.int64)
        rep[current] = current[rep_local]
        ok = (vals[:, 0] >= d) & (vals[:, 1] >= coverage) & (vals[:, 2] >= coverage) if len(vals) else np.zeros(0, bool)
        inside = ok & (rep[pairs[:, 0]] >= 0) & (rep[pairs[:, 1]] >= 0) if len(pairs) else ok
        a, b = pairs[inside, 0], pairs[inside, 1]
        same = int((rep[a] == rep[b]).sum())
        left = int(((rep[a] == a) & (rep[b] == b)).sum())
        exemplars = np.unique(rep[current])
        levels.append(dict(identity=d, sequences=int(len(current)), exemplars=int(len(exemplars)), truth_pairs=int(inside.sum()), same_cluster=same,
                           recall=round(same / float(inside.sum()), 4) if inside.sum() else None, truth_pairs_left_between_exemplars=left))
        owner = rep[owner]
        current = exemplars                                   # (ascending = file order: the next level's input, clust.py:71-92)
    tot = sum(l['truth_pairs'] for l in levels)
    # the schedule as a whole: pairs of INPUT sequences within the last level's thresholds that ended under one exemplar
    ok = (vals[:, 0] >= identity) & (vals[:, 1] >= coverage) & (vals[:, 2] >= coverage) if len(vals) else np.zeros(0, bool)
    together = int((owner[pairs[ok, 0]] == owner[pairs[ok, 1]]).sum()) if len(pairs) else 0
    return dict(sequences=len(seqs), identity=identity, coverage=coverage, full_cells=cells, full_seconds=round(t_full, 1), aligned_pairs=int(len(pairs)),
                levels=levels, recall_all_levels=round(sum(l['same_cluster'] for l in levels) / float(tot), 4) if tot else None,
                input_pairs_within_last_level=int(ok.sum()), input_pairs_under_one_final_exemplar=together,
                final_recall=round(together / float(ok.sum()), 4) if ok.sum() else None, final_exemplars=int(len(current)),
                truth_pairs_left_after_last_level=levels[-1]['truth_pairs_left_between_exemplars'])


def synth_instances(n_genes, copies, seed=8, indel_every=5):
    """alleles of a pan-genome as the front end meets them: synth.make_instances (substitution alleles) plus, for every indel_every-th gene, an
    allele with one codon deleted and one with a codon inserted - the alleles linclust's gapped stage exists for"""
    from peppan_amd import synth
    rng = np.random.default_rng(seed + 1)
    out = synth.make_instances(n_genes, copies, seed=seed)
    extra = []
    for g in range(0, n_genes, indel_every):
        s = out[g * copies]
        p = 3 * int(rng.integers(5, len(s) // 3 - 5))
        extra += [s[:p] + s[p + 3:], s[:p] + 'GCT' + s[p:]]
    order = rng.permutation(len(out) + len(extra))
    allseq = out + extra
    return [allseq[i] for i in order.tolist()]


def main(argv):
    import gzip
    for what in argv or ['synth', 'real']:
        if what == 'synth':
            seqs = synth_instances(300, 6, seed=8)
            rec = dict(workload='synthgenes-v1 instances: 300 genes (log-normal lengths) x 6 alleles + indel alleles, shuffled', **report(seqs))
        elif what == 'real':
            with gzip.open(os.path.join(ROOT, 'tests', 'golden', 'g16_real_genes.fa.gz'), 'rt') as f:
                seqs = [''.join(rec.split('\n')[1:]).upper() for rec in f.read().split('>')[1:]]
            rec = dict(workload='golden G16: %d real E. coli genes of the reference\'s examples/' % len(seqs), **report(seqs))
        else:
            raise SystemExit('unknown workload ' + what)
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    main(sys.argv[1:])
