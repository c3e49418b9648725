"""Optimality and recall of the seed-and-band heuristic against exhaustive Smith-Waterman (test infrastructure).

The reference's calls promise every target that meets the thresholds: `diamond blastp --id --query-cover --evalue 1`
(uberBlast.py:550) and `blastn -word_size 17 -perc_identity -qcov_hsp_perc -evalue 1e-2` (uberBlast.py:294).  The build's
search is a heuristic defined by oracle/align_oracle.c (and reproduced bit for bit by the HIP kernels); this module
measures what that heuristic loses against oracle/full_sw.c, an independent full-matrix Gotoh alignment:

  truth   = (query, target) pairs whose FULL-matrix optimum reaches the e-value score and whose full-matrix alignment
            passes the identity and query-cover cuts
  found   = pairs the heuristic reports under the same cuts (top-k switched off, so that ranking does not interfere)
  recall  = |truth & found| / |truth|, per identity bin of the full-matrix alignment
  optimal = reported pairs whose score equals the full-matrix optimum (a lower score = the 128-diagonal band, or the
            choice of the band, lost part of the alignment)

    python tests/recall_report.py [protein1k|protein10k_sample|nucl1k|real] ...   prints one JSON object per workload
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BINS = ((0.9, 1.01, '>=0.9'), (0.7, 0.9, '0.7-0.9'), (0.45, 0.7, '0.45-0.7'), (0.0, 0.45, '<0.45'))


def _bin_of(ident):
    for lo, hi, name in BINS:
        if lo <= ident < hi:
            return name
    return BINS[-1][2]


def protein_sets(nt_seqs, table=11):
    """queries (best forward frame) and targets (6 frames, cut at stops) exactly as runDiamond builds them (uberBlast.py:525-544)"""
    from oracle import oracle as O
    q_aa = [O.aa_codes(O.query_frame(s, table)[1]) for s in nt_seqs]
    t_aa, t_gene = [], []
    for i, s in enumerate(nt_seqs):
        for aa in O.translate_frames(s, range(1, 7), table):
            for off, c in O.ref_chunks(aa):
                t_aa.append(O.aa_codes(c))
                t_gene.append(i)
    return q_aa, t_aa, np.array(t_gene)


def report(q_seqs, t_seqs, params, min_scores, q_idx=None, threads=0, sensitive_shapes=None):
    """q_seqs / t_seqs: residue-code arrays; params: oracle Params with the cuts set; q_idx: the queries to evaluate (all targets
    are always searched).  Returns a dict of counts (see module docstring)."""
    from oracle import oracle as O, full_sw as F
    import copy
    q_idx = list(range(len(q_seqs))) if q_idx is None else list(q_idx)
    qs = [q_seqs[i] for i in q_idx]
    ms = np.asarray(min_scores, dtype=np.int32)[q_idx]
    t0 = time.perf_counter()
    M = F.score_matrix(qs, t_seqs, params, threads)
    t_full = time.perf_counter() - t0
    cells = float(sum(len(x) for x in qs)) * float(sum(len(x) for x in t_seqs))
    # the heuristic with ranking switched off: one split, unbounded k
    p = copy.copy(params)
    p.top_k, p.n_splits = 1 << 30, 1
    h, cig, st = O.search(qs, t_seqs, p, min_scores=ms)
    found = {}
    for k in range(len(h)):
        key = (int(h['q'][k]), int(h['t'][k]))
        if key not in found or h['score'][k] > found[key]:
            found[key] = int(h['score'][k])
    # the SENSITIVE mode of the translated search (four seed shapes instead of two: pep_set_sensitivity level 1), same cuts
    found_s, st_s = None, None
    if getattr(params, 'base', 0) != 4 and sensitive_shapes:
        from peppan_amd import _native as N
        ps = copy.copy(p)
        N.set_shapes(ps, list(sensitive_shapes))
        hs, _, st_s = O.search(qs, t_seqs, ps, min_scores=ms)
        found_s = set(zip(hs['q'].tolist(), hs['t'].tolist()))
    # the same without the ungapped pre-filter: separates what seeding loses from what the filter loses
    p0 = copy.copy(p)
    p0.ungapped_min = 0
    h0, _, st0 = O.search(qs, t_seqs, p0, min_scores=ms)
    found0 = set(zip(h0['q'].tolist(), h0['t'].tolist()))
    out = dict(queries=len(qs), targets=len(t_seqs), full_cells=cells, full_seconds=round(t_full, 2), reported_pairs=len(found),
               candidates=st['candidates'], bins={name: dict(truth=0, found=0) for _, _, name in BINS[:3]})
    # (i) optimality of what is reported
    sub = [k for k in found if found[k] != int(M[k[0], k[1]])]
    out['reported_below_optimum'] = len(sub)
    out['reported_above_optimum'] = sum(1 for k in sub if found[k] > int(M[k[0], k[1]]))      # must be 0: nothing beats the full matrix
    out['score_loss_of_those'] = sorted(int(M[k[0], k[1]]) - found[k] for k in sub)[-5:]
    # (ii) recall: full-matrix alignment of every pair above the score threshold
    truth, missed = set(), []
    for qi, ti in np.argwhere(M >= ms[:, None]):
        a, _ = F.align(qs[qi], t_seqs[ti], params)
        assert a.score == M[qi, ti], (a.score, M[qi, ti])                   # the vectorised score pass == the textbook matrix
        ident = a.n_ident / float(a.aln_len)
        idp = a.n_ident * 100.0 / a.aln_len
        qcov = (a.q_end - a.q_start + 1) * 100.0 / len(qs[qi])
        if idp >= params.min_id_pct and qcov >= params.min_qcov_pct:
            key = (int(qi), int(ti))
            truth.add(key)
            b = out['bins'][_bin_of(ident)]
            b['truth'] += 1
            if found_s is not None and key in found_s:
                b['found_sensitive'] = b.get('found_sensitive', 0) + 1
            if key in found:
                b['found'] += 1
            else:
                missed.append((round(ident, 3), int(a.score), int(ms[qi]), len(qs[qi]), len(t_seqs[ti]), int(a.aln_len)))
    out['truth_pairs'] = len(truth)
    out['lost_to_seeding'] = sum(1 for k in truth if k not in found0)             # no shared seed word (or no band reaching the cuts)
    out['lost_to_ungapped_filter'] = sum(1 for k in truth if k in found0 and k not in found)
    out['candidates_without_filter'] = st0['candidates']
    out['found_of_truth'] = sum(1 for k in truth if k in found)
    out['reported_not_in_truth'] = sum(1 for k in found if k not in truth)   # e.g. a band-limited alignment that passes a cut its full version fails
    for b in out['bins'].values():
        b['recall'] = round(b['found'] / b['truth'], 4) if b['truth'] else None
        if found_s is not None:
            b['found_sensitive'] = b.get('found_sensitive', 0)
            b['recall_sensitive'] = round(b['found_sensitive'] / b['truth'], 4) if b['truth'] else None
    if found_s is not None:
        out['sensitive'] = dict(shapes=list(sensitive_shapes), reported_pairs=len(found_s), candidates=st_s['candidates'], found_of_truth=sum(1 for k in truth if k in found_s),
                                recall=round(sum(1 for k in truth if k in found_s) / max(1, len(truth)), 4))
    out['recall'] = round(out['found_of_truth'] / max(1, len(truth)), 4)
    out['missed_examples(ident,score,min_score,Lq,Lt,cols)'] = sorted(missed, reverse=True)[:8]
    return out


def workload(name):
    """-> (q_seqs, t_seqs, params, min_scores, q_idx, description)"""
    from oracle import oracle as O
    from peppan_amd import synth, _native as N
    if name in ('protein1k', 'protein1k_sample', 'protein10k_sample'):
        n = 10000 if name.startswith('protein10k') else 1000
        names, seqs = synth.make_genes(n, 1002, seed=355)
        q, t, _ = protein_sets([s.decode() for s in seqs])
        p = O.default_params(45., 25., 10, 5)                     # PEPPAN.py:229-230: --min_id 0.5 - 0.05, --min_ratio 0.25
        ms = [O.min_score(len(s)) for s in q]
        q_idx = None if name == 'protein1k' else list(range(0, 256)) if name == 'protein1k_sample' else list(range(0, n, 40))
        return q, t, p, ms, q_idx, 'synthgenes-v1 %d genes x 1002 nt, translated search (diamond replacement)' % n
    if name in ('nucl1k', 'nucl1k_sample'):
        names, seqs = synth.make_genes(1000, 1002, seed=355)
        comp = bytes.maketrans(b'ACGT', b'TGCA')
        q = [O.nt_codes(s) for s in seqs]
        t = []
        for s in seqs:                                           # both strands of every subject (RunBlast.runBlast)
            t += [O.nt_codes(s), O.nt_codes(s.translate(comp)[::-1])]
        pn = N.nucleotide_params(45., 25.)
        p = O.params_from(pn)
        ms = [O.min_score(len(s), pn.dbsize, pn.max_evalue, pn.ka_lambda, pn.ka_k) for s in q]
        q_idx = None if name == 'nucl1k' else list(range(0, 128))
        return q, t, p, ms, q_idx, 'synthgenes-v1 1000 genes x 1002 nt, nucleotide search (blastn replacement)'
    if name in ('real', 'real_sample'):
        import gzip
        seqs, cur = {}, None
        with gzip.open(os.path.join(ROOT, 'tests', 'golden', 'g16_real_genes.fa.gz'), 'rt') as f:
            for line in f:
                if line.startswith('>'):
                    cur = int(line[1:])
                else:
                    seqs[cur] = line.strip().upper()
        nts = [seqs[k] for k in sorted(seqs)]
        q, t, _ = protein_sets(nts)
        p = O.default_params(45., 25., 10, 5)
        ms = [O.min_score(len(s)) for s in q]
        q_idx = None if name == 'real' else list(range(0, len(q), 12))
        return q, t, p, ms, q_idx, 'golden G16: %d real E. coli genes of the reference examples/, translated search' % len(nts)
    raise SystemExit('unknown workload ' + name)


if __name__ == '__main__':
    for w in sys.argv[1:] or ['protein1k_sample']:
        q, t, p, ms, q_idx, desc = workload(w)
        from peppan_amd import _native as N
        r = report(q, t, p, ms, q_idx, sensitive_shapes=N.DEFAULT_SHAPES + N.SENSITIVE_SHAPES)
        r['workload'] = desc
        print(json.dumps(r))
