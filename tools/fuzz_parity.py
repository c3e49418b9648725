"""Randomised parity of the whole search against the CPU oracle: random protein sets (families, exact duplicates, X and B at the ends and inside,
repeats), random thresholds, score tables other than BLOSUM62 (ties on the diagonal, residues that are neither dominant nor harmless, cheap
gaps), every combination of the test switches (exact sizing of the alignment stage, identical pairs compared / swept, 32-bit sweeps, LSD sort).
python3 tools/fuzz_parity.py [cases] [seed] [nucl]   - exits 1 on the first difference; nucl: the nucleotide configuration of the engine"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from peppan_amd import _native as N, synth
from oracle import oracle as O

FIELDS = ('q', 't', 'q_start', 'q_end', 't_start', 't_end', 'score', 'nm', 'n_ident', 'aln_len', 'cigar_runs', 'bin', 'cigar_off', 'cells')


def make_set(rng):
    n = int(rng.integers(20, 400))
    base = synth.make_proteins(n, length=(int(rng.integers(20, 80)), int(rng.integers(90, 900))), seed=int(rng.integers(1, 1 << 30)),
                               family=int(rng.integers(1, 6)), sub=float(rng.choice([0., 0.03, 0.15, 0.3])))
    out = list(base)
    for _ in range(int(rng.integers(0, 30))):
        p = base[int(rng.integers(0, n))].copy()
        kind = int(rng.integers(0, 8))
        if kind == 1: p = np.concatenate([p, np.full(int(rng.integers(1, 4)), 23, np.uint8)])       # stop(s) at the end
        elif kind == 2: p[int(rng.integers(0, len(p)))] = 23                                         # X somewhere
        elif kind == 3: p = np.concatenate([np.full(1, 23, np.uint8), p])                            # X in front
        elif kind == 4: p[int(rng.integers(0, len(p)))] = 1                                          # B
        elif kind == 5: p = np.concatenate([p[:17]] * int(rng.integers(2, 12)))                      # a repeat
        elif kind == 6: p = p[:max(12, len(p) - int(rng.integers(1, 40)))]
        elif kind == 7: p[int(rng.integers(0, len(p)))] = int(rng.integers(0, 20))
        out.append(p.copy()); out.append(p.copy())
    order = rng.permutation(len(out))
    return [out[i] for i in order]


def nucleotide_cases(ctx, cases, rng):
    """the blastn-like configuration (base-4 exact 17-mers, +2 / -3, gaps 6 + 2k, both strands): genes, duplicates, runs of N, fragments"""
    rc = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    for case in range(cases):
        names, seqs = synth.make_genes(int(rng.integers(30, 500)), 0, seed=int(rng.integers(1, 1 << 30)))
        codes = [O.nt_codes(x) for x in seqs]
        for _ in range(int(rng.integers(0, 20))):
            c = codes[int(rng.integers(0, len(codes)))].copy()
            kind = int(rng.integers(0, 4))
            if kind == 1: c[int(rng.integers(0, len(c) - 5)):][:int(rng.integers(1, 5))] = 4
            elif kind == 2: c = c[int(rng.integers(0, 30)):]
            elif kind == 3: c[-1] = 4
            codes.append(c); codes.append(c.copy())
        targets = codes + [rc[c[::-1]] for c in codes]
        p = N.nucleotide_params(float(rng.choice([70., 90.])), float(rng.choice([25., 50.])), top_k=int(rng.choice([1000, 1000, 3, 1])), hsp_mode=int(rng.choice([1, 2])))
        flags = int(rng.choice([0, 1, 2, 3, 4, 5]))
        p.reserved2 = flags
        p.reserved[1] = int(rng.integers(0, 4) == 0)
        subjects = None
        if case % 2:                               # the nucleotide tool's own layout (forward strands, then reverse complements): the two strands of a sequence are ONE subject in hsp_mode 2
            texts = [''.join('ACGTN'[x] for x in c) for c in codes]
            ctx.set_query_nt(texts); ctx.set_ref_nt(texts, 6, 11); ctx.use_nt_as_residues(2)
            subjects = list(range(len(codes))) * 2 if p.hsp_mode == 2 else None
        else:
            ctx.set_query_aa(codes); ctx.set_ref_aa(targets)
        gh, gc, st = ctx.search(p)
        ms = np.array([O.min_score(len(c), p.dbsize, p.max_evalue, p.ka_lambda, p.ka_k) for c in codes], dtype=np.int32)
        oh, oc, ost = O.search(codes, targets, O.params_from(p), min_scores=ms, subjects=subjects)
        bad = [f for f in FIELDS if len(gh) != len(oh) or not np.array_equal(gh[f], oh[f])]
        bad += ['cigar arena'] if not np.array_equal(gc, oc) else []
        bad += ['stat ' + k for k in ('candidates', 'pairs', 'cells', 'tracebacks') if st[k] != ost[k]]
        print('nucleotide case %3d: %4d sequences x 2 strands, hsp_mode %d top_k %4d, switches %d/%d: %6d candidates (%5d settled), %6d hits  %s'
              % (case, len(codes), p.hsp_mode, p.top_k, flags, p.reserved[1], st['candidates'], st['candidates_settled'], len(gh), 'ok' if not bad else 'DIFFERENT: ' + ', '.join(bad)), flush=True)
        if bad:
            sys.exit(1)
    print('all %d nucleotide cases identical to the oracle' % cases)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    ctx = N.Context(0)
    if len(sys.argv) > 3 and sys.argv[3] == 'nucl':
        return nucleotide_cases(ctx, cases, rng)
    for case in range(cases):
        prots = make_set(rng)
        p = N.default_params(float(rng.choice([0., 30., 45., 70.])), float(rng.choice([0., 10., 25., 60.])), int(rng.choice([1, 2, 10, 50])), int(rng.choice([1, 5])))
        table = int(rng.integers(0, 5))
        sub = np.frombuffer(bytes(p.sub), dtype=np.int8).reshape(32, 32).copy()
        if table == 1:                         # +5 / -4 on all letters (X dominant too)
            sub[:26, :26] = -4
            sub[np.arange(26), np.arange(26)] = 5
        elif table == 2:                       # a tie on the diagonal's row: A as good against S as against itself
            sub[0, 18] = sub[18, 0] = sub[0, 0]
        elif table == 3:                       # X scores +1 against everything
            sub[23, :26] = 1; sub[:26, 23] = 1
        elif table == 4:                       # cheap gaps
            p.gap_open, p.gap_ext = int(rng.integers(0, 3)), int(rng.integers(1, 3))
        for i in range(32):
            for j in range(32):
                p.sub[i * 32 + j] = int(sub[i, j])
        flags = int(rng.choice([0, 1, 2, 3, 4, 5]))
        p.reserved2 = flags
        p.reserved[1] = int(rng.integers(0, 4) == 0)
        p.reserved[2] = int(rng.integers(0, 4) == 0)
        ctx.set_query_aa(prots); ctx.set_ref_aa(prots)
        gh, gc, st = ctx.search(p)
        op = O.params_from(p)
        for i in range(1024):
            op.sub[i] = p.sub[i]
        oh, oc, ost = O.search(prots, prots, op)
        bad = [f for f in FIELDS if len(gh) != len(oh) or not np.array_equal(gh[f], oh[f])]
        bad += ['cigar arena'] if not np.array_equal(gc, oc) else []
        bad += ['stat ' + k for k in ('candidates', 'pairs', 'cells', 'tracebacks') if st[k] != ost[k]]
        print('case %3d: %4d proteins, table %d, gaps %d+%d, switches %d/%d/%d, min_id %.0f top_k %d: %6d candidates (%5d settled), %6d hits  %s'
              % (case, len(prots), table, p.gap_open, p.gap_ext, flags, p.reserved[1], p.reserved[2], p.min_id_pct, p.top_k, st['candidates'], st['candidates_settled'], len(gh),
                 'ok' if not bad else 'DIFFERENT: ' + ', '.join(bad)), flush=True)
        if bad:
            sys.exit(1)
    print('all %d cases identical to the oracle' % cases)


if __name__ == '__main__':
    main()
