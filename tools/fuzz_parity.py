"""Randomised parity of the whole search against the CPU oracle: random protein sets (families, exact duplicates, X and B at the ends and inside,
repeats), random thresholds, score tables other than BLOSUM62 (ties on the diagonal, residues that are neither dominant nor harmless, cheap
gaps), every combination of the test switches (exact sizing of the alignment stage, identical pairs compared / swept, 32-bit sweeps, LSD sort).
python3 tools/fuzz_parity.py [cases] [seed] [nucl | nt]   - exits 1 on the first difference; nucl: the nucleotide configuration of the engine; nt: the translated
search from nucleotide sets (K1 inside, the self-search path) against the oracle and against the plain stream"""
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


def from_nucleotide_cases(ctx, cases, rng):
    """the translated search FROM NUCLEOTIDES (K1 inside; the self-search path of round 6: self_prepare decides which targets repeat a query, seed_match drops their
    diagonal-0 hits): gene sets with stops in frame 1, frame shifts, ambiguous bases, duplicates, empty and tiny genes, genes of several chunks; reference = the queries,
    or the queries with genes replaced / edited / reordered / missing / added; K1 in front of or inside the search; the plain stream (reserved[0] = 10) beside it"""
    sense = [a + b + c for a in 'ACGT' for b in 'ACGT' for c in 'ACGT' if a + b + c not in ('TAA', 'TAG', 'TGA')]
    for case in range(cases):
        names, seqs = synth.make_genes(int(rng.integers(20, 260)), 0, seed=int(rng.integers(1, 1 << 30)))
        seqs = [bytes(x) for x in seqs]
        for _ in range(int(rng.integers(0, 12))):
            i = int(rng.integers(0, len(seqs)))
            g = seqs[i]
            kind = int(rng.integers(0, 9))
            if kind == 0 and len(g) > 40: at = 3 * int(rng.integers(1, len(g) // 3 - 2)); g = g[:at] + b'TAA' + g[at + 3:]         # a stop inside frame 1
            elif kind == 1: g = b'ACG'[:int(rng.integers(1, 3))] + g                                                               # another reading frame
            elif kind == 2: g = b''
            elif kind == 3: g = b'ATGAAATAA'
            elif kind == 4: g = g.lower()
            elif kind == 5 and len(g) > 60: at = int(rng.integers(0, len(g) - 9)); g = g[:at] + b'NNNRYKN'[:int(rng.integers(1, 8))] + g[at + 7:]
            elif kind == 6: seqs.append(g)                                                                                          # a duplicate
            elif kind == 7: g = ('ATG' + ''.join(sense[k] for k in rng.integers(0, 61, size=int(rng.integers(1001, 1500)))) + 'TAA').encode()     # several chunks per frame
            elif kind == 8 and len(g) > 90: at = 3 * int(rng.integers(1, len(g) // 3 - 12)); g = g[:at] + b'NNN' * int(rng.integers(1, 12)) + g[at + 3:]
            seqs[i] = g
        ref = list(seqs)
        how = int(rng.integers(0, 7))
        if how == 1 and len(ref) > 3: ref[int(rng.integers(0, len(ref)))] = ref[int(rng.integers(0, len(ref)))]
        elif how == 2:
            i = int(rng.integers(0, len(ref)))
            if len(ref[i]) > 50:
                at = int(rng.integers(0, len(ref[i]) - 1)); ref[i] = ref[i][:at] + (b'A' if ref[i][at:at + 1].upper() != b'A' else b'C') + ref[i][at + 1:]
        elif how == 3: ref = ref[::-1]
        elif how == 4: ref = ref[:-int(rng.integers(1, 4))]
        elif how == 5: ref = ref + [ref[0]]
        qry = seqs if how != 6 else seqs[int(rng.integers(0, 10)):][:int(rng.integers(5, 120))]
        gtable = int(rng.choice([11, 11, 4]))
        frames = 6
        inside = int(rng.integers(0, 2))
        res = []
        for flag in (0, 10):
            p = N.default_params(float(rng.choice([45., 45., 30., 70.])) if flag == 0 else res_p.min_id_pct, 25., 10, 5)
            if flag == 0:
                p.top_k = int(rng.choice([10, 10, 2, 50])); res_p = p
            else:
                p.top_k = res_p.top_k
            p.reserved[0] = flag
            ctx.set_query_nt(qry, gtable); ctx.set_ref_nt(ref, frames, gtable)
            if inside:
                ctx.invalidate_translation()
            res.append(ctx.search(p))
        (gh, gc, st), (ph, pc, pst) = res
        q_aa = [O.aa_codes(O.query_frame(x.decode(), gtable)[1]) for x in qry]
        t_aa = []
        for x in ref:
            for aa in O.translate_frames(x.decode(), range(1, 7), gtable):
                t_aa += [O.aa_codes(c) for o, c in O.ref_chunks(aa)]
        op = O.default_params(res_p.min_id_pct, 25., res_p.top_k, 5)
        oh, oc, ost = O.search(q_aa, t_aa, op)
        bad = [f for f in FIELDS if len(gh) != len(oh) or not np.array_equal(gh[f], oh[f])]
        bad += ['cigar arena'] if not np.array_equal(gc, oc) else []
        bad += ['stat ' + k for k in ('candidates', 'pairs', 'cells', 'tracebacks') if st[k] != ost[k]]
        bad += ['plain stream: ' + f for f in FIELDS if len(gh) != len(ph) or not np.array_equal(gh[f], ph[f])]
        bad += ['plain stream: stat ' + k for k in ('query_seeds', 'target_seeds', 'seed_hits', 'candidates') if st[k] != pst[k]]
        print('nt case %3d: %4d queries x %4d reference genes (variant %d, table %d, K1 %s), min_id %.0f top_k %d: %8d raw hits %6d candidates, %6d hits  %s'
              % (case, len(qry), len(ref), how, gtable, 'inside' if inside else 'in front', res_p.min_id_pct, res_p.top_k, st['seed_hits'], st['candidates'], len(gh),
                 'ok' if not bad else 'DIFFERENT: ' + ', '.join(bad)), flush=True)
        if bad:
            sys.exit(1)
    print('all %d from-nucleotide cases identical to the oracle and to the plain stream' % cases)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    ctx = N.Context(0)
    if len(sys.argv) > 3 and sys.argv[3] == 'nucl':
        return nucleotide_cases(ctx, cases, rng)
    if len(sys.argv) > 3 and sys.argv[3] == 'nt':
        return from_nucleotide_cases(ctx, cases, rng)
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
