/*
 * full_sw.c - INDEPENDENT GROUND TRUTH for the self-defined aligner (test infrastructure, NOT product code).
 *
 * Textbook Gotoh local alignment over the FULL dynamic-programming matrix: no seeds, no diagonal band, no ungapped
 * filter, no candidate selection.  It shares no code with align_oracle.c (different recurrences layout: column sweep with
 * explicit H / E / F matrices, traceback by re-deriving each move from the stored matrices instead of direction bits).
 * What it pins: the search that replaces `diamond blastp --id --query-cover --evalue 1` (uberBlast.py:550) and
 * `blastn -word_size 17 ... -evalue 1e-2` (uberBlast.py:294) promises EVERY target that meets the thresholds; the
 * heuristic (seeds -> ungapped filter -> 128-diagonal band) of align_oracle.c / the HIP kernels can lose some.
 * tests/test_recall_full_sw.py counts, on BASELINE configs[1] and on the real genes of fixture G16,
 *   (i)  whether every reported score is the pair's full-matrix optimum (band losses), and
 *   (ii) recall = reported pairs / pairs whose full-matrix alignment passes the same score / identity / cover cuts.
 *
 * Scoring conventions = the ones the command lines imply: substitution table sub[q*32 + t] (BLOSUM62 for proteins,
 * +2 / -3 for nucleotides), a gap of length k costs gap_open + k * gap_ext (11 + k, 6 + 2k).
 *
 *   fullsw_score_matrix   best local score of every (query, target) pair.  Inter-sequence vectorisation: LANES
 *                         targets advance together through one query (Rognes' SWIPE layout), lane loops are plain C
 *                         that gcc vectorises; 16-bit lanes when the score cannot overflow, 32-bit otherwise.
 *   fullsw_align          one pair: score, end cell (first maximum in row-major order), traceback -> start cell,
 *                         identities, alignment columns, CIGAR (len << 2 | op, 0 = M, 1 = I query only, 2 = D target only).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LANES 32
#define PADCODE 31

typedef struct {
    int32_t score;
    int32_t q_start, q_end, t_start, t_end;      /* 1-based inclusive; 0 when score == 0 */
    uint32_t n_ident, aln_len, n_runs;
} fullsw_aln;

/* ---- score-only, LANES targets at a time.  TYPE is the lane type; NEGV a value no path can reach from below. */
#define DEFINE_BATCH(NAME, TYPE, NEGV)                                                                                   \
    __attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))                                                        \
    static void NAME(const int8_t *sub, int oe, int ext, const uint8_t *q, int m, const uint8_t *const *tp,             \
                     const int32_t *tl, int nmax, TYPE *Hc, TYPE *Ec, int32_t *best_out)                                 \
    {                                                                                                                    \
        TYPE prof[32][LANES] __attribute__((aligned(64)));                                                               \
        TYPE best[LANES] __attribute__((aligned(64)));                                                                   \
        for (int l = 0; l < LANES; ++l) best[l] = 0;                                                                     \
        for (int i = 0; i < m; ++i)                                                                                      \
            for (int l = 0; l < LANES; ++l) { Hc[(size_t)i * LANES + l] = 0; Ec[(size_t)i * LANES + l] = NEGV; }         \
        for (int j = 0; j < nmax; ++j) {                                                                                 \
            uint8_t tc[LANES];                                                                                           \
            for (int l = 0; l < LANES; ++l) tc[l] = (j < tl[l]) ? (uint8_t)(tp[l][j] & 31) : PADCODE;                    \
            for (int r = 0; r < 32; ++r)                                                                                 \
                for (int l = 0; l < LANES; ++l) prof[r][l] = sub[r * 32 + tc[l]];                                        \
            TYPE hdiag[LANES] __attribute__((aligned(64))), hup[LANES] __attribute__((aligned(64))),                     \
                F[LANES] __attribute__((aligned(64)));                                                                   \
            for (int l = 0; l < LANES; ++l) { hdiag[l] = 0; hup[l] = 0; F[l] = NEGV; }                                   \
            for (int i = 0; i < m; ++i) {                                                                                \
                const TYPE *pr = prof[q[i] & 31];                                                                        \
                TYPE *H = Hc + (size_t)i * LANES, *E = Ec + (size_t)i * LANES;                                           \
                for (int l = 0; l < LANES; ++l) {                                                                        \
                    TYPE hold = H[l];                                                                                    \
                    TYPE e1 = (TYPE)(E[l] - ext), e2 = (TYPE)(hold - oe);                                                \
                    TYPE e = e1 > e2 ? e1 : e2;                                                                          \
                    TYPE f1 = (TYPE)(F[l] - ext), f2 = (TYPE)(hup[l] - oe);                                              \
                    TYPE f = f1 > f2 ? f1 : f2;                                                                          \
                    TYPE h = (TYPE)(hdiag[l] + pr[l]);                                                                   \
                    h = h > e ? h : e;                                                                                   \
                    h = h > f ? h : f;                                                                                   \
                    h = h > 0 ? h : 0;                                                                                   \
                    best[l] = best[l] > h ? best[l] : h;                                                                 \
                    E[l] = e; F[l] = f; H[l] = h;                                                                        \
                    hdiag[l] = hold; hup[l] = h;                                                                         \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
        for (int l = 0; l < LANES; ++l) best_out[l] = best[l];                                                           \
    }

DEFINE_BATCH(batch16, int16_t, -30000)
DEFINE_BATCH(batch32, int32_t, -(1 << 28))

typedef struct { uint32_t idx; int32_t len; } tl_t;
static int cmp_tl(const void *a, const void *b)
{
    const tl_t *x = a, *y = b;
    if (x->len != y->len) return x->len < y->len ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

void fullsw_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
    (void)n;
#endif
}

/* out[q * nt + t] = best local score of query q against target t */
int fullsw_score_matrix(const int8_t *sub, int gap_open, int gap_ext, const uint8_t *q_res, const uint64_t *q_off, uint32_t nq,
                        const uint8_t *t_res, const uint64_t *t_off, uint32_t nt, int32_t *out)
{
    if (!nq || !nt) return 0;
    const int oe = gap_open + gap_ext;
    int maxs = 1;
    for (int i = 0; i < 1024; ++i) if (sub[i] > maxs) maxs = sub[i];
    tl_t *ord = malloc(sizeof(tl_t) * nt);
    for (uint32_t t = 0; t < nt; ++t) { ord[t].idx = t; ord[t].len = (int32_t)(t_off[t + 1] - t_off[t]); }
    qsort(ord, nt, sizeof(tl_t), cmp_tl);              /* lanes of a batch have similar lengths */
    const uint32_t nb = (nt + LANES - 1) / LANES;
    int32_t qmax = 0;
    for (uint32_t q = 0; q < nq; ++q) { int32_t L = (int32_t)(q_off[q + 1] - q_off[q]); if (L > qmax) qmax = L; }
    #pragma omp parallel
    {
        void *Hc = NULL, *Ec = NULL;
        if (posix_memalign(&Hc, 64, (size_t)(qmax + 1) * LANES * 4) || posix_memalign(&Ec, 64, (size_t)(qmax + 1) * LANES * 4)) abort();
        #pragma omp for schedule(dynamic, 4) collapse(2)
        for (uint32_t q = 0; q < nq; ++q)
            for (uint32_t b = 0; b < nb; ++b) {
                const uint8_t *qs = q_res + q_off[q];
                const int m = (int)(q_off[q + 1] - q_off[q]);
                const uint8_t *tp[LANES];
                int32_t tl[LANES], best[LANES], nmax = 0;
                for (int l = 0; l < LANES; ++l) {
                    uint32_t k = b * LANES + l;
                    if (k < nt) { tp[l] = t_res + t_off[ord[k].idx]; tl[l] = ord[k].len; }
                    else { tp[l] = t_res; tl[l] = 0; }
                    if (tl[l] > nmax) nmax = tl[l];
                }
                const long bound = (long)(m < nmax ? m : nmax) * maxs;
                if (bound < 30000) batch16(sub, oe, gap_ext, qs, m, tp, tl, nmax, Hc, Ec, best);
                else batch32(sub, oe, gap_ext, qs, m, tp, tl, nmax, Hc, Ec, best);
                for (int l = 0; l < LANES; ++l) {
                    uint32_t k = b * LANES + l;
                    if (k < nt) out[(size_t)q * nt + ord[k].idx] = best[l];
                }
            }
        free(Hc); free(Ec);
    }
    free(ord);
    return 0;
}

/* ---- one pair with traceback.  Matrices are (Lq+1) x (Lt+1), row 0 / column 0 = the empty prefix.
 * E[i][j]: best score of an alignment ending in a gap that consumes target residue j (horizontal move),
 * F[i][j]: ... that consumes query residue i (vertical move).  Returns 0, -1 on allocation failure, -2 when cigar_cap is too small. */
#define NEG32 (-(1 << 28))
int fullsw_align(const int8_t *sub, int gap_open, int gap_ext, const uint8_t *q, int32_t Lq, const uint8_t *t, int32_t Lt,
                 fullsw_aln *o, uint32_t *cigar, uint32_t cigar_cap)
{
    memset(o, 0, sizeof(*o));
    if (Lq <= 0 || Lt <= 0) return 0;
    const int oe = gap_open + gap_ext;
    const size_t W = (size_t)Lt + 1, N = ((size_t)Lq + 1) * W;
    int32_t *H = malloc(N * sizeof(int32_t)), *E = malloc(N * sizeof(int32_t)), *F = malloc(N * sizeof(int32_t));
    if (!H || !E || !F) { free(H); free(E); free(F); return -1; }
    for (size_t j = 0; j < W; ++j) { H[j] = 0; E[j] = NEG32; F[j] = NEG32; }
    int32_t best = 0, bi = 0, bj = 0;
    for (int32_t i = 1; i <= Lq; ++i) {
        int32_t *h = H + (size_t)i * W, *e = E + (size_t)i * W, *f = F + (size_t)i * W;
        const int32_t *hp = h - W, *fp = f - W;
        const int8_t *row = sub + (q[i - 1] & 31) * 32;
        h[0] = 0; e[0] = NEG32; f[0] = NEG32;
        for (int32_t j = 1; j <= Lt; ++j) {
            int32_t ev = e[j - 1] - gap_ext, eo = h[j - 1] - oe;
            e[j] = ev > eo ? ev : eo;
            int32_t fv = fp[j] - gap_ext, fo = hp[j] - oe;
            f[j] = fv > fo ? fv : fo;
            int32_t v = hp[j - 1] + row[t[j - 1] & 31];
            if (e[j] > v) v = e[j];
            if (f[j] > v) v = f[j];
            if (v < 0) v = 0;
            h[j] = v;
            if (v > best) { best = v; bi = i; bj = j; }     /* first maximum in row-major order */
        }
    }
    o->score = best;
    int rc = 0;
    if (best > 0) {
        /* traceback: at an H cell prefer the diagonal, then E, then F; inside a gap keep extending only when extending is
         * strictly better than opening (the same preferences a left-to-right reader of the recurrences would choose) */
        uint32_t *rev = malloc(((size_t)Lq + Lt + 2) * sizeof(uint32_t));
        uint32_t nr = 0;
        int32_t i = bi, j = bj, state = 0;
        uint32_t nid = 0, cols = 0;
        int32_t is = bi, js = bj;
        #define PUSH(op) do { if (nr && (rev[nr - 1] & 3u) == (uint32_t)(op)) rev[nr - 1] += 4; else rev[nr++] = 4u | (uint32_t)(op); ++cols; } while (0)
        while (i > 0 && j > 0) {
            size_t c = (size_t)i * W + j;
            if (state == 0) {
                if (H[c] == 0) break;
                int32_t d = H[c - W - 1] + sub[(q[i - 1] & 31) * 32 + (t[j - 1] & 31)];
                if (H[c] == d) { PUSH(0); if (q[i - 1] == t[j - 1]) ++nid; is = i; js = j; --i; --j; }
                else if (H[c] == E[c]) state = 1;
                else state = 2;
            } else if (state == 1) {
                PUSH(2);
                state = (E[c - 1] - gap_ext > H[c - 1] - oe) ? 1 : 0;
                --j;
            } else {
                PUSH(1);
                state = (F[c - W] - gap_ext > H[c - W] - oe) ? 2 : 0;
                --i;
            }
        }
        #undef PUSH
        o->q_start = is; o->q_end = bi; o->t_start = js; o->t_end = bj;
        o->n_ident = nid; o->aln_len = cols; o->n_runs = nr;
        if (cigar) {
            if (nr > cigar_cap) rc = -2;
            else for (uint32_t k = 0; k < nr; ++k) cigar[k] = rev[nr - 1 - k];
        }
        free(rev);
    }
    free(H); free(E); free(F);
    return rc;
}
