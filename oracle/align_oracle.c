/*
 * align_oracle.c - CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the seed-and-extend protein search that replaces the
 * `diamond blastp` calls of the reference (uberBlast.py:531-552).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED for this half: the reference delegates the arithmetic to the
 * DIAMOND binary, which is absent from /root/reference (.MISSING_LARGE_BLOBS:2),
 * unpinned (setup.py:22 lists no version) and cannot be run here.  This file
 * therefore DEFINES the algorithm (published DIAMOND design: reduced-alphabet
 * spaced seeds -> banded affine Smith-Waterman, BLOSUM62 11/1, e-value /
 * identity / query-cover filters, top-k targets per query) and the HIP kernels
 * are held bit-exact to it.  The reference's own call site fixes the parameters:
 *   --id 100*min_id --query-cover 100*min_ratio --evalue 1 -k 10 --dbsize 5000000
 *   5 round-robin database splits                     (uberBlast.py:546-552)
 * and the consumer fixes the fields each hit must carry: POS, CIGAR, |SEQ|, NM,
 * ZR (raw score), ZS (query start)                    (uberBlast.py:25-58).
 *
 * Algorithm (all integer; every tie broken explicitly so results are unique):
 *  1. seeds: residues -> 11-letter reduced alphabet; for each spaced shape a key
 *     = sum g[p+off_k]*base^k; every (query pos, target pos) pair with equal key
 *     is a seed hit on diagonal d = tpos - qpos.
 *  2. candidates: unique (q, t, bin) with bin = floor((d + 2^23)/64); the band
 *     of a candidate covers diagonals [c-32, c+95], c = 64*bin - 2^23.
 *  3. banded affine local alignment per candidate (gap of length k costs
 *     open+k*ext); best cell = max H, ties -> smallest i, then smallest j.
 *  4. per (q,t): keep the best candidate (score desc, bin asc); require
 *     score >= min_score[q] (Karlin-Altschul e-value cut, see oracle_min_score).
 *  5. traceback (H==0 stop, then diagonal, then E, then F; gap extension only
 *     when strictly better than opening) -> CIGAR (M / I = query only / D =
 *     target only), start cell, identities.
 *  6. filters: 100*ident/alnlen >= min_id_pct, 100*qspan/qlen >= min_qcov_pct.
 *  7. top-k per (q, t mod n_splits): rank by (score desc, t asc), keep rank<k.
 *  8. output ordered by (q, t).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

#define NEG (-(1 << 28))
#define DIAG_OFF (1 << 23)
#define BIN_W 64
#define BAND 128
#define BAND_LEAD 32
#define STAGE1_LEN 16

typedef struct {
    int32_t gap_open;      /* 11 */
    int32_t gap_ext;       /* 1  */
    int32_t n_shapes;
    int32_t base;          /* size of the reduced alphabet */
    int32_t weight[4];
    int32_t offs[4][32];
    uint8_t reduce[32];    /* residue code -> reduced letter, 0xFF = never seeds */
    int8_t  sub[32 * 32];  /* substitution scores, [q*32 + t] */
    double  min_id_pct;    /* diamond --id */
    double  min_qcov_pct;  /* diamond --query-cover */
    int32_t top_k;         /* diamond -k */
    int32_t n_splits;      /* 5 database splits */
    int32_t ungapped_min;  /* a seed hit only nominates a candidate if its ungapped x-drop score reaches this (0 = off) */
    int32_t xdrop;         /* x-drop of the ungapped extension */
    int32_t ext_right;     /* max residues scored to the right, starting at the seed's first position */
    int32_t ext_left;      /* max residues scored to the left of the seed's first position */
    int32_t hsp_mode;      /* 0: one alignment per (q, t) = its best band (diamond --max-hsps 1);
                              1: every band of (q, t) that reaches min_score, minus duplicates (same end cell: keep the
                                 higher score, then the lower bin) - the nucleotide tool, where a subject can carry several copies;
                              2: BLAST's way with the HSPs of a subject (blastn behind uberBlast.py:294, `-num_alignments 1000`, no binary here to pin it
                                 against: NCBI's published behaviour restated) - the bands of (q, t) that reach min_score are taken in the order score
                                 descending, bin ascending, and one is dropped when an ACCEPTED one before it shares its start cell or its end cell
                                 (Blast_HSPListPurgeHSPsWithCommonEndpoints) or holds its query range and its subject range inside its own (the interval-tree
                                 containment test of the gapped stage); and top_k counts SUBJECTS (oracle_set_subjects: targets that are strands / frames of
                                 one sequence), ranked by their best alignment - every alignment of a kept subject stays */
    int32_t t_base;        /* index of target 0 in the whole reference set when the targets are one shard of it: split = (t + t_base) mod n_splits */
    int32_t stage1_min;    /* first stage of the pre-filter: the right extension must have reached this after its first STAGE1_LEN residues (0 = off) */
    int32_t pad0;
} oracle_params;

typedef struct {
    uint32_t q, t;
    uint32_t q_start, q_end;   /* 1-based, inclusive, residues */
    uint32_t t_start, t_end;
    int32_t  score;
    uint32_t nm;               /* aln_len - n_ident */
    uint32_t n_ident, aln_len;
    uint32_t cigar_runs;
    int32_t  bin;
    uint64_t cigar_off;
    uint64_t cells;            /* in-band in-matrix cells of the winning candidate */
} oracle_hit;

/* standard BLOSUM62, order ARNDCQEGHILKMFPSTWYVBZX* (public domain, NCBI) */
static const char B62_ORDER[] = "ARNDCQEGHILKMFPSTWYVBZX*";
static const int8_t B62[24][24] = {
    { 4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0,-2,-1, 0,-4},
    {-1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3,-1, 0,-1,-4},
    {-2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3, 3, 0,-1,-4},
    {-2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3, 4, 1,-1,-4},
    { 0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1,-3,-3,-2,-4},
    {-1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2, 0, 3,-1,-4},
    {-1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
    { 0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3,-1,-2,-1,-4},
    {-2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3, 0, 0,-1,-4},
    {-1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3,-3,-3,-1,-4},
    {-1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1,-4,-3,-1,-4},
    {-1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2, 0, 1,-1,-4},
    {-1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1,-3,-1,-1,-4},
    {-2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1,-3,-3,-1,-4},
    {-1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2,-2,-1,-2,-4},
    { 1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2, 0, 0, 0,-4},
    { 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0,-1,-1, 0,-4},
    {-3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3,-4,-3,-2,-4},
    {-2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1,-3,-2,-1,-4},
    { 0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4,-3,-2,-1,-4},
    {-2,-1, 3, 4,-3, 0, 1,-1, 0,-3,-4, 0,-3,-3,-2, 0,-1,-4,-3,-3, 4, 1,-1,-4},
    {-1, 0, 0, 1,-3, 3, 4,-2, 0,-3,-3, 1,-1,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
    { 0,-1,-1,-1,-2,-1,-1,-1,-1,-1,-1,-1,-1,-1,-2, 0, 0,-2,-1,-1,-1,-1,-1,-4},
    {-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4, 1},
};

/* Fill protein defaults: residue code = letter - 'A'; J,O,U score as X; codes >= 26 are
 * padding and score -64 against everything.  Reduced alphabet (Buchfink et al. 2015):
 * [KREDQN] [C] [G] [H] [ILV] [M] [F] [Y] [W] [P] [STA]; B,J,O,U,X,Z never seed.
 * Shapes: 111101110111 and 111011010010111 (weight 10). */
void oracle_default_params(oracle_params *p)
{
    memset(p, 0, sizeof(*p));
    p->gap_open = 11; p->gap_ext = 1;
    int idx[26];
    for (int c = 0; c < 26; ++c) {
        const char *f = strchr(B62_ORDER, 'A' + c);
        idx[c] = f ? (int)(f - B62_ORDER) : 22;   /* X */
    }
    for (int a = 0; a < 32; ++a)
        for (int b = 0; b < 32; ++b)
            p->sub[a * 32 + b] = (a < 26 && b < 26) ? B62[idx[a]][idx[b]] : -64;
    memset(p->reduce, 0xFF, 32);
    const char *groups[11] = {"KREDQN", "C", "G", "H", "ILV", "M", "F", "Y", "W", "P", "STA"};
    for (int g = 0; g < 11; ++g)
        for (const char *c = groups[g]; *c; ++c) p->reduce[*c - 'A'] = (uint8_t)g;
    p->base = 11;
    const char *shapes[2] = {"111101110111", "111011010010111"};
    p->n_shapes = 2;
    for (int s = 0; s < 2; ++s) {
        int w = 0;
        for (int k = 0; shapes[s][k]; ++k)
            if (shapes[s][k] == '1') p->offs[s][w++] = k;
        p->weight[s] = w;
    }
    p->min_id_pct = 0.; p->min_qcov_pct = 0.; p->top_k = 10; p->n_splits = 5;
    p->ungapped_min = 55; p->xdrop = 12; p->ext_right = 40; p->ext_left = 24; p->stage1_min = 24; p->pad0 = 0;
}

/* smallest raw score whose e-value m*n*K*exp(-lambda*S) is <= max_evalue
 * (gapped BLOSUM62 11/1 statistics: lambda 0.267, K 0.041; n = --dbsize) */
int32_t oracle_min_score_ka(uint32_t qlen, double dbsize, double max_evalue, double lambda, double K)
{
    double s = log(K * (double)qlen * dbsize / max_evalue) / lambda;
    int32_t r = (int32_t)ceil(s);
    return r < 1 ? 1 : r;
}

int32_t oracle_min_score(uint32_t qlen, double dbsize, double max_evalue)
{
    const double lambda = 0.267, K = 0.041;
    double s = log(K * (double)qlen * dbsize / max_evalue) / lambda;
    int32_t r = (int32_t)ceil(s);
    return r < 1 ? 1 : r;
}

/* ------------------------------------------------------------------ seeds */
typedef struct { uint64_t key; uint32_t seq, pos; } seed_t;

static int cmp_seed(const void *a, const void *b)
{
    const seed_t *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y);
}

static int seed_key(const oracle_params *p, int sh, const uint8_t *s, uint32_t len, uint32_t pos, uint64_t *key)
{
    int w = p->weight[sh];
    if ((uint64_t)pos + p->offs[sh][w - 1] >= len) return 0;
    uint64_t k = 0, mul = 1;
    for (int i = 0; i < w; ++i) {
        uint8_t c = s[pos + p->offs[sh][i]];
        uint8_t g = c < 32 ? p->reduce[c] : 0xFF;
        if (g == 0xFF) return 0;
        k += mul * g; mul *= (uint64_t)p->base;
    }
    *key = k;
    return 1;
}

/* ungapped x-drop score of the diagonal through a seed hit: right part starts AT (qpos, tpos), left part just before it;
 * each part stops at the sequence end, after `ext` residues, or once the running sum falls more than xdrop below its best */
static int32_t ungapped_score(const oracle_params *p, const uint8_t *q, int32_t Lq, int32_t qpos, const uint8_t *t, int32_t Lt, int32_t tpos)
{
    int32_t s = 0, br = 0, bl = 0, k = 0;
    for (; k < p->ext_right && qpos + k < Lq && tpos + k < Lt; ++k) {
        /* stage 1: a hit whose right extension has not reached stage1_min within its first STAGE1_LEN residues - the seed itself and a
         * few residues behind it - is a chance hit of the reduced alphabet and is dropped before the rest is looked at */
        if (k == STAGE1_LEN && br < p->stage1_min) return -1;
        s += p->sub[(q[qpos + k] & 31) * 32 + (t[tpos + k] & 31)];
        if (s > br) br = s;
        else if (br - s > p->xdrop) break;
    }
    if (br < p->stage1_min) return -1;          /* (the extension ended inside the first STAGE1_LEN residues) */
    s = 0;
    for (int32_t k = 1; k <= p->ext_left && qpos - k >= 0 && tpos - k >= 0; ++k) {
        s += p->sub[(q[qpos - k] & 31) * 32 + (t[tpos - k] & 31)];
        if (s > bl) bl = s;
        else if (bl - s > p->xdrop) break;
    }
    return br + bl;
}

static inline uint64_t cand_key(uint32_t q, uint32_t t, int32_t bin)
{
    return ((uint64_t)q << 43) | ((uint64_t)t << 18) | (uint64_t)(uint32_t)bin;
}

/* candidates: sorted unique keys (q:21 | t:25 | bin:18) */
/* phase clocks of oracle_search (bench.py's cpu_baseline reports them): wall seconds of the seed stage and of the alignment stage, and the
 * thread-seconds the alignment stage spent in score-only sweeps and in band_align (traceback sweeps, rule 5a, walks) */
static double g_phase[4];
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
void oracle_phase_seconds(double *out, int reset) { for (int k = 0; k < 4; ++k) { out[k] = g_phase[k]; if (reset) g_phase[k] = 0.; } }

static uint64_t *find_candidates(const oracle_params *p,
                                 const uint8_t *qr, const uint64_t *qo, uint32_t nq,
                                 const uint8_t *tr, const uint64_t *to, uint32_t nt, uint64_t *n_out)
{
    uint64_t cap = 1 << 16, n = 0;
    uint64_t *c = malloc(cap * sizeof(uint64_t));
    uint64_t aq = qo[nq];
    seed_t *qs = malloc((aq + 1) * sizeof(seed_t));
    for (int sh = 0; sh < p->n_shapes; ++sh) {
        uint64_t m = 0;
        for (uint32_t q = 0; q < nq; ++q) {
            uint32_t len = (uint32_t)(qo[q + 1] - qo[q]);
            for (uint32_t pos = 0; pos < len; ++pos) {
                uint64_t k;
                if (seed_key(p, sh, qr + qo[q], len, pos, &k)) { qs[m].key = k; qs[m].seq = q; qs[m].pos = pos; ++m; }
            }
        }
        qsort(qs, m, sizeof(seed_t), cmp_seed);
        /* targets are independent; the candidate SET does not depend on the order they are visited in (sorted + deduplicated below) */
        #pragma omp parallel
        {
            uint64_t lcap = 1 << 12, ln = 0;
            uint64_t *lc = malloc(lcap * sizeof(uint64_t));
            #pragma omp for schedule(dynamic, 64) nowait
            for (uint32_t t = 0; t < nt; ++t) {
                uint32_t len = (uint32_t)(to[t + 1] - to[t]);
                for (uint32_t pos = 0; pos < len; ++pos) {
                    uint64_t k;
                    if (!seed_key(p, sh, tr + to[t], len, pos, &k)) continue;
                    uint64_t lo = 0, hi = m;
                    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (qs[mid].key < k) lo = mid + 1; else hi = mid; }
                    for (; lo < m && qs[lo].key == k; ++lo) {
                        if (p->ungapped_min > 0) {
                            uint32_t qq = qs[lo].seq;
                            if (ungapped_score(p, qr + qo[qq], (int32_t)(qo[qq + 1] - qo[qq]), (int32_t)qs[lo].pos, tr + to[t], (int32_t)len, (int32_t)pos) < p->ungapped_min) continue;
                        }
                        int32_t d = (int32_t)pos - (int32_t)qs[lo].pos;
                        int32_t bin = (d + DIAG_OFF) / BIN_W;
                        if (ln == lcap) { lcap *= 2; lc = realloc(lc, lcap * sizeof(uint64_t)); }
                        lc[ln++] = cand_key(qs[lo].seq, t, bin);
                    }
                }
            }
            #pragma omp critical
            {
                if (n + ln > cap) { while (n + ln > cap) cap *= 2; c = realloc(c, cap * sizeof(uint64_t)); }
                memcpy(c + n, lc, ln * sizeof(uint64_t));
                n += ln;
            }
            free(lc);
        }
    }
    free(qs);
    qsort(c, n, sizeof(uint64_t), cmp_u64);
    uint64_t u = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (i == 0 || c[i] != c[i - 1]) c[u++] = c[i];
    *n_out = u;
    return c;
}

/* ------------------------------------------------------------------ banded SW */
typedef struct {
    int32_t score, iend, jend;
    int32_t end_lane;   /* lowest diagonal pair (band column / 2) that holds a cell with the maximal score */
    int32_t band;       /* diagonals of this band (row length of dir) */
    uint64_t cells;
    uint8_t *dir;       /* [Lq][band] nibble per byte: bits0-1 src, bit2 eExt, bit3 fExt */
} sw_out;

/* affine local alignment restricted to the `band` diagonals dlo .. dlo + band - 1 (band <= BAND) */
static void banded_sw(const oracle_params *p, const uint8_t *q, int32_t Lq, const uint8_t *t, int32_t Lt,
                      int32_t dlo, int32_t band, int keep_dir, sw_out *o)
{
    const int32_t oe = p->gap_open + p->gap_ext, ext = p->gap_ext;
    /* rolling rows indexed by band column c = d - dlo */
    int32_t *base = malloc(sizeof(int32_t) * (BAND + 2) * 4);
    int32_t *Hp = base;
    int32_t *Fp = Hp + (BAND + 2), *Hc = Fp + (BAND + 2), *Fc = Hc + (BAND + 2);
    for (int c = 0; c < BAND + 2; ++c) { Hp[c] = 0; Fp[c] = NEG; Hc[c] = 0; Fc[c] = NEG; }
    o->score = 0; o->iend = -1; o->jend = -1; o->cells = 0; o->end_lane = -1; o->band = band;
    o->dir = keep_dir ? calloc((size_t)Lq * band, 1) : NULL;
    for (int32_t i = 0; i < Lq; ++i) {
        int32_t jlo = i + dlo;
        int32_t hl = 0, el = NEG;            /* left neighbour (i, j-1) inside the band */
        for (int c = 0; c < band; ++c) {
            int32_t j = jlo + c;
            if (j < 0 || j >= Lt) { Hc[c] = 0; Fc[c] = NEG; hl = 0; el = NEG; continue; }
            /* previous row, same diagonal -> band column c; up (i-1, j) -> diagonal d+1 -> column c+1 */
            int32_t hd = (i > 0 && j > 0) ? Hp[c] : 0;
            int32_t hu = 0, fu = NEG;
            if (i > 0 && c + 1 < band) { hu = Hp[c + 1]; fu = Fp[c + 1]; }
            if (c == 0 || j == 0) { hl = 0; el = NEG; }
            int32_t e_ext = el - ext, e_open = hl - oe;
            int32_t f_ext = fu - ext, f_open = hu - oe;
            int32_t E = e_ext > e_open ? e_ext : e_open;
            int32_t F = f_ext > f_open ? f_ext : f_open;
            int32_t h = hd + p->sub[(q[i] & 31) * 32 + (t[j] & 31)];
            int32_t H = h;
            if (E > H) H = E;
            if (F > H) H = F;
            if (H < 0) H = 0;
            if (keep_dir) {
                uint8_t src = (H == 0) ? 0 : (H == h) ? 1 : (H == E) ? 2 : 3;
                o->dir[(size_t)i * band + c] = (uint8_t)(src | ((e_ext > e_open) ? 4 : 0) | ((f_ext > f_open) ? 8 : 0));
            }
            if (H > o->score) { o->score = H; o->iend = i; o->jend = j; o->end_lane = c >> 1; }   /* row-major scan => min i, then min j */
            else if (H == o->score && H > 0 && (c >> 1) < o->end_lane) o->end_lane = c >> 1;
            Hc[c] = H; Fc[c] = F;
            hl = H; el = E;
            o->cells++;
        }
        int32_t *tmp = Hp; Hp = Hc; Hc = tmp; tmp = Fp; Fp = Fc; Fc = tmp;
    }
    free(base);
}

typedef struct { uint32_t *runs; uint32_t n, cap; } runbuf;
static void push_op(runbuf *b, uint32_t op)
{
    if (b->n && (b->runs[b->n - 1] & 3) == op) { b->runs[b->n - 1] += 4; return; }
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 16; b->runs = realloc(b->runs, b->cap * sizeof(uint32_t)); }
    b->runs[b->n++] = (1u << 2) | op;
}

/* ops: 0 = M, 1 = I (query only), 2 = D (target only); runs come out reversed (end -> start) */
static void traceback(const sw_out *o, const uint8_t *q, const uint8_t *t, int32_t dlo,
                      runbuf *rb, int32_t *istart, int32_t *jstart, uint32_t *n_ident, uint32_t *aln_len)
{
    int32_t i = o->iend, j = o->jend, state = 0;
    *n_ident = 0; *aln_len = 0; rb->n = 0;
    *istart = i; *jstart = j;
    for (;;) {
        uint8_t nib = o->dir[(size_t)i * o->band + (j - i - dlo)];
        if (state == 0) {
            uint8_t src = nib & 3;
            if (src == 0) break;
            if (src == 1) {
                push_op(rb, 0); ++*aln_len;
                if (q[i] == t[j]) ++*n_ident;
                *istart = i; *jstart = j;
                if (i == 0 || j == 0) break;
                --i; --j;
            } else state = (src == 2) ? 1 : 2;
        } else if (state == 1) {         /* E: gap consuming the target */
            push_op(rb, 2); ++*aln_len;
            state = (nib & 4) ? 1 : 0;
            --j;
        } else {                          /* F: gap consuming the query */
            push_op(rb, 1); ++*aln_len;
            state = (nib & 8) ? 2 : 0;
            --i;
        }
    }
}

/* The alignment a band reports.  Its score T is the maximum of the 128-diagonal band (the score pass).  The traceback is taken in
 * the 64-diagonal SUB-band centred on the lowest diagonal pair that holds a cell with score T (pairs = the two diagonals one GPU
 * lane owns; sub-band = pairs [L0, L0 + 32), L0 = that pair - 16 clamped to [0, 32]) whenever the sub-band alone reaches T: its end
 * cell is then the first cell with score T in row-major order INSIDE the sub-band.  An alignment that needs more room (the sub-band's
 * maximum stays below T) is traced in the full band, from the full band's first maximal cell.  Either way the reported alignment has
 * score T; the rule only fixes WHICH optimal alignment is reported and lets the traceback pass sweep half the cells. */
#define SUB_BAND 64
static uint64_t g_traced = 0, g_full_band = 0, g_gapless = 0;      /* how often the sub-band was enough / no DP was needed (oracle_trace_counts) */
void oracle_trace_counts(uint64_t *out, int reset) { out[0] = g_traced; out[1] = g_full_band; out[2] = g_gapless; if (reset) g_traced = g_full_band = g_gapless = 0; }

/* Gapless shortcut (rule 5a).  T = the band's score, d = one diagonal of the band.  If an UNGAPPED segment of d
 * scores T, the alignment that ends in the segment's last cell is that segment and nothing else: H >= the running ungapped sum in every
 * cell of d, so an H above the sum anywhere inside the segment would carry on to an H above T at its end - impossible, T is the band's
 * maximum - hence H == h (the diagonal move, which has priority in the traceback) in every cell of the segment and H == 0 in front of
 * it.  The scan is Kadane's: the running sum restarts AFTER a cell that brings it to <= 0 (so every prefix of the reported segment is
 * positive) and the first cell at which it equals T ends the segment.  Returns 1 and the segment [is, ie] x [js, je], else 0. */
/* test switch: 0 = every reported alignment comes from the traceback (the definition before rule 5a was added); the tests hold the two equal */
static int g_rule5a = 1;
void oracle_set_rule5a(int on) { g_rule5a = on; }
static int gapless_segment(const oracle_params *p, const uint8_t *q, int32_t Lq, const uint8_t *t, int32_t Lt, int32_t d, int32_t T,
                           int32_t *is, int32_t *ie)
{
    int32_t i0 = d < 0 ? -d : 0, i1 = Lq - 1 < Lt - 1 - d ? Lq - 1 : Lt - 1 - d;
    int32_t run = 0, start = i0;
    for (int32_t i = i0; i <= i1; ++i) {
        run += p->sub[(q[i] & 31) * 32 + (t[i + d] & 31)];
        if (run <= 0) { run = 0; start = i + 1; continue; }
        if (run == T) { *is = start; *ie = i; return 1; }
    }
    return 0;
}
static void band_align(const oracle_params *p, const uint8_t *q, int32_t Lq, const uint8_t *t, int32_t Lt, int32_t dlo,
                       sw_out *o, runbuf *rb, int32_t *is, int32_t *js, uint32_t *nid, uint32_t *al)
{
    sw_out wide;
    banded_sw(p, q, Lq, t, Lt, dlo, BAND, 0, &wide);
    *o = wide;
    if (wide.score <= 0) return;
    #pragma omp atomic
    ++g_traced;
    /* rule 5a: the two diagonals of the lowest diagonal pair that holds the score (lower diagonal first) are tried for an ungapped
     * segment with that score; the first one found IS the reported alignment - one M run, no second pass over the band */
    for (int32_t x = 0; x < 2 && g_rule5a; ++x) {
        const int32_t d = dlo + 2 * wide.end_lane + x;
        int32_t gs, ge;
        if (gapless_segment(p, q, Lq, t, Lt, d, wide.score, &gs, &ge)) {
            rb->n = 0;
            for (int32_t i = gs; i <= ge; ++i) push_op(rb, 0);
            *is = gs; *js = gs + d; *al = (uint32_t)(ge - gs + 1); *nid = 0;
            for (int32_t i = gs; i <= ge; ++i) *nid += q[i] == t[i + d];
            o->iend = ge; o->jend = ge + d;
            #pragma omp atomic
            ++g_gapless;
            return;
        }
    }
    int32_t L0 = wide.end_lane - SUB_BAND / 4;
    if (L0 < 0) L0 = 0;
    if (L0 > (BAND - SUB_BAND) / 2) L0 = (BAND - SUB_BAND) / 2;
    const int32_t ndlo = dlo + 2 * L0;
    sw_out narrow;
    banded_sw(p, q, Lq, t, Lt, ndlo, SUB_BAND, 1, &narrow);
    if (narrow.score == wide.score) {
        traceback(&narrow, q, t, ndlo, rb, is, js, nid, al);
        o->iend = narrow.iend; o->jend = narrow.jend;
        free(narrow.dir);
        return;
    }
    free(narrow.dir);
    #pragma omp atomic
    ++g_full_band;
    banded_sw(p, q, Lq, t, Lt, dlo, BAND, 1, &wide);
    traceback(&wide, q, t, dlo, rb, is, js, nid, al);
    free(wide.dir);
    o->dir = NULL;
}

static int cmp_hit_rank(const void *a, const void *b)
{
    const oracle_hit *x = *(const oracle_hit *const *)a, *y = *(const oracle_hit *const *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->bin < y->bin ? -1 : (x->bin > y->bin);
}

typedef struct { int32_t bin, score, iend, jend, is, js; uint32_t nid, al, nruns; uint64_t cells; uint32_t *runs; int dead; } band_aln;

/* hsp_mode 2: the subject (sequence) every target belongs to; NULL = every target is a subject of its own */
static const uint32_t *g_subject = NULL;
void oracle_set_subjects(const uint32_t *subject) { g_subject = subject; }

#ifdef _OPENMP
#include <omp.h>
#endif
/* threads used by oracle_search (0 = all cores); the results do not depend on it */
void oracle_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
    (void)n;
#endif
}
int oracle_get_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

typedef struct { oracle_hit *hits; uint32_t nh, caph; uint32_t *cig; uint64_t ncig, capcig, cells, traced; } group_out;

static oracle_hit *group_new_hit(group_out *r, uint32_t n_runs)
{
    if (r->nh == r->caph) { r->caph = r->caph ? r->caph * 2 : 4; r->hits = realloc(r->hits, r->caph * sizeof(oracle_hit)); }
    if (r->ncig + n_runs > r->capcig) { r->capcig = (r->ncig + n_runs) * 2 + 64; r->cig = realloc(r->cig, r->capcig * sizeof(uint32_t)); }
    return &r->hits[r->nh++];
}

/* all bands [g0, g1) of one (query, target) pair; cigar_off of the emitted hits is relative to r->cig */
static void align_group(const oracle_params *p, const uint64_t *cand, uint64_t g0, uint64_t g1,
                        const uint8_t *q_res, const uint64_t *q_off, const uint8_t *t_res, const uint64_t *t_off,
                        const int32_t *min_score, group_out *r)
{
    runbuf rb = {0};
    double t_score = 0., t_trace = 0., t0;
    {
        uint32_t q = (uint32_t)(cand[g0] >> 43), t = (uint32_t)((cand[g0] >> 18) & ((1u << 25) - 1));
        const uint8_t *qs = q_res + q_off[q], *ts = t_res + t_off[t];
        int32_t Lq = (int32_t)(q_off[q + 1] - q_off[q]), Lt = (int32_t)(t_off[t + 1] - t_off[t]);
        int32_t best = -1, best_bin = 0;
        for (uint64_t g = g0; g < g1; ++g) {
            int32_t bin = (int32_t)(cand[g] & ((1u << 18) - 1));
            int32_t dlo = bin * BIN_W - DIAG_OFF - BAND_LEAD;
            sw_out o;
            t0 = now_s();
            banded_sw(p, qs, Lq, ts, Lt, dlo, BAND, 0, &o);
            t_score += now_s() - t0;
            r->cells += o.cells;
            if (o.score > best) { best = o.score; best_bin = bin; }
        }
        if (p->hsp_mode == 0) {
            if (best > 0 && best >= min_score[q]) {
                int32_t dlo = best_bin * BIN_W - DIAG_OFF - BAND_LEAD;
                sw_out o;
                int32_t is, js; uint32_t nid, al;
                t0 = now_s();
                band_align(p, qs, Lq, ts, Lt, dlo, &o, &rb, &is, &js, &nid, &al);
                t_trace += now_s() - t0;
                ++r->traced;
                double idp = (double)nid * 100.0 / (double)al;
                double qcov = (double)(o.iend - is + 1) * 100.0 / (double)Lq;
                if (idp >= p->min_id_pct && qcov >= p->min_qcov_pct) {
                    oracle_hit *h = group_new_hit(r, rb.n);
                    h->q = q; h->t = t; h->q_start = is + 1; h->q_end = o.iend + 1; h->t_start = js + 1; h->t_end = o.jend + 1;
                    h->score = o.score; h->n_ident = nid; h->aln_len = al; h->nm = al - nid; h->bin = best_bin;
                    h->cigar_runs = rb.n; h->cigar_off = r->ncig; h->cells = o.cells;
                    for (uint32_t x = 0; x < rb.n; ++x) r->cig[r->ncig++] = rb.runs[rb.n - 1 - x];
                }
            }
        } else {
            uint64_t nb = g1 - g0, na = 0;
            band_aln *al_ = calloc(nb + 1, sizeof(band_aln));
            for (uint64_t g = g0; g < g1; ++g) {
                int32_t bin = (int32_t)(cand[g] & ((1u << 18) - 1));
                int32_t dlo = bin * BIN_W - DIAG_OFF - BAND_LEAD;
                sw_out o;
                banded_sw(p, qs, Lq, ts, Lt, dlo, BAND, 0, &o);
                if (o.score > 0 && o.score >= min_score[q]) {
                    band_aln *b = &al_[na++];
                    band_align(p, qs, Lq, ts, Lt, dlo, &o, &rb, &b->is, &b->js, &b->nid, &b->al);
                    ++r->traced;
                    b->bin = bin; b->score = o.score; b->iend = o.iend; b->jend = o.jend; b->cells = o.cells; b->nruns = rb.n;
                    b->runs = malloc((rb.n + 1) * sizeof(uint32_t));
                    for (uint32_t x = 0; x < rb.n; ++x) b->runs[x] = rb.runs[rb.n - 1 - x];
                }
            }
            if (p->hsp_mode == 2) {
                /* in rank order (score descending, bin ascending = position ascending): against the accepted ones in front */
                uint64_t *ord = malloc((na + 1) * sizeof(uint64_t));
                for (uint64_t x = 0; x < na; ++x) ord[x] = x;
                for (uint64_t x = 1; x < na; ++x) {                       /* insertion sort: stable, al_ is in bin order */
                    uint64_t v = ord[x], y = x;
                    while (y > 0 && al_[ord[y - 1]].score < al_[v].score) { ord[y] = ord[y - 1]; --y; }
                    ord[y] = v;
                }
                for (uint64_t a = 0; a < na; ++a) {
                    band_aln *x = &al_[ord[a]];
                    for (uint64_t b = 0; b < a && !x->dead; ++b) {
                        const band_aln *y = &al_[ord[b]];
                        if (y->dead) continue;
                        if ((x->iend == y->iend && x->jend == y->jend) || (x->is == y->is && x->js == y->js) ||
                            (y->is <= x->is && x->iend <= y->iend && y->js <= x->js && x->jend <= y->jend)) x->dead = 1;
                    }
                }
                free(ord);
            } else
            for (uint64_t x = 0; x < na; ++x)
                for (uint64_t y = 0; y < na; ++y)
                    if (x != y && al_[x].iend == al_[y].iend && al_[x].jend == al_[y].jend &&
                        (al_[y].score > al_[x].score || (al_[y].score == al_[x].score && al_[y].bin < al_[x].bin))) al_[x].dead = 1;
            for (uint64_t x = 0; x < na; ++x) {
                band_aln *b = &al_[x];
                double idp = (double)b->nid * 100.0 / (double)b->al;
                double qcov = (double)(b->iend - b->is + 1) * 100.0 / (double)Lq;
                if (!b->dead && idp >= p->min_id_pct && qcov >= p->min_qcov_pct) {
                    oracle_hit *h = group_new_hit(r, b->nruns);
                    h->q = q; h->t = t; h->q_start = b->is + 1; h->q_end = b->iend + 1; h->t_start = b->js + 1; h->t_end = b->jend + 1;
                    h->score = b->score; h->n_ident = b->nid; h->aln_len = b->al; h->nm = b->al - b->nid; h->bin = b->bin;
                    h->cigar_runs = b->nruns; h->cigar_off = r->ncig; h->cells = b->cells;
                    memcpy(r->cig + r->ncig, b->runs, b->nruns * sizeof(uint32_t));
                    r->ncig += b->nruns;
                }
                free(b->runs);
            }
            free(al_);
        }
    }
    free(rb.runs);
    #pragma omp atomic
    g_phase[2] += t_score;
    #pragma omp atomic
    g_phase[3] += t_trace;
}

/* full search.  q_off/t_off have n+1 entries (plain concatenation, no padding).
 * min_score[nq] per query.  Returns 0; caller frees *hits and *cigar with oracle_free. */
int oracle_search(const oracle_params *p,
                  const uint8_t *q_res, const uint64_t *q_off, uint32_t nq,
                  const uint8_t *t_res, const uint64_t *t_off, uint32_t nt,
                  const int32_t *min_score,
                  oracle_hit **hits_out, uint64_t *n_hits, uint32_t **cigar_out, uint64_t *n_cigar,
                  uint64_t *stats /* [0]=candidates [1]=cells over all candidates [2]=(q,t) pairs [3]=tracebacks */)
{
    uint64_t nc = 0;
    const double t_begin = now_s();
    uint64_t *cand = find_candidates(p, q_res, q_off, nq, t_res, t_off, nt, &nc);
    const double t_seeded = now_s();
    g_phase[0] += t_seeded - t_begin;
    /* (q, t) groups of candidates are independent: aligned in parallel, results appended in group order */
    uint64_t n_grp = 0;
    uint64_t *grp = malloc((nc + 2) * sizeof(uint64_t));
    for (uint64_t g = 0; g < nc; ++g)
        if (g == 0 || (cand[g] >> 18) != (cand[g - 1] >> 18)) grp[n_grp++] = g;
    grp[n_grp] = nc;
    group_out *res = calloc(n_grp + 1, sizeof(group_out));
    #pragma omp parallel for schedule(dynamic, 16)
    for (uint64_t k = 0; k < n_grp; ++k)
        align_group(p, cand, grp[k], grp[k + 1], q_res, q_off, t_res, t_off, min_score, &res[k]);
    g_phase[1] += now_s() - t_seeded;
    uint64_t nh = 0, ncig = 0, cells_all = 0, traced = 0;
    const uint64_t pairs = n_grp;
    for (uint64_t k = 0; k < n_grp; ++k) { nh += res[k].nh; ncig += res[k].ncig; }
    oracle_hit *hits = malloc((nh + 1) * sizeof(oracle_hit));
    uint32_t *cig = malloc((ncig + 1) * sizeof(uint32_t));
    nh = 0; ncig = 0;
    for (uint64_t k = 0; k < n_grp; ++k) {
        for (uint32_t x = 0; x < res[k].nh; ++x) {
            oracle_hit h = res[k].hits[x];
            h.cigar_off += ncig;
            hits[nh++] = h;
        }
        if (res[k].ncig) memcpy(cig + ncig, res[k].cig, res[k].ncig * sizeof(uint32_t));
        ncig += res[k].ncig;
        cells_all += res[k].cells; traced += res[k].traced;
        free(res[k].hits); free(res[k].cig);
    }
    free(res); free(grp);
    /* top-k per (q, split): hits are ordered by (q, t) already */
    uint64_t out = 0, a = 0;
    uint8_t *keep = calloc(nh + 1, 1);
    const oracle_hit **tmp = malloc((nh + 1) * sizeof(*tmp));
    while (a < nh) {
        uint64_t b = a;
        while (b < nh && hits[b].q == hits[a].q) ++b;
        for (int s = 0; s < p->n_splits; ++s) {
            uint64_t m = 0;
            for (uint64_t k = a; k < b; ++k) if ((int)((hits[k].t + (uint32_t)(p->t_base % p->n_splits)) % (uint32_t)p->n_splits) == s) tmp[m++] = &hits[k];
            qsort(tmp, m, sizeof(*tmp), cmp_hit_rank);
            if (p->hsp_mode == 2) {
                /* the first alignment of a subject in rank order is its best: subjects are counted as they show up, every alignment of the first top_k stays */
                uint64_t n_subj = 0;
                uint32_t *seen = malloc((m + 1) * sizeof(uint32_t));
                for (uint64_t k = 0; k < m; ++k) {
                    const uint32_t sj = g_subject ? g_subject[tmp[k]->t] : tmp[k]->t;
                    uint64_t at = 0;
                    while (at < n_subj && seen[at] != sj) ++at;
                    if (at == n_subj) seen[n_subj++] = sj;
                    if (at < (uint64_t)p->top_k) keep[tmp[k] - hits] = 1;
                }
                free(seen);
            } else
            for (uint64_t k = 0; k < m && k < (uint64_t)p->top_k; ++k) keep[tmp[k] - hits] = 1;
        }
        a = b;
    }
    /* compact hits and cigar arena */
    uint32_t *cig2 = malloc((ncig + 1) * sizeof(uint32_t)); uint64_t nc2 = 0;
    for (uint64_t k = 0; k < nh; ++k) if (keep[k]) {
        oracle_hit h = hits[k];
        memcpy(cig2 + nc2, cig + h.cigar_off, h.cigar_runs * sizeof(uint32_t));
        h.cigar_off = nc2; nc2 += h.cigar_runs;
        hits[out++] = h;
    }
    free(cig); free(keep); free(tmp); free(cand);
    *hits_out = hits; *n_hits = out; *cigar_out = cig2; *n_cigar = nc2;
    if (stats) { stats[0] = nc; stats[1] = cells_all; stats[2] = pairs; stats[3] = traced; }
    return 0;
}

/* one banded alignment with traceback, for kernel-level tests */
int oracle_align_one(const oracle_params *p, const uint8_t *q, int32_t Lq, const uint8_t *t, int32_t Lt, int32_t bin,
                     oracle_hit *h, uint32_t *cigar, uint32_t cigar_cap)
{
    int32_t dlo = bin * BIN_W - DIAG_OFF - BAND_LEAD;
    sw_out o; runbuf rb = {0};
    int32_t is = 0, js = 0; uint32_t nid = 0, al = 0;
    band_align(p, q, Lq, t, Lt, dlo, &o, &rb, &is, &js, &nid, &al);
    memset(h, 0, sizeof(*h));
    h->score = o.score; h->cells = o.cells; h->bin = bin;
    if (o.score > 0) {
        h->q_start = is + 1; h->q_end = o.iend + 1; h->t_start = js + 1; h->t_end = o.jend + 1;
        h->n_ident = nid; h->aln_len = al; h->nm = al - nid; h->cigar_runs = rb.n;
        for (uint32_t r = 0; r < rb.n && r < cigar_cap; ++r) cigar[r] = rb.runs[rb.n - 1 - r];
    }
    free(rb.runs);
    return 0;
}

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------ nucleotide rescoring, mode 1
 * restates cigar2score mode 1 (uberBlast.py:226-249) on integer counts: walks the nt CIGAR
 * (len<<2|op, op 0=M 1=I 2=D) over encoded bases (A0 C1 G3 T4 other 2, uberBlast.py:270-271);
 * reverse hits read the reference slice backwards as 4-code (uberBlast.py:412).
 * out[5] = nMatch, nMismatch, nGap, bGap, mGap */
void oracle_rescore_counts(const uint8_t *q, const uint8_t *r, int64_t qs, int64_t rs, int64_t re,
                           const uint32_t *cigar, uint32_t n_runs, int64_t *out)
{
    int64_t qi = qs - 1, nmatch = 0, nmis = 0, ngap = 0, bgap = 0, mgap = 0;
    int rev = rs > re;
    int64_t ri = rev ? rs - 1 : rs - 1;   /* 0-based index of the first aligned reference base */
    for (uint32_t k = 0; k < n_runs; ++k) {
        int64_t n = cigar[k] >> 2; int op = cigar[k] & 3;
        if (op == 0) {
            for (int64_t x = 0; x < n; ++x) {
                int a = q[qi + x];
                int b = rev ? 4 - r[ri - x] : r[ri + x];
                if (a == b) ++nmatch; else ++nmis;
            }
            qi += n; ri += rev ? -n : n;
        } else {
            ++ngap; bgap += n; if (n > 3) mgap += n;
            if (op == 1) qi += n; else ri += rev ? -n : n;
        }
    }
    out[0] = nmatch; out[1] = nmis; out[2] = ngap; out[3] = bgap; out[4] = mgap;
}

/* ------------------------------------------------------------------ connected components (single linkage)
 * label[x] = smallest node id in x's component; restates the PARTITION produced by the
 * reference's union-find (PEPPAN.py:1598-1607), not its root choice / member order. */
static uint32_t uf_find(uint32_t *p, uint32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
void oracle_components(uint32_t n, uint64_t m, const uint32_t *a, const uint32_t *b, uint32_t *label)
{
    for (uint32_t i = 0; i < n; ++i) label[i] = i;
    for (uint64_t e = 0; e < m; ++e) {
        uint32_t x = uf_find(label, a[e]), y = uf_find(label, b[e]);
        if (x == y) continue;
        if (x < y) label[y] = x; else label[x] = y;
    }
    for (uint32_t i = 0; i < n; ++i) label[i] = uf_find(label, i);
}

/* ------------------------------------------------------------------ linear-time clustering (replaces `mmseqs linclust`)
 * PARITY UNPINNED (MMseqs2 is absent, clust.py:62-66 only consumes its "representative, member" relation).
 * Definition (published linclust idea, Steinegger & Soeding 2018, reduced to an exactly reproducible form):
 *  1. every sequence selects the m k-mers with the smallest (hash, position) among its valid k-mers
 *     (codes < base; hash = splitmix64 finaliser of the base-`base` k-mer value);
 *  2. the centre of a k-mer is the longest sequence that selected it (ties: lowest index);
 *  3. each (sequence s, selected k-mer) whose centre c != s defines one diagonal (position in c - position in s);
 *     the UNGAPPED overlap of s and c on that diagonal is accepted when matches >= min_id * overlap and
 *     overlap >= min_cov * len(c) and overlap >= min_cov * len(s)   (double arithmetic, no division);
 *  3b. a centre that no diagonal of s accepted is aligned WITH gaps: banded affine local alignment (the engine of the search,
 *     band = the 128 diagonals around 64 * floor((d + 2^23) / 64) - 2^23 - 32 for every failed diagonal d, best band by score
 *     then lowest bin; nucleotides +2 / -3 gap 6 + 2k, amino acids BLOSUM62 gap 11 + k); accepted when identical columns >=
 *     min_id * alignment columns and the aligned span covers >= min_cov of s and of c;
 *  4. greedy assignment in priority order (length desc, index asc): an unassigned sequence becomes a
 *     representative and takes every still unassigned sequence it was accepted as the centre of.
 * rep_out[s] = index of s's representative. */
static uint64_t lc_mix(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

typedef struct { uint64_t h, key; uint32_t pos; } lc_sel;

static int lc_cmp_sel(const void *a, const void *b)
{
    const lc_sel *x = a, *y = b;
    if (x->h != y->h) return x->h < y->h ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

typedef struct { uint64_t key; uint32_t len, idx; } lc_ctr;

static int lc_cmp_ctr(const void *a, const void *b)
{
    const lc_ctr *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->len != y->len) return x->len > y->len ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

typedef struct { uint32_t len, idx; } lc_pri;
static int lc_cmp_pri(const void *a, const void *b)
{
    const lc_pri *x = a, *y = b;
    if (x->len != y->len) return x->len > y->len ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* scoring of the gapped verification (mirrors k9_params of linclust.hip): base 4 = nucleotides, +2 / -3, gap 6 + 2k;
 * otherwise amino acids in the order ACDEFGHIKLMNPQRSTVWY (+ one code for anything else = X), BLOSUM62, gap 11 + k */
static void lc_gapped_params(int base, oracle_params *p)
{
    oracle_params d;
    oracle_default_params(&d);
    *p = d;
    for (int a = 0; a < 32; ++a) for (int b = 0; b < 32; ++b) p->sub[a * 32 + b] = -64;
    if (base == 4) {
        for (int a = 0; a < 5; ++a) for (int b = 0; b < 5; ++b) p->sub[a * 32 + b] = (int8_t)((a == b && a < 4) ? 2 : -3);
        p->gap_open = 6; p->gap_ext = 2;
    } else {
        static const char *letters = "ACDEFGHIKLMNPQRSTVWYX";
        for (int a = 0; a <= 20 && a <= base; ++a)
            for (int b = 0; b <= 20 && b <= base; ++b) p->sub[a * 32 + b] = d.sub[(letters[a] - 'A') * 32 + (letters[b] - 'A')];
        p->gap_open = 11; p->gap_ext = 1;
    }
}

int oracle_linclust(const uint8_t *res, const uint64_t *off, uint32_t n, int base, int k, int m, double min_id, double min_cov,
                    uint32_t *rep_out, uint64_t *stats /* [0] selected k-mers [1] verified pairs [2] accepted edges */)
{
    lc_sel *sel = calloc((size_t)n * m + 1, sizeof(lc_sel));
    uint32_t *cnt = calloc(n + 1, sizeof(uint32_t));
    uint32_t maxlen = 0;
    for (uint32_t s = 0; s < n; ++s) { uint32_t L = (uint32_t)(off[s + 1] - off[s]); if (L > maxlen) maxlen = L; }
    uint64_t n_sel = 0;
    /* the m smallest (hash, position) of every sequence: kept in a small array while the positions are walked (a replacement is rare once the
     * array holds small hashes), sorted at the end - the same m entries in the same order as sorting all of them.  Sequences are independent: one
     * OpenMP thread each (a million sequences in well under a minute instead of half an hour; the result does not depend on the schedule). */
    (void)maxlen;
    #pragma omp parallel for schedule(dynamic, 256) reduction(+ : n_sel)
    for (uint32_t s = 0; s < n; ++s) {
        const uint8_t *q = res + off[s];
        uint32_t L = (uint32_t)(off[s + 1] - off[s]), c = 0, worst = 0;
        lc_sel best[64];
        const uint32_t mm = (uint32_t)(m < 64 ? m : 64);
        for (uint32_t p = 0; p + k <= L; ++p) {
            uint64_t key = 0, mul = 1; int ok = 1;
            for (int i = 0; i < k; ++i) { if (q[p + i] >= base) { ok = 0; break; } key += mul * q[p + i]; mul *= (uint64_t)base; }
            if (!ok) continue;
            lc_sel e; e.h = lc_mix(key); e.key = key; e.pos = p;
            if (c < mm) {
                best[c] = e;
                if (c == 0 || lc_cmp_sel(&best[c], &best[worst]) > 0) worst = c;
                ++c;
            } else if (lc_cmp_sel(&e, &best[worst]) < 0) {
                best[worst] = e;
                worst = 0;
                for (uint32_t z = 1; z < mm; ++z) if (lc_cmp_sel(&best[z], &best[worst]) > 0) worst = z;
            }
        }
        qsort(best, c, sizeof(lc_sel), lc_cmp_sel);
        cnt[s] = c;
        memcpy(sel + (size_t)s * m, best, c * sizeof(lc_sel));
        n_sel += c;
    }
    /* centres */
    lc_ctr *ctr = malloc((n_sel + 1) * sizeof(lc_ctr));
    uint64_t nc = 0;
    for (uint32_t s = 0; s < n; ++s)
        for (uint32_t r = 0; r < cnt[s]; ++r) { ctr[nc].key = sel[(size_t)s * m + r].key; ctr[nc].len = (uint32_t)(off[s + 1] - off[s]); ctr[nc].idx = s; ++nc; }
    qsort(ctr, nc, sizeof(lc_ctr), lc_cmp_ctr);
    oracle_params gp;
    lc_gapped_params(base, &gp);
    /* accepted centres per member */
    uint32_t *acc = malloc(((size_t)n * m + 1) * sizeof(uint32_t));
    uint32_t *nacc = calloc(n + 1, sizeof(uint32_t));
    uint64_t n_ver = 0, n_acc = 0;
    /* (every member works on its own slice of acc / nacc: one OpenMP thread per member) */
    #pragma omp parallel for schedule(dynamic, 64) reduction(+ : n_ver, n_acc)
    for (uint32_t s = 0; s < n; ++s) {
        const uint8_t *qs = res + off[s];
        int64_t Ls = (int64_t)(off[s + 1] - off[s]);
        uint32_t seen_c[64]; int64_t seen_d[64]; int ns = 0;
        uint32_t pend_c[64]; int32_t pend_bin[64]; int np = 0;      /* (centre, band) of the diagonals that failed without gaps */
        for (uint32_t r = 0; r < cnt[s]; ++r) {
            uint64_t key = sel[(size_t)s * m + r].key;
            uint64_t lo = 0, hi = nc;
            while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ctr[mid].key < key) lo = mid + 1; else hi = mid; }
            uint32_t c = ctr[lo].idx;                       /* first of its key group = longest, lowest index */
            if (c == s) continue;
            int64_t pc = -1;
            for (uint32_t r2 = 0; r2 < cnt[c]; ++r2) if (sel[(size_t)c * m + r2].key == key) { pc = sel[(size_t)c * m + r2].pos; break; }
            int64_t d = pc - (int64_t)sel[(size_t)s * m + r].pos;
            int dup = 0;
            for (int z = 0; z < ns; ++z) if (seen_c[z] == c && seen_d[z] == d) { dup = 1; break; }
            if (dup) continue;
            seen_c[ns] = c; seen_d[ns] = d; ++ns;
            ++n_ver;
            const uint8_t *qc = res + off[c];
            int64_t Lc = (int64_t)(off[c + 1] - off[c]);
            int64_t x0 = d < 0 ? -d : 0, x1 = Ls < Lc - d ? Ls : Lc - d, match = 0;
            for (int64_t x = x0; x < x1; ++x) match += qs[x] == qc[x + d];
            int64_t ovl = x1 - x0;
            if (ovl > 0 && (double)match >= min_id * (double)ovl && (double)ovl >= min_cov * (double)Lc && (double)ovl >= min_cov * (double)Ls) {
                int have = 0;
                for (uint32_t z = 0; z < nacc[s]; ++z) if (acc[(size_t)s * m + z] == c) have = 1;
                if (!have) { acc[(size_t)s * m + nacc[s]++] = c; ++n_acc; }
            } else {
                pend_c[np] = c; pend_bin[np] = (int32_t)((d + DIAG_OFF) / BIN_W); ++np;
            }
        }
        /* gapped verification: for every centre not accepted yet, the best band (score, then lowest bin) among its failed
         * diagonals' bands is aligned with traceback; accepted when identities >= min_id * alignment columns and the aligned
         * span covers min_cov of both sequences */
        for (int a = 0; a < np; ++a) {
            uint32_t c = pend_c[a];
            int first = 1, have = 0;
            for (int b = 0; b < a; ++b) if (pend_c[b] == c) first = 0;
            for (uint32_t z = 0; z < nacc[s]; ++z) if (acc[(size_t)s * m + z] == c) have = 1;
            if (!first || have) continue;
            const uint8_t *qc = res + off[c];
            int32_t Lc = (int32_t)(off[c + 1] - off[c]);
            int32_t best = 0, best_bin = 0;
            for (int b = a; b < np; ++b) {
                if (pend_c[b] != c) continue;
                sw_out o;
                banded_sw(&gp, qs, (int32_t)Ls, qc, Lc, pend_bin[b] * BIN_W - DIAG_OFF - BAND_LEAD, BAND, 0, &o);
                if (o.score > best || (o.score == best && o.score > 0 && pend_bin[b] < best_bin)) { best = o.score; best_bin = pend_bin[b]; }
            }
            if (best < 1) continue;
            int32_t dlo = best_bin * BIN_W - DIAG_OFF - BAND_LEAD, is, js; uint32_t nid, al;
            sw_out o; runbuf rb = {0};
            band_align(&gp, qs, (int32_t)Ls, qc, Lc, dlo, &o, &rb, &is, &js, &nid, &al);
            free(rb.runs);
            double qspan = (double)(o.iend - is + 1), tspan = (double)(o.jend - js + 1);
            if ((double)nid >= min_id * (double)al && qspan >= min_cov * (double)Ls && tspan >= min_cov * (double)Lc) {
                acc[(size_t)s * m + nacc[s]++] = c; ++n_acc;
            }
        }
    }
    /* greedy in priority order */
    lc_pri *pri = malloc((n + 1) * sizeof(lc_pri));
    uint32_t *rank = malloc((n + 1) * sizeof(uint32_t));
    for (uint32_t s = 0; s < n; ++s) { pri[s].len = (uint32_t)(off[s + 1] - off[s]); pri[s].idx = s; }
    qsort(pri, n, sizeof(lc_pri), lc_cmp_pri);
    for (uint32_t i = 0; i < n; ++i) rank[pri[i].idx] = i;
    for (uint32_t s = 0; s < n; ++s) rep_out[s] = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t s = pri[i].idx;
        /* s is a member of its highest-priority accepted centre that is itself a representative; else a representative */
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t z = 0; z < nacc[s]; ++z) {
            uint32_t c = acc[(size_t)s * m + z];
            if (rep_out[c] == c && (best == 0xFFFFFFFFu || rank[c] < rank[best])) best = c;
        }
        rep_out[s] = best == 0xFFFFFFFFu ? s : best;
    }
    if (stats) { stats[0] = n_sel; stats[1] = n_ver; stats[2] = n_acc; }
    free(sel); free(cnt); free(ctr); free(acc); free(nacc); free(pri); free(rank);
    return 0;
}

/* ------------------------------------------------------------------ aligned-allele strings of the genome mapping (K12)
 * restates the per-hit loop of iter_map_bsn (PEPPAN.py:812-835) and the base-5 packing (PEPPAN.py:846-848).
 * PINNED: tests check it against the `bsn` arrays the reference's own iter_map_bsn wrote (tests/golden/g14_mapbsn.json).
 * row:   contig, q_start (1-based), rs/re (1-based, rs > re = reverse strand), CIGAR runs len<<2|op (0=M 1=I 2=D) in nt.
 * per row:  ms = the contig bases under the M columns ('-' for I columns, D columns skipped); reverse rows read the
 *           reverse complement (any letter outside ACGT complements to 'N', configure.py:152-154);
 *           in_frame = max over the three frames of the M columns counted in it (frame shifts: D -> f-n, I -> f+n mod 3);
 *           orf = longest distance between consecutive stop codons of ms read in non-overlapping triples from its start
 *                 (0 and |ms| count as borders);  stops TAG TAA TGA, table 4: TAA TAG.
 * per group: codes[q_len] (A1 C2 G3 T4 else 0) with every row's ms written at q_start-1 in row order (later rows win),
 *           packed[j] = codes[j]*25 + codes[s+j]*5 + codes[2s+j] (0 beyond q_len), s = ceil(q_len/3).
 * returns 0, or -1 on inconsistent input. */
typedef struct { uint32_t contig, q_start, rs, re, cigar_runs, pad; uint64_t cigar_off; } oracle_locus;

static int allele_code(uint8_t ch) { switch (ch) { case 'A': return 1; case 'C': return 2; case 'G': return 3; case 'T': return 4; default: return 0; } }

int oracle_alleles(const uint8_t *nt, const uint64_t *nt_off, uint32_t n_contigs, uint64_t n_rows, const oracle_locus *rows,
                   const uint32_t *cigar, uint32_t n_groups, const uint64_t *grp_off, const uint32_t *grp_qlen, int gtable,
                   int64_t *in_frame, int64_t *orf, uint8_t *packed)
{
    uint64_t pk = 0;
    if (n_groups && grp_off[n_groups] != n_rows) return -1;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const int64_t ql = grp_qlen[g], s = (ql + 2) / 3;
        uint8_t *codes = calloc((size_t)(3 * s + 3), 1);
        for (uint64_t r = grp_off[g]; r < grp_off[g + 1]; ++r) {
            const oracle_locus *L = rows + r;
            if (L->contig >= n_contigs) { free(codes); return -1; }
            const uint8_t *c = nt + nt_off[L->contig];
            const int64_t cl = (int64_t)(nt_off[L->contig + 1] - nt_off[L->contig]);
            const int rev = L->rs > L->re;
            int64_t span = 0, rcons = 0;
            for (uint32_t k = 0; k < L->cigar_runs; ++k) {
                const uint32_t run = cigar[L->cigar_off + k];
                if ((run & 3) != 2) span += run >> 2;
                if ((run & 3) != 1) rcons += run >> 2;
            }
            const int64_t lo = rev ? L->re : L->rs, hi = rev ? L->rs : L->re;
            if (lo < 1 || hi > cl || rcons != hi - lo + 1 || L->q_start < 1 || (int64_t)L->q_start - 1 + span > ql) { free(codes); return -1; }
            uint8_t *ms = malloc((size_t)span + 1);
            int64_t at = 0, o = 0, fr[3] = {0, 0, 0};
            int f = 0;
            for (uint32_t k = 0; k < L->cigar_runs; ++k) {
                const uint32_t run = cigar[L->cigar_off + k];
                const int64_t n = run >> 2;
                const int op = run & 3;
                if (op == 0) {
                    for (int64_t x = 0; x < n; ++x) {
                        int b = rev ? allele_code(c[L->rs - 1 - (at + x)]) : allele_code(c[L->rs - 1 + at + x]);
                        if (rev && b) b = 5 - b;
                        ms[o++] = (uint8_t)b;
                    }
                    at += n; fr[f] += n;
                } else if (op == 2) {
                    at += n; f = (int)(((f - n) % 3 + 3) % 3);
                } else {
                    for (int64_t x = 0; x < n; ++x) ms[o++] = 0;
                    f = (int)((f + n) % 3);
                }
            }
            int64_t best = fr[0] > fr[1] ? fr[0] : fr[1];
            if (fr[2] > best) best = fr[2];
            in_frame[r] = best;
            int64_t prev = 0, longest = 0;
            for (int64_t cd = 0; cd + 3 <= span; cd += 3) {
                const int a = ms[cd], b = ms[cd + 1], d = ms[cd + 2];
                const int stop = a == 4 && ((b == 1 && (d == 1 || d == 3)) || (gtable != 4 && b == 3 && d == 1));
                if (stop) { if (cd - prev > longest) longest = cd - prev; prev = cd; }
            }
            if (span - prev > longest) longest = span - prev;
            orf[r] = longest;
            for (int64_t x = 0; x < span; ++x) codes[L->q_start - 1 + x] = ms[x];
            free(ms);
        }
        for (int64_t j = 0; j < s; ++j) packed[pk + j] = (uint8_t)(codes[j] * 25 + codes[s + j] * 5 + (2 * s + j < ql ? codes[2 * s + j] : 0));
        pk += s;
        free(codes);
    }
    return 0;
}
