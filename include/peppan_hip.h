/*
 * peppan_hip.h - C ABI of libpeppan_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the one data-parallel hot path of PEPPAN: the similarity search that
 * modules/uberBlast.py delegates to the `diamond` executable and the single-linkage grouping
 * of its hits.  Plain pointers and sizes only; every buffer handed IN is caller-owned and only
 * read; every buffer handed OUT is filled into caller-allocated memory (sizes are queried first),
 * except pep_result handles, which are released with pep_result_free.  A pep_result stays valid and unchanged
 * whatever is called on its context afterwards (the newest one is served from the context's pinned staging area and is
 * given its own copy before that area is reused).
 *
 * All functions returning int return PEP_OK (0) or a negative PEP_ERR_* code; the message of the
 * last failure on a context is available from pep_last_error().  Nothing is ever silently dropped:
 * capacity problems are errors, not truncation.
 *
 * Threading: one context per GPU; calls on one context must be serialised by the caller; distinct
 * contexts are independent.  A context must be created in the process that uses it (HIP state does
 * not survive fork(); the reference calls uberBlast from forked pool workers, PEPPAN.py:922).
 *
 * Reference interface each entry point replaces (file:line in zheminzhou/PEPPAN):
 *   pep_set_query_nt + K1   transeq(frame='F') + per-gene frame choice     uberBlast.py:525-529, configure.py:160-194
 *   pep_set_ref_nt   + K1   transeq(frames 6|3) + stop-codon chunking      uberBlast.py:535-544
 *   pep_search              `diamond makedb` + 5 x `diamond blastp ... --outfmt 101`   uberBlast.py:531-533, 546-552
 *   pep_hit fields          the SAM fields parseDiamond consumes (POS, CIGAR, |SEQ|, NM, ZR, ZS)   uberBlast.py:25-58
 *   pep_rescore_nt          cigar2score mode 1 inside RunBlast.reScore      uberBlast.py:226-249, 397-415
 *   pep_components(_of_hits) union-find of get_gene_group (partition only)  PEPPAN.py:1598-1607
 *   pep_linclust            `mmseqs createdb / linclust / createtsv`        clust.py:62-66
 *   pep_overlaps            numba tab2overlaps inside returnOverlap          uberBlast.py:73-97, 378-395
 *   pep_sha1, pep_dedup     hashlib.sha1 per gene + the duplicate collapse of writeGenes   PEPPAN.py:62, 1019, 1023-1039
 *   pep_ovl_filter          RunBlast.ovlFilter (host C++)                    uberBlast.py:417-452
 *   pep_linear_merge        RunBlast.linearMerge + _linearMerge (host C++)   uberBlast.py:100-218, 453-460
 *   pep_alleles             aligned-allele strings + base-5 packing of iter_map_bsn   PEPPAN.py:812-835, 846-848
 *   pep_store_mat_member / pep_store_seq_member   the 1000-group members of the .mat / .seq stores get_map_bsn writes (host C++:
 *                           the .npy pickle stream emitted from the numeric hit table)   PEPPAN.py:950-966
 *   pep_table_from_hits / pep_cols_fix_end / pep_cols_order / pep_cols_gather   RunBlast.run's numeric chain between a search and the caller (host C++:
 *                           parseDiamond / parseBlast columns, fixEnd, the final sort)   uberBlast.py:25-58, 275-290, 375, 462-480
 *   pep_store_tab_members   all members of the .tab store (gene -> int rows) as finished zip entries, host threads   PEPPAN.py:91-113, 972-975
 *   pep_similar_classify / pep_similar_scan / pep_pair_support / pep_similar_resolve   the pass of get_similar_pairs over the all-vs-all table and its
 *                           get_similar (row-local tests, host state machine, K14 on the GPU, host dictionary)   PEPPAN.py:195-224, 231-276, 294
 *   pep_known_order         compare_prediction over a genome's hit table (host C++)   PEPPAN.py:869-901
 */
#ifndef PEPPAN_HIP_H
#define PEPPAN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PEP_ABI_VERSION 16

#define PEP_OK 0
#define PEP_ERR_HIP (-1)       /* a HIP runtime call failed */
#define PEP_ERR_ARG (-2)       /* invalid argument */
#define PEP_ERR_LIMIT (-3)     /* input exceeds a documented limit */
#define PEP_ERR_STATE (-4)     /* call order (e.g. search before sequences were set) */
#define PEP_ERR_INTERNAL (-5)

/* documented limits (candidate key packs q:21 | t:25 | diagonal bin:18) */
#define PEP_MAX_QUERIES ((1u << 21) - 2)
#define PEP_MAX_TARGETS ((1u << 25) - 1)
#define PEP_MAX_SEQ_LEN ((1u << 23) - 256)
#define PEP_MAX_RESIDUES ((1u << 29) - 1)   /* per side, padded */

typedef struct pep_ctx pep_ctx;
typedef struct pep_result pep_result;

/* search parameters; pep_default_params fills the protein defaults that mirror the reference's
 * diamond command line (uberBlast.py:550): BLOSUM62, gap 11/1, --evalue 1, --dbsize 5000000, -k 10, 5 splits */
typedef struct {
    int32_t gap_open, gap_ext;    /* a gap of length k costs gap_open + k * gap_ext; 0 <= gap_open <= 255, 1 <= gap_ext <= 255 */
    int32_t n_shapes, base;       /* spaced seeds over a reduced alphabet of `base` letters */
    int32_t weight[4];
    int32_t offs[4][32];
    uint8_t reduce[32];           /* residue code -> reduced letter; 0xFF never seeds */
    int8_t sub[1024];             /* substitution score [q*32 + t]; row and column 31 (padding code) must be <= -64 */
    double min_id_pct;            /* --id           (100 * identities / alignment length) */
    double min_qcov_pct;          /* --query-cover  (100 * aligned query span / query length) */
    int32_t top_k;                /* -k : targets kept per query per split */
    int32_t n_splits;             /* targets are dealt round-robin into this many splits */
    double dbsize, max_evalue;    /* --dbsize, --evalue */
    int32_t use_lds;              /* 1: residues staged in LDS (default); 0: read from global memory */
    int32_t ungapped_min;         /* a seed hit nominates a candidate only if its ungapped x-drop score reaches this (0 = off) */
    int32_t xdrop;                /* x-drop of that ungapped extension (must stay below 64) */
    int32_t ext_right, ext_left;  /* residues scored right of (from) / left of the seed's first position (<= 48 / <= 48) */
    int32_t reserved[3];          /* profiling/test switches, 0 in production: [0] seed-join stage limiter (1 keys, 2 +buckets, 3 +entries,
                                     9 no wave-level de-duplication of seed hits; 8 the nucleotide tool's plain 17-mer matcher instead of look-up words
                                     at a stride; 10 the plain stream of a self-search instead of dropping diagonal-0 self hits, 11 the filter word asked inside self
                                     stretches too, 12 a bucket's two ends as two loads - same results, for comparison: tools/ab/self_hits_ab.py), [1] != 0 forces the 32-bit Smith-Waterman passes instead
                                     of the packed 16-bit ones, [2] != 0 forces the count -> scan -> fill build of the query seed index */
    double ka_lambda, ka_k;       /* Karlin-Altschul parameters of the scoring system (protein default 0.267 / 0.041) */
    int32_t hsp_mode;             /* 0: one alignment per (q, t), its best band (diamond --max-hsps 1); 1: every band reaching the
                                     score threshold, duplicates (same end cell) removed - several copies on one subject; 2 (UNPINNED: restated from NCBI's published rules, no blastn binary to hold it to): BLAST's way with the
                                     HSPs of a subject (blastn behind uberBlast.py:294): the bands of (q, t) in the order score descending, band ascending,
                                     one dropped when an ACCEPTED one in front shares its start or its end cell or holds its query and subject ranges
                                     inside its own; top_k counts subjects (the reference sequence a target is a strand / frame of), every alignment of
                                     a kept subject stays.  Needs targets made by pep_translate / pep_use_nt_as_residues. */
    int32_t t_index_base;         /* index of this context's target 0 in the WHOLE reference set when the targets are one shard of it (multi-GPU
                                     target sharding, peppan_amd/dist.py): split membership is (t + t_index_base) mod n_splits, so a shard ranks
                                     its hits exactly as the unsharded search would; 0 otherwise.  Ignored with pep_set_target_groups. */
    int32_t stage1_min;           /* first stage of the ungapped pre-filter (only with ungapped_min > 0): the extension to the right of a seed hit must
                                     have reached this score after its first 16 residues - the seed and a few residues behind it - or the hit is
                                     dropped before the rest of its windows is fetched (chance hits of the reduced alphabet); 0 = off */
    int32_t reserved2;            /* test switches (bits), 0 in production: bit 0 makes the alignment stage synchronise with the host after its selection
                                     step and size the traceback buffers exactly (what it does by itself when its upper bounds would cost too much
                                     memory) instead of running from the candidate count to the result sizes without a host round trip; bit 1 makes the
                                     score pass sweep identical pairs like any other pair instead of settling them by comparison
                                     (pep_stats.candidates_settled), bit 2 makes it settle them also in searches below 16 384 candidates, where
                                     the comparison costs more than it saves */
} pep_search_params;

/* one alignment; coordinates are 1-based, inclusive, in residues of the query / target protein */
typedef struct {
    uint32_t q, t;                /* query index, target (chunk) index */
    uint32_t q_start, q_end;      /* ZS .. */
    uint32_t t_start, t_end;      /* POS .. */
    int32_t score;                /* ZR raw score */
    uint32_t nm;                  /* NM = aln_len - n_ident */
    uint32_t n_ident, aln_len;
    uint32_t cigar_runs;
    int32_t bin;                  /* diagonal bin of the band that produced it */
    uint64_t cigar_off;           /* into the result's CIGAR arena: runs of (len << 2 | op), op 0=M 1=I 2=D */
    uint64_t cells;               /* in-band, in-matrix DP cells of this alignment's band */
} pep_hit;

typedef struct { uint32_t seq, frame, aa_len, nt_len; } pep_query_meta;       /* frame 1..3 */
typedef struct { uint32_t seq, frame, chunk_off, aa_len; } pep_target_meta;   /* frame 1..6; name = seq:frame:chunk_off */

/* a nucleotide-level hit for rescoring (uberBlast.py:412): 1-based inclusive nt coordinates, rs > re = reverse strand */
typedef struct {
    uint32_t q, r;
    uint32_t qs, qe, rs, re;
    uint32_t cigar_runs, pad;
    uint64_t cigar_off;
} pep_nt_hit;

typedef struct {
    uint64_t query_residues, target_residues;
    uint64_t query_seeds, target_seeds, seed_hits, seed_hits_passed;   /* seed_hits_passed: runs of seed hits whose candidate passed the ungapped filter or was known already */
    uint64_t candidates;          /* unique (q, t, band) */
    uint64_t pairs;               /* unique (q, t) */
    uint64_t tracebacks;
    uint64_t hits;
    uint64_t cells;               /* in-band in-matrix DP cells over all candidates, score pass (SW cell updates) */
    uint64_t cells_swept;         /* 64 lanes x anti-diagonal steps executed by the score pass */
    uint64_t dir_bytes;           /* traceback direction workspace written by the SW kernel */
    uint64_t sw_launches;
    uint64_t cells_trace;         /* DP cells recomputed by the traceback pass (selected pairs only) */
    uint64_t cells_swept_trace;   /* 64 lanes x steps executed by the traceback pass */
    uint64_t tracebacks_gapless;  /* of `tracebacks`: pairs whose alignment is one ungapped run, settled without a traceback sweep (rule 5a) */
    uint64_t candidates_settled;  /* of `candidates`: identical pairs the score pass settles without a sweep (same residues, band over diagonal 0, every
                                     residue dominant in the score table: the optimum is the whole diagonal) */
    uint64_t cells_settled;       /* of `cells`: the cells of those candidates (cells - cells_settled = cells the score pass sweeps) */
    double ms_seed, ms_sw, ms_trace, ms_total;   /* HIP-event times on the context's stream; ms_sw = score pass kernel */
    double ms_k1, ms_sw_trace;                   /* ms_sw_trace = traceback-pass kernel (selected pairs only) */
    double ms_seed_match;                        /* seed_match kernel, summed over the seed shapes (one launch per shape) */
    double ms_reserved[3];
} pep_stats;

int pep_version(void);
int pep_device_count(void);
int pep_ctx_create(int device, pep_ctx **out);
void pep_ctx_destroy(pep_ctx *ctx);
const char *pep_last_error(const pep_ctx *ctx);
void pep_default_params(pep_search_params *p);
/* sensitivity of the translated search: 0 = DIAMOND's two default-mode seed shapes (what the reference's command line runs, uberBlast.py:550),
 * 1 = four shapes (recall 0.93 -> 0.985 between 0.45 and 0.7 identity at twice the seed-stage cost).  Rewrites n_shapes / weight / offs. */
int pep_set_sensitivity(pep_search_params *p, int level);
/* smallest raw score passing the e-value cut for a query of qlen residues */
int32_t pep_min_score(uint32_t qlen, double dbsize, double max_evalue);
int32_t pep_min_score_ka(uint32_t qlen, double dbsize, double max_evalue, double ka_lambda, double ka_k);

/* nucleotide inputs (ASCII, any case), concatenated, off[n+1].  Stored on the device; K1 (translation,
 * frame choice / chunking, packing) runs on the GPU at the next pep_translate or pep_search. */
int pep_set_query_nt(pep_ctx *ctx, const uint8_t *nt, const uint64_t *off, uint32_t n, int gtable);
int pep_set_ref_nt(pep_ctx *ctx, const uint8_t *nt, const uint64_t *off, uint32_t n, int frames /* 6 or 3 */, int gtable);
/* protein inputs (residue code = letter - 'A'), used as they are (no K1) */
int pep_set_query_aa(pep_ctx *ctx, const uint8_t *codes, const uint64_t *off, uint32_t n);
int pep_set_ref_aa(pep_ctx *ctx, const uint8_t *codes, const uint64_t *off, uint32_t n);
/* run K1 for the sides given as nucleotides (idempotent until the inputs change; force != 0 re-runs it) */
int pep_translate(pep_ctx *ctx, int force);
/* the protein sets made from nucleotide inputs are stale: the next pep_search translates again (K1), inside the search - both sides queued
 * at once and the seed stage's first kernels behind them, so that the GPU does not wait for the host between K1 and the search as it does
 * with a pep_translate(force) call in front */
int pep_invalidate_translation(pep_ctx *ctx);
/* The nucleotide search (the reference's blastn call, uberBlast.py:294, 482-509) on device-resident inputs: the nucleotide sets given to
 * pep_set_query_nt / pep_set_ref_nt THEMSELVES become the residue sets of the following searches, as base codes A0 C1 G2 T3 (anything
 * else 4), packed on the GPU - queries forward; the reference, per reference set (pep_set_target_groups, else the whole list), all
 * forward strands followed (strands = 2) by all reverse complements.  Target meta: seq = reference sequence, frame 1 forward / 4 reverse.
 * Stays in force until the next pep_translate or pep_set_*: pep_search does not run K1 in between, and a translated search afterwards
 * needs pep_translate first.  PEP_ERR_LIMIT for a sequence beyond PEP_MAX_SEQ_LEN (the caller then windows it, peppan_amd/uberBlast.py). */
int pep_use_nt_as_residues(pep_ctx *ctx, int strands);
int pep_query_count(pep_ctx *ctx, uint32_t *n, uint64_t *residues);
int pep_target_count(pep_ctx *ctx, uint32_t *n, uint64_t *residues);
int pep_get_query_meta(pep_ctx *ctx, pep_query_meta *out, uint32_t cap);
int pep_get_target_meta(pep_ctx *ctx, pep_target_meta *out, uint32_t cap);
/* download the packed proteins (plain concatenation, off[n+1]) - for tests and debugging */
int pep_get_query_aa(pep_ctx *ctx, uint8_t *codes, uint64_t cap, uint64_t *off);
int pep_get_target_aa(pep_ctx *ctx, uint8_t *codes, uint64_t cap, uint64_t *off);

/* Optional: batch several reference sets (e.g. genomes) into one search.  group[i] = set of reference sequence i (sequences
 * of one set must be contiguous).  The top-k / split competition then runs inside each set, with targets numbered from 0
 * inside it, so the hits of a set are exactly those of searching it alone.  n = 0 clears the grouping.  Call after the
 * reference sequences are set; a new pep_set_ref_* clears it. */
int pep_set_target_groups(pep_ctx *ctx, const uint32_t *group, uint32_t n);

/* on_device != 0: the searches of this context leave their hit table on the device (pep_result_device) and skip the copy to the host at
 * the end of pep_search; pep_result_copy / pep_result_data then fetch it on demand, while the result is still the context's newest
 * (PEP_ERR_STATE afterwards).  For callers that go on working on the GPU: the all-gather of a multi-GPU search, K10. */
int pep_set_result_mode(pep_ctx *ctx, int on_device);
/* Phase timers of the searches of this context (the ms_* fields of pep_stats): HIP events recorded on the search's stream, each of which
 * leaves the GPU idle for about 6 us between the two kernels it separates.  level 0 (the default): none, the fields stay 0;
 * 1: the Smith-Waterman score pass only (ms_sw); 2: every phase (sixteen events per search).  ms_k1 (pep_translate) follows the same switch. */
int pep_set_timing(pep_ctx *ctx, int level);
/* K2..K8: seeds, candidates, banded Smith-Waterman, traceback, filters, top-k.  Hits ordered by (q, t). */
int pep_search(pep_ctx *ctx, const pep_search_params *params, pep_result **out);
int pep_result_size(const pep_result *r, uint64_t *n_hits, uint64_t *n_cigar);
int pep_result_copy(const pep_result *r, pep_hit *hits, uint32_t *cigar);
/* zero-copy access: pointers to the n_hits records / n_cigar runs of pep_result_size, valid until the next call on the result's
 * context that produces a hit table or re-translates (pep_search, pep_translate, pep_linclust), pep_ctx_destroy or pep_result_free,
 * whichever comes first: they point into the context's pinned staging area, which such calls may re-use or re-allocate
 * (use pep_result_copy for anything that must outlive that) */
int pep_result_data(const pep_result *r, const pep_hit **hits, const uint32_t **cigar);
/* device copy of the table (same records, same order), for callers that go on working on the GPU - an all-gather from device memory, K10:
 * pointers into the context's workspace, valid (PEP_OK) only while r is its context's NEWEST result and nothing has reused that workspace
 * (the next pep_search / pep_linclust does); PEP_ERR_STATE and null pointers otherwise */
int pep_result_device(const pep_result *r, const pep_hit **d_hits, const uint32_t **d_cigar);
int pep_result_stats(const pep_result *r, pep_stats *stats);
void pep_result_free(pep_result *r);

/* Host-side (no GPU work, no context): merge of the hit tables of several TARGET shards of one search (multi-GPU, after the
 * all-gather).  hits carry global q and t indices; every (q, t) lives in exactly one shard, each shard already applied top-k per
 * (q, split) locally, and the global top-k of a (q, split) is contained in the union of the local ones.  Ranks inside
 * (q, t mod n_splits) by (score desc, t asc, bin asc) - the order of K8 - keeps rank < top_k and writes the survivors ordered by
 * (q, t, bin) with a compacted CIGAR arena.  out_hits / out_cigar must hold n / n_cigar entries. */
int pep_merge_hits(uint64_t n, const pep_hit *hits, const uint32_t *cigar, uint64_t n_cigar, int32_t top_k, int32_t n_splits,
                   pep_hit *out_hits, uint32_t *out_cigar, uint64_t *n_out, uint64_t *n_cigar_out);

/* K7: integer counts of mode-1 rescoring per hit, out[5*i..] = nMatch, nMismatch, nGap, bGap, mGap.
 * Uses the nucleotide sets given to pep_set_query_nt / pep_set_ref_nt. */
int pep_rescore_nt(pep_ctx *ctx, uint64_t n, const pep_nt_hit *hits, const uint32_t *cigar, uint64_t n_cigar, int64_t *out);

/* K7 as the tail of every search of this context (-s 1 in PEPPAN's hot call, PEPPAN.py:229-230: every hit of both tools is rescored, uberBlast.py:352-353):
 * with on = 1 pep_search ends with the count of identical nucleotide columns of every hit it emits - K7's n_match, the one of its five counts that needs the
 * sequences - computed from the table on the device (the row a hit becomes follows from the hit and the descriptors of its packed sequences; nothing is
 * uploaded again, same stream, same wait).  pep_result_nt_match hands out the counts, one per hit in hit order (NULL when the search had none or the switch
 * was off); pep_table_from_hits turns them into the rescored identity and score.  Needs packed sets made from the context's nucleotide sets (K1 or
 * pep_use_nt_as_residues): a search over sets given as residues fails with PEP_ERR_STATE while the switch is on. */
int pep_set_nt_match(pep_ctx *ctx, int on);
int pep_result_nt_match(const pep_result *r, const uint32_t **nt_match);

/* K10: connected components; label[x] = smallest node id of x's component */
int pep_components(pep_ctx *ctx, uint32_t n_nodes, uint64_t n_edges, const uint32_t *a, const uint32_t *b, uint32_t *label);
/* the same over the edges of a hit table: (hit.q + q_base, node_of_target[hit.t]) - the single-linkage step of get_gene_group on
 * the table of the all-vs-all search (PEPPAN.py:1598-1607), without building the two edge columns on the caller's side */
int pep_components_of_hits(pep_ctx *ctx, uint32_t n_nodes, uint64_t n_hits, const pep_hit *hits, uint32_t q_base,
                           const uint32_t *node_of_target, uint64_t n_targets, uint32_t *label);

/* the same for a search result: while r is the context's newest result its device copy is read in place (no edge columns are built or
 * uploaded; node_of_target is uploaded only when it changed since the last call), otherwise as pep_components_of_hits */
/* single linkage as the tail of every search of this context (the north-star path: all-vs-all search + grouping, PEPPAN.py:229-230 +
 * 1598-1607): with n_nodes > 0 pep_search ends with K10 over the edges (hit.q + q_base, node_of_target[hit.t]) of the table it has just
 * produced - same stream, no second call, no second wait - and pep_result_labels hands out label[g] = smallest node of g's component.
 * node_of_target needs an entry for every target of the searches that follow (else they fail with PEP_ERR_STATE); n_nodes = 0 switches it off. */
int pep_set_grouping(pep_ctx *ctx, uint32_t n_nodes, uint32_t q_base, const uint32_t *node_of_target, uint64_t n_targets);
int pep_result_labels(const pep_result *r, uint32_t *label, uint32_t n_nodes);
int pep_components_of_result(pep_ctx *ctx, const pep_result *r, uint32_t n_nodes, uint32_t q_base, const uint32_t *node_of_target, uint64_t n_targets,
                             uint32_t *label);

/* K9: linear-time clustering.  codes: residue codes (< base are valid k-mer letters), concatenated, off[n+1].
 * rep[i] = index of sequence i's representative (rep[i] == i for representatives).  stats (may be NULL):
 * [0] selected k-mers, [1] verified (sequence, centre, diagonal) pairs, [2] accepted edges.
 * Pairs that fail on their k-mer diagonal without gaps are aligned by the search's banded Smith-Waterman engine (base 4:
 * +2 / -3, gap 6 + 2k; otherwise residues in the order ACDEFGHIKLMNPQRSTVWY + one code for anything else, BLOSUM62, gap 11 + k).
 * That stage uses the context's packed sequence sets: after pep_linclust a pep_search re-runs K1 for nucleotide inputs and
 * needs pep_set_query_aa / pep_set_ref_aa inputs to be given again (PEP_ERR_STATE otherwise). */
int pep_linclust(pep_ctx *ctx, const uint8_t *codes, const uint64_t *off, uint32_t n, int base, int k, int m,
                 double min_id, double min_cov, uint32_t *rep, uint64_t *stats);

/* K11: overlapping reference intervals (flag -O; tab2overlaps, uberBlast.py:73-97).  Rows sorted by (contig, start, end).
 * out receives (id1, id2, overlap) triples in (i, j) order; *n_pairs is the full count - when it exceeds cap nothing is
 * written and the caller calls again with a buffer of *n_pairs triples. */
int pep_overlaps(pep_ctx *ctx, uint64_t n, const int32_t *contig, const int64_t *start, const int64_t *end, const int64_t *row_id,
                 double ovl_l, double ovl_p, int64_t *out, uint64_t cap, uint64_t *n_pairs);

/* K12: aligned alleles of the genes->genomes mapping (iter_map_bsn, PEPPAN.py:812-835, 846-848).
 * A row is one aligned locus of a gene on a contig; rows of one gene group are contiguous, group g owning rows
 * [grp_off[g], grp_off[g+1]) in the order the reference writes them (later rows overwrite earlier ones where they share
 * query positions).  nt / nt_off[n_contigs+1]: the contigs, ASCII, upper case.  CIGAR runs are len<<2|op (0=M 1=I 2=D), nt.
 * Outputs: in_frame[row] = most M columns falling into one codon frame; orf[row] = longest stretch of the allele string
 * between stop codons (gtable 4: TAA TAG, else TAG TAA TGA); packed = for every group ceil(q_len/3) bytes
 * codes[j]*25 + codes[s+j]*5 + codes[2s+j] over the gene's coordinate system (A1 C2 G3 T4, anything else 0), groups
 * concatenated in order.  packed_cap = bytes available in `packed`. */
typedef struct {
    uint32_t contig;          /* index into nt_off */
    uint32_t q_start;         /* 1-based first query position of the alignment (column 6) */
    uint32_t rs, re;          /* 1-based reference start / end (columns 8, 9); rs > re = reverse strand */
    uint32_t cigar_runs, group;
    uint64_t cigar_off;
} pep_locus;
int pep_alleles(pep_ctx *ctx, const uint8_t *nt, const uint64_t *nt_off, uint32_t n_contigs, uint64_t n_rows, const pep_locus *rows,
                const uint32_t *cigar, uint64_t n_cigar, uint32_t n_groups, const uint64_t *grp_off, const uint32_t *grp_qlen, int gtable,
                int64_t *in_frame, int64_t *orf, uint8_t *packed, uint64_t packed_cap);

/* K14 and its two host passes: the consumer of the all-vs-all table (get_similar_pairs, PEPPAN.py:194-294).
 *
 * pep_similar_scan (host, no context): ONE ordered pass over the table as RunBlast.run returns it (sorted by query, reference, score).
 *   q / r: gene codes in [0, n_genes) that keep the order of the gene ids.  action[k]: the row-local classification of PEPPAN.py:246-260
 *   (PEP_ROW_*), forward[k] = column 8 < column 9, iden4[k] = int(identity * 10000).  The pass keeps the reference's state - genes absorbed
 *   by a near-identical partner or found repetitive are dead from then on, forward rows of one (q, r) pair are collected until a row of
 *   another pair survives - and reports: alive[g] (0 = absorbed / repetitive), seen_as_query[g], the absorbed edges (kept, dropped, iden4)
 *   in order, and the events that write ortho_pairs, in order: PEP_EVENT_CONFLICT (pair a < b gets -2, PEPPAN.py:249) or
 *   PEP_EVENT_SUPPORT (pair a < b is to be judged by get_similar over rows ev_rows[ev_row_off[e] .. ev_row_off[e+1]), PEPPAN.py:268-276).
 *   Capacities: absorbed 3 n, ev_* n + 1 (ev_row_off n + 2), ev_rows n.
 * pep_pair_support (K14): get_similar (PEPPAN.py:195-224) for many groups of forward alignments at once.  Group g = rows
 *   [grp_off[g], grp_off[g+1]) in table order, all of one (query, reference) pair of lengths grp_qlen / grp_rlen.  value[g] =
 *   PEP_SUPPORT_NONE (no decision), 0 (similar but too short a support) or int(mean identity * 10000), with the mean taken over the
 *   covered query positions exactly as numpy.mean takes it (pairwise summation in first-cover order).  At most 255 rows per group.
 * pep_similar_resolve (host, no context): the dictionary ortho_pairs from the events and the values K14 returned for them (ev_value[e]
 *   is ignored for conflicts): out = (a, b, value) triples with value != 0 in first-insertion order (capacity 3 n_events). */
#define PEP_ROW_ORDINARY 0
#define PEP_ROW_CONFLICT 1
#define PEP_ROW_ABSORB_QUERY 2
#define PEP_ROW_ABSORB_REF 3
#define PEP_EVENT_CONFLICT 0
#define PEP_EVENT_SUPPORT 1
#define PEP_SUPPORT_NONE (-2147483647 - 1)
typedef struct {
    uint32_t q_start, r_start;    /* columns 6 and 8: 1-based first query / reference nucleotide (forward alignments only) */
    uint32_t cigar_runs, pad;
    uint64_t cigar_off;           /* runs of (len << 2 | op) in nucleotides, op 0=M 1=I 2=D */
    double identity;              /* column 2 */
} pep_support_row;
typedef struct {
    double match_len[3], match_prop[3];   /* params match_len, match_len1, match_len2 / match_prop, match_prop1, match_prop2 (PEPPAN.py:211, 214-216) */
    double identity_x1e4;                 /* params match_identity * 10000 (PEPPAN.py:213) */
    int32_t any_frame, pad;               /* 'f' in params incompleteCDS (PEPPAN.py:205) */
} pep_support_limits;
/* pep_similar_classify (host, no context; ABI 14): the row-local tests of PEPPAN.py:244-263 over the numeric columns, one pass - what pep_similar_scan takes
 *   as action / forward / iden4.  q / r: gene codes; rank_ge[k] / rank_le[k]: priority rank of row k's query gene >= / <= that of its reference gene
 *   (PEPPAN.py:252, 257: the caller compares whatever type its priorities have); near_identity = clust_identity, cover = clust_match_prop.
 *   Float arithmetic as the reference's float() conversions make it (IEEE double, sqrt(cover) correctly rounded). */
int pep_similar_classify(uint64_t n, const int64_t *q, const int64_t *r, const double *iden, const int64_t *qs, const int64_t *qe, const int64_t *ss,
                         const int64_t *se, const int64_t *ql, const int64_t *sl, const uint8_t *rank_ge, const uint8_t *rank_le, double near_identity,
                         double cover, uint8_t *action, uint8_t *forward, int32_t *iden4);
int pep_similar_scan(uint64_t n, const int64_t *q, const int64_t *r, const uint8_t *action, const uint8_t *forward, const int32_t *iden4, uint64_t n_genes,
                     uint8_t *alive, uint8_t *seen_as_query, int64_t *absorbed, uint64_t *n_absorbed,
                     uint8_t *ev_kind, int64_t *ev_a, int64_t *ev_b, uint64_t *ev_row_off, uint64_t *ev_rows, uint64_t *n_events);
int pep_pair_support(pep_ctx *ctx, uint64_t n_rows, const pep_support_row *rows, const uint32_t *cigar, uint64_t n_cigar, uint64_t n_groups,
                     const uint64_t *grp_off, const uint32_t *grp_qlen, const uint32_t *grp_rlen, const pep_support_limits *lim, int32_t *value);
int pep_similar_resolve(uint64_t n_events, const uint8_t *ev_kind, const int64_t *ev_a, const int64_t *ev_b, const int32_t *ev_value,
                        int64_t *out, uint64_t *n_out);
/* the exemplar rewrite at the end of get_similar_pairs (PEPPAN.py:278-288; host, no context): the FASTA file `path` keeps the records whose
 * name (first token of the header line, a decimal integer) is in ids[0..n_ids) (sorted ascending), verbatim; the file is not touched when
 * every record stays.  PEP_ERR_ARG (file untouched) when a name is not a plain decimal integer: the caller then applies its own rules. */
int pep_fasta_keep(const char *path, const int64_t *ids, uint64_t n_ids, uint64_t *n_records, uint64_t *n_kept);
/* the clusterer's input (the FASTA file clust.py:62-66 hands to `mmseqs createdb`; host, no context): the sequences of the FASTA text
 * data[0..n) as codes[] = table[byte] with off[0 .. *n_records] the start of each record's codes.  A record starts at a '>' at the start of a
 * line, its first line is the header; body lines starting with '#' are dropped, ASCII blanks removed.  codes must hold n bytes, off cap + 1
 * entries; PEP_ERR_LIMIT when the text has more than cap records.  *non_ascii = 1 when a body holds a byte >= 0x80 (the caller then applies
 * its own, Unicode-aware rules). */
int pep_fasta_scan(const uint8_t *data, uint64_t n, const uint8_t *table, uint8_t *codes, uint64_t *off, uint64_t cap, uint64_t *n_records,
                   int32_t *non_ascii);
/* the same pass for the sequence reader of the search (readFasta, configure.py:118-128 of the reference, behind uberBlast.py:339-341): also the records' names -
 * name_off[r] / name_len[r] = the first blank-delimited token of record r's header line inside data (length 0: a header without a name).
 * name_off and name_len hold cap entries. */
int pep_fasta_records(const uint8_t *data, uint64_t n, const uint8_t *table, uint8_t *codes, uint64_t *off, uint64_t *name_off, uint32_t *name_len,
                      uint64_t cap, uint64_t *n_records, int32_t *non_ascii);

/* K13: exact-duplicate collapse of gene instances (front end of the clustering path).
 * pep_sha1: digest[20*i..] = SHA-1 of sequence i (bytes[off[i]..off[i+1])), big-endian bytes as hashlib.sha1(seq).digest();
 *   PEPPAN keys duplicates with int(hexdigest, 16) of it (PEPPAN.py:62, 1019).
 * pep_dedup: writeGenes (PEPPAN.py:1023-1039) over n genes ALREADY in priority order: rep[i] = index of the first gene j <= i
 *   with len[j..i] all equal (the same "length run": the reference forgets what it has seen whenever a different length shows
 *   up, PEPPAN.py:1032-1033) and the same digest; rep[i] == i for a gene that is written out. */
int pep_sha1(pep_ctx *ctx, const uint8_t *bytes, const uint64_t *off, uint32_t n, uint8_t *digest);
int pep_dedup(pep_ctx *ctx, uint32_t n, const uint32_t *len, const uint8_t *digest, uint32_t *rep);

/* Host-side C++ (no GPU work, no context): the order-dependent greedy filters of the genome mapping.
 * Both take the numeric columns of the hit table ALREADY SORTED the way the reference sorts it, with reverse-strand
 * reference coordinates negated (uberBlast.py:420, 455): q / r = integer codes of the query / reference names.
 *
 * pep_ovl_filter (flag -f, RunBlast.ovlFilter uberBlast.py:417-452): rows sorted by (r, q, ss, qs).  iden is in/out:
 * a dropped row gets iden = -1.
 *
 * pep_linear_merge (flag -m, RunBlast.linearMerge + _linearMerge uberBlast.py:100-218, 453-460): rows sorted by
 * (q, r, ss, qs).  Per query (maximal run of equal q) it reports
 *   keep_seq[query_off[k] .. query_off[k+1])  row indices that survive, in the insertion order of the reference's `used`
 *                                             dictionary (duplicates possible); query_ascending[k] = 1 when nothing was chained
 *                                             and every row is kept in order.  The reference builds a Python set from this
 *                                             sequence and emits rows in the set's iteration order; the caller does the same.
 *   per row i: grp_score/grp_iden/grp_span[i] and grp_ids[grp_ids_off[i] .. grp_ids_off[i+1]) = column 16 of the reference
 *                                             ([score, identity, span, row ids...]); grp_span = -1 for rows without a group.
 * n_keep / n_ids return the needed sizes; when they exceed keep_cap / ids_cap nothing is written and the caller calls again. */
int pep_ovl_filter(uint64_t n, const int64_t *q, const int64_t *r, const int64_t *qs, const int64_t *qe, const int64_t *ss, const int64_t *se,
                   const double *score, double *iden, double coverage, double delta);
/* pep_known_order (host, no context; ABI 15): compare_prediction (PEPPAN.py:869-901) over the columns of one genome's hit table.  For every hit the largest fraction of an
 * original gene of the same contig that it covers in frame and on the same strand (0.1 when there is none): the genes of contig c (row code ri) are
 * g1 / g2 / g_plus [g_off[c], g_off[c + 1]) - start, end, strand '+' - in the store's order; g_sorted[c] != 0: in start order (every gene between the first whose
 * running maximum of ends reaches the hit and the last that starts at or before its end is judged), else the reference's forward-only pointer sweep.  The table is
 * walked by (contig code, lower reference coordinate) and returned in the order (query code, contig code, score), each stable: order[k] = the row that comes k-th,
 * known[k] = its value. */
int pep_known_order(uint64_t n, const int64_t *ri, const int64_t *r_code, const int64_t *q_code, const int64_t *ss, const int64_t *se, const int64_t *qs,
                    const int64_t *qe, const int64_t *ql, const double *score, uint64_t n_contigs, const uint64_t *g_off, const int64_t *g1, const int64_t *g2,
                    const uint8_t *g_plus, const uint8_t *g_sorted, int64_t *order, double *known);
int pep_linear_merge(uint64_t n, const int64_t *q, const int64_t *r, const double *iden, const int64_t *qs, const int64_t *qe, const int64_t *ss,
                     const int64_t *se, const double *score, const int64_t *ql, const int64_t *sl, const int64_t *rid, double gap_dist, double len_diff,
                     int64_t *keep_seq, uint64_t keep_cap, uint64_t *n_keep, uint64_t *query_off, uint8_t *query_ascending, uint64_t *n_query,
                     double *grp_score, double *grp_iden, int64_t *grp_span, uint64_t *grp_ids_off, int64_t *grp_ids, uint64_t ids_cap, uint64_t *n_ids);

/* Members of the genome-mapping stores (PEPPAN.py:950-966: 1000 groups per member of <prefix>.mat.npz / <prefix>.seq.npz), emitted as
 * the pickle stream numpy writes for an object array, straight from numeric columns - no Python object is made for a stored row.
 * Host C++, no context.
 *   pep_store_mat_member: the stream of ndarray(object)[n_groups] whose element g is ndarray(object)[row_off[g+1] - row_off[g], 16]
 *     of the hit rows [row_off[g], row_off[g+1]) of `cols`: columns 0-15 of the reference's table with q / r
 *     as integers (PEPPAN's encoded names), the CIGAR as text ("150M3D150M", uberBlast.py:480), score as int when score_is_int.
 *   pep_store_seq_member: ndarray(object)[n_groups] of ndarray(uint8) = packed[pack_off[g] .. pack_off[g+1]) (the base-5 packed
 *     alleles, PEPPAN.py:846-848).
 * recon_module: the module that holds numpy's `_reconstruct` ("numpy._core.multiarray" / "numpy.core.multiarray").
 * Both return the length of the stream; when it exceeds `cap` the buffer holds nothing usable and the caller calls again with a
 * larger one (cap = 0 measures).  Negative: PEP_ERR_ARG. */
typedef struct pep_mat_cols {
    const int64_t *q, *r;
    const double *iden;
    const int64_t *aln, *mis, *gap, *qs, *qe, *ss, *se;
    const double *evalue, *score;
    const int64_t *ql, *sl;
    const uint32_t *arena;          /* CIGAR runs len << 2 | op (0 M, 1 I, 2 D), nucleotide units */
    const int64_t *c_off, *c_runs, *rid;
    int32_t score_is_int, reserved;
} pep_mat_cols;
int64_t pep_store_mat_member(const pep_mat_cols *cols, const int64_t *row_off, int64_t n_groups, const char *recon_module, uint8_t *out, int64_t cap);
int64_t pep_store_seq_member(const uint8_t *packed, const int64_t *pack_off, int64_t n_groups, const char *recon_module, uint8_t *out, int64_t cap);

/* The gene table store <prefix>.tab.npz (PEPPAN.py:972-975 through MapBsn.update, PEPPAN.py:91-113) written into an EMPTY archive: the
 * complete zip entries - local file header + payload - of all members back to back in `out`, made by up to `threads` host threads.
 * Member m is the .npy file of int64[off[m+1] - off[m], n_cols] = rows [off[m], off[m+1]) of the row-major table - of the table taken in the
 * order `order` when that is not NULL (row i of the sorted table = row order[i] of `rows`: the gather happens here, by the threads) -, named by the decimal
 * key[m]; stored below 4 KiB, raw deflate (level 1) from there on - what MapBsn writes member by member.  crc / csize / usize / at[m]
 * (offset of the entry in `out`) are what the archive's central directory needs; the caller appends `out` to the archive and lists
 * the entries.  Returns the length of `out`'s content; when it exceeds `cap` nothing usable was written and the caller calls again
 * with that much room.  Negative: PEP_ERR_ARG. */
int64_t pep_store_tab_members(const int64_t *rows, int64_t n_cols, const int64_t *order, const int64_t *off, const int64_t *key, int64_t n_members, uint32_t dos_time, uint32_t dos_date,
                              int32_t threads, uint8_t *out, int64_t cap, uint32_t *crc, int64_t *csize, int64_t *usize, int64_t *at);

/* The same members as a COMPLETE archive: entries, central directory, end record - what <prefix>.tab.npz is when it is written once (no zipfile
 * object is needed for a store that is then closed).  Plain zip: PEP_ERR_LIMIT from 65 535 members or 4 GiB on (the caller then writes member-wise).
 * Returns the archive's length; when it exceeds `cap` nothing usable was written. */
int64_t pep_store_tab_archive(const int64_t *rows, int64_t n_cols, const int64_t *order, const int64_t *off, const int64_t *key, int64_t n_members, uint32_t dos_time, uint32_t dos_date,
                              int32_t threads, uint8_t *out, int64_t cap);

/* ---- RunBlast.run's numeric chain between a search and the caller, host C++ (no GPU work, no context).  The columns of the reference's hit table
 * ("blastab": SURVEY.md section 8) as flat arrays - every column int64 or double, one CIGAR arena of runs len << 2 | op in nucleotide units. */
typedef struct pep_hit_cols {
    int64_t *qi, *ri;               /* indices into the caller's query / reference name tables (columns 0, 1) */
    double *iden;                   /* column 2 */
    int64_t *aln, *mis, *gap, *qs, *qe, *ss, *se;    /* columns 3 - 9 */
    double *evalue, *score;         /* columns 10, 11 */
    int64_t *ql, *sl;               /* columns 12, 13 */
    int64_t *c_off, *c_runs;        /* column 14: the row's runs in the arena */
    int64_t *rid;                   /* column 15 (may be NULL where a function only fills it with -1) */
} pep_hit_cols;

/* Hit records of a search -> the rows the reference's parsers keep, as columns (room for n rows each); returns the number of rows, in hit order.
 * tool 0, the translated search: parseDiamond's algebra (uberBlast.py:25-58) - names q:frame / r:frame:offset from q_meta / t_meta, CIGAR x 3
 *   (arena_out takes all n_cigar runs, rewritten in nucleotides), identity 1 - round(3 NM / length, 3), coordinates on either strand of the
 *   reference, the cuts qm * 3 >= min_cov, qm * 3 / ql >= min_ratio, identity >= min_id.  t_seq .. evalue unused (NULL).
 * tool 1, the nucleotide search: parseBlast's columns (uberBlast.py:275-290): t_seq / t_rev = reference sequence and strand of every target, identity
 *   with blastn's three printed decimals, mismatches = length - identities - gap columns, `evalue[i]` per HIT as the caller computed it, the cuts
 *   identity >= min_id, aligned query span >= min_cov and >= min_ratio * ql; win_off / home_lo / home_hi (NULL or per target): targets that are
 *   windows of a long strand - a hit is shifted to strand coordinates and kept by the window whose home stretch holds its midpoint.  q_meta / t_meta unused.
 * q_len / r_len: nucleotide lengths per sequence.  Negative: PEP_ERR_ARG.
 * nt_match (NULL, or per HIT the count pep_result_nt_match hands out): the rows come out rescored - RunBlast.reScore's mode 1 (uberBlast.py:397-415,
 *   cigar2score :226-249) applied to every kept row: identity = matches / (matches + mismatches + gap bases - gap bases of gaps longer than 3) and
 *   score = 3 matches - mismatches - 5 gaps - gap bases in float64, rounded to three decimals; the cuts still look at the tool's own identity. */
int64_t pep_table_from_hits(int32_t tool, uint64_t n, const pep_hit *hits, const uint32_t *cigar, uint64_t n_cigar, const pep_query_meta *q_meta,
                            const pep_target_meta *t_meta, const int64_t *q_len, const int64_t *r_len, const int64_t *t_seq, const uint8_t *t_rev,
                            const int64_t *win_off, const int64_t *home_lo, const int64_t *home_hi, const double *evalue, double min_id, double min_cov,
                            double min_ratio, pep_hit_cols *out, uint32_t *arena_out, const uint32_t *nt_match);

/* RunBlast.fixEnd (uberBlast.py:462-480) over all rows, in place: an alignment is stretched over an unaligned query head of at most se_lim / tail of
 * at most ee_lim bases as far as the reference sequence allows, its first / last CIGAR run growing by the same amount.  The rows' runs are copied
 * into arena_out (sum of c_runs words; rows may share runs in arena_in) and c_off is rewritten.  Returns the number of rows that changed;
 * PEP_ERR_ARG also for a row without runs that would have to be extended (the reference fails there). */
int64_t pep_cols_fix_end(uint64_t n, pep_hit_cols *cols, const uint32_t *arena_in, uint64_t n_arena_in, uint32_t *arena_out, double se_lim, double ee_lim);

/* The sort that ends RunBlast.run (uberBlast.py:375: DataFrame.sort_values([0, 1, 11])): order[k] = the row that comes k-th by (q_code, r_code,
 * score), stable; the codes are non-negative integers that sort like the names of columns 0 and 1 do (the caller ranks the name tables once). */
int pep_cols_order(uint64_t n, const int64_t *q_code, const int64_t *r_code, const double *score, int64_t *order);

/* order = numpy.lexsort(keys) for n_keys int64 key columns of n rows: stable, the LAST key the primary one - the sorts in front of RunBlast.ovlFilter (uberBlast.py:421:
 * by reference, query, start, query start) and _linearMerge (:455) of every genome's table in the mapping path.  Radix passes over (key - its minimum); PEP_ERR_LIMIT
 * when a key's range needs more than 44 bits (the caller sorts with numpy then). */
int pep_lex_order(uint64_t n, int32_t n_keys, const int64_t *const *keys, int64_t *order);

/* dst[c][k] = src[c][idx[k]] for n_cols columns of 8-byte elements (n_src rows each): the rows `idx` of a whole table in one call */
int pep_cols_gather(int32_t n_cols, const void *const *src, void *const *dst, const int64_t *idx, uint64_t n_idx, uint64_t n_src);

/* Host threads: the passes above (pep_table_from_hits, pep_cols_fix_end, pep_cols_gather) split tables of 16 384 rows and more over up to `n` threads that
 * live for the call (1 .. 16).  Default: the environment's PEPPAN_HOST_THREADS, else a quarter of the hardware threads, four at most; 0 returns to that.
 * Returns the value in force before.  The results do not depend on it. */
int pep_set_host_threads(int n);

/* A raw DEFLATE stream (RFC 1951; what a zip member of method 8 holds) of `src` made of dynamic-Huffman blocks with literals only - entropy
 * coding without a match search, for the members of <prefix>.seq.npz (packed alleles: nothing to match).  Host C++, no context, any inflate
 * reads it.  Returns the stream's length; when it exceeds `cap` nothing usable was written (n + n / 64 + 512 always suffices).  Negative: PEP_ERR_ARG. */
int64_t pep_deflate_literals(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap);

/* A raw DEFLATE stream of `src` with matches found by ONE probe of a 4-byte hash per position (greedy; the matcher of the fastest levels of the
 * modern deflate libraries) in dynamic-Huffman blocks of 64 KiB of input - for the members of <prefix>.mat.npz (PEPPAN.py:959-966; the reference deflates
 * them at zlib's default level inside zipfile): about the size of zlib's level 1 at several times its rate.  Host C++, no context, any inflate reads
 * it.  Returns the stream's length; when it exceeds `cap` nothing usable was written (n + n / 8 + 1024 always suffices).  Negative: PEP_ERR_ARG. */
int64_t pep_deflate_fast(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap);

/* CRC-32 of the zip format (what zlib.crc32 returns) of src[0 .. n) continued from `crc` (0 to start): by carry-less multiplication where the CPU has the
 * instruction (an order of magnitude above zlib's table walk), zlib's otherwise.  Every member MapBsn appends carries one (zipfile computes it inside
 * ZipFile.writestr, PEPPAN.py:81-85 through numpy.savez). */
uint32_t pep_crc32(const uint8_t *src, int64_t n, uint32_t crc);

/* A store member ready for its archive in one call: the raw DEFLATE stream of `src` by coder 0 (pep_deflate_literals) or 1 (pep_deflate_fast) and, in *crc,
 * the CRC-32 of `src`.  Returns the stream's length; beyond `cap` nothing usable was written.  Negative: PEP_ERR_ARG. */
int64_t pep_pack_member(const uint8_t *src, int64_t n, int32_t coder, uint8_t *out, int64_t cap, uint32_t *crc);
/* np.argsort(v.astype(object)) for float64 v without NaN (host, no context): the order numpy's generic index quicksort leaves an object column of
 * Python floats in, ties included - the order of equal scores in the reference's .tab store (PEPPAN.py:957-960 sorts an object array's score column).
 * PEP_ERR_LIMIT when the sort's depth limit is reached (numpy switches to heapsort there: the caller asks numpy itself). */
int pep_argsort_object_order(const double *v, int64_t n, int64_t *order);

#ifdef __cplusplus
}
#endif
#endif
