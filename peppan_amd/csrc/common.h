// Internal declarations shared by the HIP translation units of libpeppan_hip.so.
// gfx950 (MI355X, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/peppan_hip.h"

#define PEP_WAVE 64
// copies of the 1 KiB substitution table in LDS (one per bank group): 32 = conflict-free gather, 16 = half the LDS for at most 2-way conflicts
#define PEP_TAB_REP 16

#define PEP_HIP(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            return pep_fail((ctx), PEP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
        }                                                                                             \
    } while (0)

#define PEP_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != PEP_OK) return _rc; \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// packed sequence set on the device.  Residues of sequence i live at res[off[i] .. off[i]+len[i]);
// at least PEP_SEQ_GAP bytes of PEP_PAD_CODE separate consecutive sequences and PEP_END_PAD pad both ends,
// so a spaced seed can never straddle two sequences.
#define PEP_PAD_CODE 31
#define PEP_SEQ_GAP 16
#define PEP_END_PAD 64
struct SeqSet {
    uint32_t n = 0;
    uint64_t total = 0;          // bytes in res (with padding)
    uint64_t residues = 0;       // sum of len
    uint32_t max_len = 0;
    DevBuf res;                  // u8
    DevBuf off;                  // u32[n+1]  (off[n] = total, sentinel for binary search)
    DevBuf len;                  // u32[n]
    DevBuf blk2seq;              // uint2[total/32]: (sequence whose residues a 32-byte block holds, start of that sequence) - starts are 16-aligned, gaps >= 16, so never two; one look-up
                                 // turns a packed position into (sequence, position inside it)
    // host mirrors
    std::vector<uint32_t> h_off, h_len;
};

// nucleotide sequence set (ASCII upper-cased on upload is NOT required: kernels fold case)
struct NtSet {
    uint32_t n = 0;
    uint64_t total = 0;
    DevBuf nt;                   // u8 ASCII, concatenated
    DevBuf off;                  // u64[n+1]
    std::vector<uint64_t> h_off;
};

// pinned host memory of a context: small read-backs (counts) and the staging area of downloaded tables.  A device -> host copy into
// pageable memory costs ~27 us per round trip on this box, into pinned memory ~16 us (tools/micro/sync_cost.hip), and large copies
// run at DMA speed only into pinned memory.
struct PinBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
};

struct pep_result;

// K1's descriptor of a packed sequence (translate.hip): the nucleotide sequence it was translated from, the frame (1 - 3 forward, 4 - 6 reverse strand), the residue its
// chunk starts at inside that frame, its length in residues.  One per query (d_k1_desc_q) / target (d_k1_desc_t), on the device; the host's meta records are made from them.
struct PackDesc {
    uint32_t seq;
    uint32_t frame;
    uint32_t aa_off;
    uint32_t len;
};
// ... and of a packed nucleotide sequence (pep_use_nt_as_residues: NuclSide::d_desc): the sequence and whether it is its reverse complement
struct NuclDesc { uint32_t seq, rev; };

struct pep_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    PinBuf pin_small;                       // 16 KiB: counters read back between kernels
    size_t pin_small_used = 0;
    struct PendingRead { void *dst; size_t off, n; } pending[32];
    int n_pending = 0;
    PinBuf pin_k1, pin_k1q;                 // grow-only: what K1's kernels write for the host - the set's summary (reference), summary + lengths (queries)
    DevBuf d_k1_desc_q, d_k1_desc_t;        // K1's descriptors (+ the summary's accumulators behind them): fetched only when the host tables are asked for
    DevBuf d_self_delta, d_self_t;          // seed stage, self-search: per 32-byte block of the target layout the distance to the query that IS that target (pep_self_map, self_prepare);
                                            // per reference sequence the packed sequence its frame 1 starts with (K1: k1_ref_desc)
    uint32_t self_first_n = 0;              // entries of d_self_t (reference sequences of the last K1 of the reference side)
    const uint32_t *self_first = nullptr;   // what pep_self_map found applicable to the current sets: d_self_t (K1's) or the nucleotide tool's nucl_t.d_first; its entries; may a target be longer than its query
    uint32_t self_first_cnt = 0;
    int self_exact_len = 0;
    hipEvent_t k1_event = nullptr;          // the point of the stream where the reference side's downloads have arrived
    hipEvent_t k1q_event = nullptr;         // ... and the query side's
    bool k1q_event_set = false;
    bool k1_t_deferred = false;             // pep_search: the reference side's K1 is queued, its summary not taken yet
    hipEvent_t wait_event = nullptr;        // pep_stream_wait: marks the point of the stream the host is waiting for
    hipEvent_t k1_t0 = nullptr, k1_t1 = nullptr;   // pep_translate's timing pair (created once)
    uint32_t k1_desc_cap = 0;               // descriptor slots of the reference side's last K1 (the summary sits behind them in pin_k1)
    bool t_tables_lazy = false, q_tables_lazy = false;   // a side's host tables (meta records, h_off, h_len) have not been built from its pinned descriptors yet (pep_k1_host_tables)
    PinBuf pin_up;                          // grow-only: staging area of uploads out of pageable memory (pep_h2d)
    size_t pin_up_used = 0;
    hipEvent_t up_event = nullptr;          // the point of the stream where the last upload out of pin_up has left it
    bool up_event_set = false;
    PinBuf pin_down;                        // grow-only: staging area of downloads into pageable memory (pep_d2h_queue / pep_d2h_finish)
    size_t pin_down_used = 0;
    struct PendingDown { void *dst; size_t off, n; } down[16];
    int n_down = 0;
    PinBuf pin_stage;                       // grow-only: the hit table of the newest search
    PinBuf pin_ms;                          // grow-only: per-query score thresholds on their way to the device
    pep_result *staged_result = nullptr;    // the result whose hits still live in pin_stage (materialised before it is overwritten)
    std::string err;
    pep_search_params params;
    // inputs
    NtSet q_nt, r_nt;
    SeqSet q, t;
    std::vector<pep_query_meta> q_meta;     // frame chosen per query (K1)
    std::vector<pep_target_meta> t_meta;    // (seq, frame, chunk offset, length) per target (K1)
    bool q_from_nt = false, t_from_nt = false;
    bool q_ready = false, t_ready = false, sub_ready = false, codon_ready = false;
    bool resid_from_nucl = false;           // ctx->q / ctx->t hold base codes made by pep_use_nt_as_residues (until the next pep_translate / pep_set_*)
    int q_gtable = 11, t_gtable = 11, t_frames = 6;
    // K1 reference side: chunk-slot prefix per (sequence, frame), a function of the input lengths only - kept between translations
    DevBuf d_k1_base;
    std::vector<uint64_t> k1_base;
    uint64_t k1_upper = 0;
    int k1_base_frames = 0;            // 0 = not computed for the current reference set
    // ... and the tables of its LONG frames (contigs of genomes: k1_stop_mask / k1_ref_chunks_mask / k1_ref_desc_fill): tiles of the stop mask, the long frames'
    // records (+ frame -> record), the stop mask itself, tiles of chunk slots
    DevBuf d_k1_seg, d_k1_long, d_k1_spec, d_k1_tiles;
    uint32_t k1_n_seg = 0, k1_n_long = 0, k1_n_tiles = 0;
    // what pep_use_nt_as_residues makes of the nucleotide sets apart from the residues themselves (translate.hip: pep_nucl_sets), kept per pair of uploads
    struct NuclSide {
        uint32_t n = 0, max_len = 0;
        uint64_t total = 0, residues = 0;
        std::vector<uint32_t> h_off, h_len;
        std::vector<pep_query_meta> q_meta;
        std::vector<pep_target_meta> t_meta;
        DevBuf d_off, d_len, d_desc;
        DevBuf d_first;                     // targets: per reference sequence the packed sequence that is its forward strand (seeds.hip: self_prepare)
        uint32_t n_first = 0;
    } nucl_q, nucl_t;
    bool nucl_valid = false;
    int nucl_strands = 0;
    DevBuf d_min_score;
    DevBuf d_trace_defer;                   // traceback pass: pairs that left their sub-band, waiting for their full-band sweep (sw.hip: run_deferred)
    DevBuf d_trace_mode;                    // per traced pair: first lane of the sub-band its traceback codes cover, -1 = the full band (sw.hip)
    std::vector<uint32_t> group_of_seq;     // optional: competition group of every reference sequence (pep_set_target_groups)
    DevBuf d_t_class;
    DevBuf d_t_subject;                         // hsp_mode 2: the reference sequence every target is a strand / frame / chunk of (uploaded per search)
    bool t_class_ready = false;
    // workspaces (grow-only, reused between searches)
    DevBuf ws[24];
    DevBuf sub_lds;                         // replicated substitution table image (32 KiB)
    DevBuf d_params;                        // device copy of seed params
    int8_t d_params_host[1024] = {};        // what d_params holds (uploaded again only when the substitution table changes)
    bool d_params_valid = false;
    // phase timers: events recorded on the stream, read once after the search's final synchronisation (waiting for an end event in
    // the middle of a search costs a host round trip with the GPU idle, and lets nothing be queued behind a running SW pass)
    hipEvent_t tm_a[12] = {}, tm_b[12] = {};
    int tm_state[12] = {};                   // 0 idle, 1 begun, 2 ended (waiting to be read)
    int timing_level = 0;                    // pep_set_timing: 0 no phase timers, 1 the score pass only, 2 all of them
    PinBuf pin_labels;                       // grow-only: K10's labels on their way to the caller
    uint64_t trace_swept = 0;               // pairs the last traceback pass swept (the rest were settled by the gapless shortcut)
    struct ScanState { DevBuf buf; uint32_t epoch = 0, ticket_base = 0; bool dirty = false; };     // dirty: a failure may have left host and device ticket counts apart - next use starts from a cleared area (pep_fail)
    ScanState fused_state[4];               // kernels that scan while they compute (lookback.h): select (count, run capacity), top-k (hits, CIGAR runs)
    ScanState scan_state[2];                // single-launch scans (u32, u64): ticket counter + one status word per tile (scan.hip)
    DevBuf sort_state, sort_hist;           // one-launch-per-pass radix sort (sort.hip): ticket + status words per (tile, digit); its own histograms
    uint32_t sort_epoch = 0, sort_ticket_base = 0;
    bool sort_dirty = false;
    DevBuf d_set;                           // candidate hash set of the seed stage (seeds.hip); set_compact leaves every slot it read EMPTY again
    uint64_t set_clean_slots = 0;           // leading slots of d_set known to be EMPTY (0 while a search is using it)
    DevBuf uf_nodes;                        // K10 over a device-resident hit table: node of every target (uploaded when it changes)
    std::vector<uint32_t> uf_nodes_host;
    struct { bool pending = false; pep_result *res = nullptr; const void *d_hits = nullptr, *d_cig = nullptr; const uint32_t *d_n_hits = nullptr; uint64_t n_bound = 0; bool parent_ready = false; } ext;
                                            // a search whose result left through pack_out and whose host half (sizes, statistics, views) is still to be done (pep_extend_finish)
    struct { void *d_dst = nullptr; const void *pinned_src = nullptr; uint64_t n_words = 0; } upload;   // an upload out of pinned memory that rides on the next read-back kernel
    bool want_nt_match = false;              // pep_set_nt_match: the searches of this context end with K7's count of identical nucleotide columns per hit (rescore.hip: pep_k7_hits_queue)
    DevBuf d_nt_match;
    PinBuf pin_nt_match;                     // grow-only: those counts on their way to the result
    uint32_t grp_nodes = 0, grp_q_base = 0;  // pep_set_grouping: the searches of this context end with K10 over their own hit table (0 = off)
    bool device_results = false;            // pep_set_result_mode: searches leave their table on the device; the host copy is fetched on demand
    pep_result *dev_result = nullptr;       // the result whose hit table is still intact on the device (ws[23]): the newest search's, until the workspace is reused
    DevBuf d_zero;                          // the small counters of one search, cleared by ONE fill when it starts (layout: PEP_ZERO_* below)
    bool zero_clean = false;                // the seed stage's part of d_zero is zero as of the end of what is queued (pack_out cleared all of d_zero): with every
                                            // zero_ok flag still set the next search needs no fill
    DevBuf d_mail_copy;                     // the alignment stage's counter block outside d_zero (trace.hip: emit -> pack_out, K10)
    bool zero_ok[4] = {false, false, false, false};     // which consumer regions of d_zero are still untouched since that fill (PEP_ZC_*)
    // stats of the last search
    pep_stats stats;
};

struct pep_result {
    pep_ctx *ctx = nullptr;
    std::vector<pep_hit> hits;              // used once the result no longer lives in the context's staging area
    std::vector<uint32_t> cigar;
    const pep_hit *st_hits = nullptr;       // while staged: views into ctx->pin_stage
    const uint32_t *st_cigar = nullptr;
    uint64_t n_hits = 0, n_cigar = 0;
    const pep_hit *d_hits = nullptr;        // device copy (context workspace), valid while ctx->dev_result == this
    const uint32_t *d_cigar = nullptr;
    pep_stats stats;
    std::vector<uint32_t> labels;           // pep_set_grouping: the partition of the search's hit graph (one label per node)
    std::vector<uint32_t> nt_match;         // pep_set_nt_match: identical nucleotide columns per hit (K7's n_match)
};

// HIP-event stopwatch on one stream (the kernel times bench.py reports are taken with it, inside the library,
// on the stream the kernels are launched on); destroys its events on every exit path
hipError_t pep_event_wait(hipEvent_t ev);      // polls before it sleeps (capi.hip)
unsigned pep_wait_event_flags();               // flags of an event that is only ever waited for (blocking when PEPPAN_HIP_SPIN_US=0)

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    explicit EventTimer(hipStream_t s) : st(s)
    {
        if (hipEventCreate(&a) != hipSuccess) a = nullptr;
        if (hipEventCreate(&b) != hipSuccess) b = nullptr;
        if (a) (void)hipEventRecord(a, st);
    }
    float stop()                      // records the end event, waits for it, returns milliseconds (0 on failure)
    {
        float ms = 0.f;
        if (a && b && hipEventRecord(b, st) == hipSuccess && pep_event_wait(b) == hipSuccess) (void)hipEventElapsedTime(&ms, a, b);
        return ms;
    }
    ~EventTimer()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    EventTimer(const EventTimer &) = delete;
    EventTimer &operator=(const EventTimer &) = delete;
};

enum PepTimer { TM_SEED = 0, TM_TOTAL, TM_SW, TM_SW_TRACE, TM_TRACE, TM_MATCH0, TM_MATCH1, TM_MATCH2, TM_MATCH3, TM_COUNT };
void pep_timer_begin(pep_ctx *ctx, int id);
void pep_timer_end(pep_ctx *ctx, int id);
void pep_timers_resolve(pep_ctx *ctx);     // after a stream synchronisation: elapsed times -> ctx->stats.ms_*

int pep_fail(pep_ctx *ctx, int code, const std::string &msg);
// small device -> host reads: queue any number with pep_read_back (async copy into the pinned page), then ONE pep_sync_reads
// waits for the stream and stores the values
int pep_read_back(pep_ctx *ctx, void *dst, const void *d_src, size_t n);
int pep_sync_reads(pep_ctx *ctx);
// Waits until everything queued on the context's stream so far has finished - by polling an event for up to a few hundred microseconds
// before falling back to a blocking wait: the searches are chains of short kernels with a handful of host decisions in between, and a
// sleeping host thread adds tens of microseconds of wake-up latency to each of them.
hipError_t pep_stream_wait(pep_ctx *ctx);
hipError_t pep_event_wait(hipEvent_t ev);
int pin_reserve(pep_ctx *ctx, PinBuf &b, size_t bytes);
// stream-ordered upload of caller memory: through the context's pinned staging area for anything of size (a copy command straight out of pageable
// memory has the runtime pin the pages first - 15 ms for the 10 MB of a gene set it has not seen before, 0.2 ms the second time); the caller's buffer
// is free on return
int pep_h2d(pep_ctx *ctx, void *d_dst, const void *h_src, size_t n);
// the way back: pep_d2h_queue queues a download into the pinned staging area (small ones go straight to their destination), pep_d2h_finish - called once
// the stream has been waited for - copies what arrived to where the caller wants it.  A copy command straight into pageable memory makes the driver
// register those pages; when the caller frees them (numpy arrays of a megabyte or more go back to the system at once) the driver evicts the process's
// queues to drop the registration - the NEXT call on the GPU then takes 10 - 30 ms (found in round 5 behind the last gather of RunBlast.run)
int pep_d2h_queue(pep_ctx *ctx, void *h_dst, const void *d_src, size_t n);
void pep_d2h_finish(pep_ctx *ctx);
void pep_materialise_staged(pep_ctx *ctx);
void pep_drop_dev_result(pep_ctx *ctx);
int dev_reserve(pep_ctx *ctx, DevBuf &b, size_t bytes);
void dev_release(DevBuf &b);

// Layout of pep_ctx::d_zero: every small counter block a search needs starts from zero, and one fill at the start of the search clears
// them all (each used to be a fill of its own in front of its kernel: eleven tiny launches per search).  A stage that runs without the
// seed stage in front (K9 drives the alignment stage alone) finds its flag in zero_ok unset and clears its own block.
#define PEP_ZERO_SEED 0                                         // 64 B: list length, overflow flags, statistics (seeds.hip)
#define PEP_SORT_TOP_BITS 10                                    // two-level candidate sort (sort.hip): buckets by the top 10 bits of the key ...
#define PEP_SORT_TOP_CAP 4096                                   // ... each sorted in LDS when none holds more keys than this
#define PEP_ZERO_TOP (PEP_ZERO_SEED + 64)                       // 1024 x 4 B: candidates per top digit of the dense key (set_compact; read back WITH the 64 bytes in front)
#define PEP_ZERO_SHAPE (PEP_ZERO_TOP + 4096)                    // 4 x 16 B: raw-hit and run counters per seed shape
#define PEP_ZERO_COARSE (PEP_ZERO_SHAPE + 64)                   // 4 x 8192 x 4 B: coarse-bucket counts of the query index per seed shape
#define PEP_ZERO_SORT (PEP_ZERO_COARSE + 4 * 8192 * 4)          // 8 x 2048 x 4 B: digit histograms of the candidate sort
#define PEP_ZERO_SW_BYTES (64 + 2 * 1024 * 4)                   // totals + length histogram + scatter cursors of one Smith-Waterman pass (sw.hip)
#define PEP_ZERO_SW_SCORE (PEP_ZERO_SORT + 8 * 2048 * 4)
#define PEP_ZERO_SW_TRACE (PEP_ZERO_SW_SCORE + PEP_ZERO_SW_BYTES)
#define PEP_ZERO_SELECT (PEP_ZERO_SW_TRACE + PEP_ZERO_SW_BYTES)  // 256 B: counters of the selection stage (trace.hip)
#define PEP_ZERO_TILES (PEP_ZERO_SELECT + 256)                  // 4 x 8 x 128 B: the matchers' tile claims per seed shape: eight counters, a cache line each (seeds.hip: TileClaims)
#define PEP_ZERO_TOTAL (PEP_ZERO_TILES + 4 * 8 * 128)
enum { PEP_ZC_SW_SCORE = 0, PEP_ZC_SW_TRACE, PEP_ZC_SELECT, PEP_ZC_SORT };
// the block of consumer `which` (PEP_ZC_*), zeroed: taken from the search's one fill when it is still untouched, cleared here otherwise
int pep_zero_block(pep_ctx *ctx, int which, size_t offset, size_t bytes, void **out);

// ---- scan.hip
// exclusive; d_out[n] = total (n+1 outputs); d_total (optional): a second place the total is written to (a counter block the host reads in one copy)
int pep_scan_u32(pep_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n, DevBuf &tmp, uint32_t *d_total = nullptr);
int pep_scan_u64(pep_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t n, DevBuf &tmp, uint64_t *d_total = nullptr);
int pep_copy_from_pinned(pep_ctx *ctx, void *d_dst, const void *pinned_src, uint64_t n_words);     // a kernel instead of a copy command (scan.hip)
int pep_exchange_pinned(pep_ctx *ctx, void *d_up_dst, const void *pinned_up_src, uint64_t n_up_words, void *pinned_down_dst, const void *d_down_src, uint64_t n_down_words);
// pep_read_back through a kernel that also carries an upload out of pinned memory (one launch for both directions; n a multiple of 4)
int pep_read_back_with_upload(pep_ctx *ctx, void *dst, const void *d_src, size_t n, void *d_up_dst, const void *pinned_up_src, uint64_t n_up_words);
// ---- sort.hip
// the dense candidate-key form q | t | bin - bin_min (tb / bb bits for t / bin) <-> q:21 | t:25 | bin:18 (seeds.hip); on = 0: keys pass unchanged
struct pep_key_unpack { int on, tb, bb; uint32_t bin_min; };
int pep_sort_u64(pep_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, const uint32_t *d_n, uint64_t n_bound, int bits, uint32_t *d_hist_zeroed,
                 const pep_key_unpack *unpack);   // result in d_keys
int pep_sort_u64_two_level(pep_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint64_t n, int bits, const uint32_t *d_top_hist, const pep_key_unpack *unpack);
// ---- translate.hip  (K1)
int pep_k1_query(pep_ctx *ctx, int gtable, int phase = 0);
int pep_k1_ref(pep_ctx *ctx, int frames, int gtable, int phase = 0);
int pep_k1_host_tables(pep_ctx *ctx);      // builds the per-sequence host tables of both sides if their last K1 left them to be built on demand
int pep_k1_host_tables_q(pep_ctx *ctx);    // the query side only (the search needs the query lengths for its score thresholds)
int pep_nucl_sets(pep_ctx *ctx, int strands);      // the nucleotide sets themselves as residue sets (base codes; reference: forward strands + reverse complements)
// ---- seeds.hip  (K2-K4)
// before_sync (optional): host work to do once every kernel of the stage is queued, while the GPU runs them (the stage ends with a synchronisation)
int pep_find_candidates(pep_ctx *ctx, uint64_t **d_cands, uint64_t *n_cands, int (*before_sync)(pep_ctx *) = nullptr, int (*need_targets)(pep_ctx *) = nullptr);
int pep_upload_sub_table(pep_ctx *ctx);       // ctx->params.sub -> ctx->d_params (1 KiB, uploaded when it changed)
// ---- sw.hip / trace.hip (K5, K6, K8)
int pep_sw_run(pep_ctx *ctx, const uint64_t *d_cands, uint64_t n, bool trace, const int32_t *d_known = nullptr, const int32_t *d_end_lane = nullptr, const int32_t *d_skip_mode = nullptr,
               const uint32_t *d_n = nullptr, uint64_t dir_blocks_bound = 0, unsigned long long **d_hdr = nullptr);   // kernel time: phase timers TM_SW / TM_SW_TRACE
// h_min_score: score threshold per query; nullptr = ctx->d_min_score holds them already (pep_search uploads them while the seed stage runs)
// defer: the caller waits for the stream itself and calls pep_extend_finish afterwards (pep_search: one wait for everything)
int pep_extend(pep_ctx *ctx, const uint64_t *d_cands, uint64_t n_cands, const int32_t *h_min_score, pep_result *res, bool defer = false);
int pep_extend_finish(pep_ctx *ctx);
int pep_selftest_dpp(pep_ctx *ctx);
// ---- rescore.hip (K7)
int pep_k7_rescore(pep_ctx *ctx, uint64_t n, const pep_nt_hit *h_hits, const uint32_t *h_cigar, uint64_t n_cigar, int64_t *h_out);
// ---- unionfind.hip (K10)
int pep_k10_components(pep_ctx *ctx, uint32_t n_nodes, uint64_t n_edges, const uint32_t *h_a, const uint32_t *h_b, uint32_t *h_label);
int pep_k10_components_dev(pep_ctx *ctx, uint32_t n_nodes, uint64_t n_hits, const pep_hit *d_hits, uint32_t q_base, const uint32_t *h_node_of_target,
                           uint64_t n_targets, uint32_t *h_label);

int pep_k7_hits_queue(pep_ctx *ctx, uint64_t n_hits, const pep_hit *d_hits, const uint32_t *d_cigar, const uint32_t *d_n_hits = nullptr);   // K7's match counts of a search's own hits -> ctx->pin_nt_match (no wait; rescore.hip)
int pep_k10_queue(pep_ctx *ctx, uint64_t n_hits, const pep_hit *d_hits, const uint32_t *d_n_hits = nullptr);   // d_n_hits: the count lives on the device, n_hits bounds it       // K10 behind a search, labels -> ctx->pin_labels (no wait)
int pep_k10_set_grouping(pep_ctx *ctx, uint32_t n_nodes, uint32_t q_base, const uint32_t *h_node_of_target, uint64_t n_targets);

// ---- overlaps.hip (K11)
int pep_k11_overlaps(pep_ctx *ctx, uint64_t n, const int32_t *h_contig, const int64_t *h_start, const int64_t *h_end, const int64_t *h_rid,
                     double ovl_l, double ovl_p, int64_t *h_out, uint64_t cap, uint64_t *n_pairs);
// ---- alleles.hip (K12)
int pep_k12_alleles(pep_ctx *ctx, const uint8_t *h_nt, const uint64_t *h_nt_off, uint32_t n_contigs, uint64_t n, const pep_locus *h_rows,
                    const uint32_t *h_cigar, uint64_t n_cigar, uint32_t n_groups, const uint64_t *h_grp_off, const uint32_t *h_grp_qlen,
                    int gtable, int64_t *h_in_frame, int64_t *h_orf, uint8_t *h_packed, uint64_t packed_cap);
// ---- dedup.hip (K13)
int pep_k13_sha1(pep_ctx *ctx, const uint8_t *h_bytes, const uint64_t *h_off, uint32_t n, uint8_t *h_digest);
int pep_k13_dedup(pep_ctx *ctx, uint32_t n, const uint32_t *h_len, const uint8_t *h_digest, uint32_t *h_rep);
// ---- similar.hip (K14)
int pep_k14_pair_support(pep_ctx *ctx, uint64_t n_rows, const pep_support_row *h_rows, const uint32_t *h_cigar, uint64_t n_cigar, uint64_t n_groups,
                         const uint64_t *h_grp_off, const uint32_t *h_qlen, const uint32_t *h_rlen, const pep_support_limits *lim, int32_t *h_value);
// ---- linclust.hip (K9)
int pep_k9_linclust(pep_ctx *ctx, const uint8_t *h_res, const uint64_t *h_off, uint32_t n, int base, int k, int m, double min_id, double min_cov,
                    uint32_t *h_rep, uint64_t *h_stats);

static inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
int pep_upload_blk2seq(pep_ctx *ctx, SeqSet &s);
// translate.hip, for the seed stage of a self-search (seeds.hip: self_prepare): K1 leaves ctx->d_self_t[g] = the packed sequence that frame 1 of reference sequence g
// starts with (PEP_SELF_NONE: none); pep_self_map clears ctx->d_self_delta - one word per 32-byte block of the target layout, PEP_SELF_NO_DELTA - and says whether
// the current sets came out of K1 at all (*on = 0: not applicable, nothing queued).
#define PEP_SELF_NONE 0xFFFFFFFFu
#define PEP_SELF_NO_DELTA ((int32_t)0x80808080)
int pep_self_map(pep_ctx *ctx, int *on);
int pep_upload_codes(pep_ctx *ctx, SeqSet &s, const uint8_t *codes, const uint64_t *off, uint32_t n, uint32_t max_n);
int pep_upload_sub(pep_ctx *ctx);
