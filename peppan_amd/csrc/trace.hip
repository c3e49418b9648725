// K6 + K8: choose the best band per (query, target), walk the traceback codes written by K5 (one alignment per lane),
// run-length encode the CIGAR, count identities, apply the identity / query-cover filters and the
// per-(query, split) top-k, and emit fixed-size hit records + a CIGAR arena ordered by (q, t).
#include "common.h"
#include "lookback.h"
#include <cstring>

namespace {

struct SelInfo {          // one per selected (q,t) pair
    uint32_t cand;        // index into the compacted key list of selected pairs (== own index)
    int32_t score, iend, jend;
    int32_t istart, jstart;
    uint32_t n_runs, aln_len, n_ident;
    uint32_t pass;        // survived the filters
    uint32_t keep;        // survived top-k
    uint32_t pad;
};

__device__ __forceinline__ uint32_t key_q(uint64_t k) { return (uint32_t)(k >> 43); }
__device__ __forceinline__ uint32_t key_t(uint64_t k) { return (uint32_t)((k >> 18) & ((1u << 25) - 1)); }
__device__ __forceinline__ int key_dlo(uint64_t k) { return (int)(k & ((1u << 18) - 1)) * 64 - (1 << 23) - 32; }

// Selection (group heads pick the best band of their (q, t) group: score desc, then lowest bin = first in sorted order; hsp_mode 1: every band
// that reaches the threshold), compaction and run slots in ONE launch (lookback.h): every thread judges its candidate, the tile scans two
// values at once - selected pairs (their slot in the arrays of the selected) and their run capacity (the slot of their CIGAR runs) - finds
// the totals of the tiles in front of it by look-back and writes the selected pairs' records straight to their slots.  Four launches of a
// dependent chain before (about 5 us each for 50 k candidates).
struct SelectState { uint64_t *count_state, *run_state; uint32_t ticket_base; uint64_t epoch_count, epoch_run; };
__global__ __launch_bounds__(256) void select_gather(const uint64_t *__restrict__ cands, uint64_t n, const int4 *__restrict__ sw, const int32_t *__restrict__ min_score,
                                                     int hsp_mode, const uint32_t *__restrict__ q_len, const uint32_t *__restrict__ t_len, SelInfo *__restrict__ sel,
                                                     uint64_t *__restrict__ run_off, uint64_t *__restrict__ sel_keys, int32_t *__restrict__ known, int32_t *__restrict__ end_lane,
                                                     uint32_t *__restrict__ counters /* [0] (q, t) groups, [1] selected pairs */, unsigned long long *__restrict__ total_runs,
                                                     SelectState st)
{
    __shared__ uint32_t s_tile, lds32[4];
    __shared__ uint64_t lds64[4], s_pre[2];
    const uint32_t tile = lb_take_tile(st.count_state, st.ticket_base, &s_tile);
    const uint64_t c = (uint64_t)tile * 256 + threadIdx.x;
    const bool live = c < n;
    const uint64_t g = live ? cands[c] >> 18 : 0;
    const bool head = live && (c == 0 || (cands[c - 1] >> 18) != g);
    const int heads = __syncthreads_count(head);                 // (q, t) groups that start in this tile: one atomic per block
    if (threadIdx.x == 0 && heads) atomicAdd(&counters[0], (uint32_t)heads);
    uint32_t f = 0, b = (uint32_t)c;
    if (live && hsp_mode != 0) {
        // every band that reaches the threshold is traced; duplicates are removed after the walk (dedupe_bands; hsp_mode 2: cull_hsps)
        if (sw[c].x > 0 && sw[c].x >= min_score[key_q(cands[c])]) f = 1;
    } else if (head) {
        int best = sw[c].x;
        for (uint64_t x = c + 1; x < n && (cands[x] >> 18) == g; ++x)
            if (sw[x].x > best) { best = sw[x].x; b = (uint32_t)x; }
        if (best > 0 && best >= min_score[key_q(cands[c])]) f = 1;
    }
    uint64_t key = 0, cap = 0;
    int4 r = make_int4(0, 0, 0, 0);
    if (f) {
        key = cands[b]; r = sw[b];
        cap = 2ull * min(q_len[key_q(key)], t_len[key_t(key)]) + 2;      // an alignment has at most 2 * min(Lq, Lt) + 1 runs (M runs consume a residue of both sequences)
    }
    uint32_t tot_f;
    uint64_t tot_cap;
    const uint32_t ex_f = block_excl_scan_256<uint32_t>(f, &tot_f, lds32);
    const uint64_t ex_cap = block_excl_scan_256<uint64_t>(cap, &tot_cap, lds64);
    if (threadIdx.x < 64) {
        const uint64_t p0 = lb_tile_prefix<32>(st.count_state + 1, tile, tot_f, st.epoch_count, (int)threadIdx.x);
        const uint64_t p1 = lb_tile_prefix<48>(st.run_state + 1, tile, tot_cap, st.epoch_run, (int)threadIdx.x);
        if (threadIdx.x == 0) { s_pre[0] = p0; s_pre[1] = p1; }
    }
    __syncthreads();
    if (f) {
        const uint64_t slot = s_pre[0] + ex_f;
        SelInfo s;
        s.cand = (uint32_t)slot; s.score = r.x; s.iend = s.jend = -1;      // the end cell comes from the traceback pass
        s.istart = s.jstart = 0; s.n_runs = s.aln_len = s.n_ident = 0; s.pass = s.keep = 0; s.pad = 0;
        sel[slot] = s;
        sel_keys[slot] = key;
        known[slot] = r.x;              // the traceback pass looks for the first cell that reaches this score ...
        end_lane[slot] = r.y;           // ... in the sub-band around the lane where the score pass met it
        run_off[slot] = s_pre[1] + ex_cap;
    }
    if ((uint64_t)(tile + 1) * 256 >= n && threadIdx.x == 255) {      // the last tile: totals
        counters[1] = (uint32_t)(s_pre[0] + tot_f);
        *total_runs = s_pre[1] + tot_cap;
    }
}

// Rule 5a (oracle: gapless_segment / band_align): before any traceback sweep, the two diagonals of the lowest lane
// that met the band's score T in the score pass are scanned for an UNGAPPED segment scoring T.  Such a segment IS the alignment that ends
// in its last cell (H equals the running ungapped sum all along it - anything larger would end above T - and the diagonal move has
// priority), so the pair is settled as one M run: mode -2, no codes, no walk.
// One wavefront per pair; a diagonal is taken 512 cells at a time, eight consecutive cells per lane (two unaligned 8-byte loads - the
// packed sets carry >= 16 bytes of padding behind every sequence - and eight independent table reads).  With P = prefix sum of the
// scores and M = prefix minimum of P (the 0 in front included), run = P - M is Kadane's running sum with its restart after every cell
// that brings it to <= 0: lane sums and lane minima are scanned across the wavefront with DPP moves (no LDS round trips: a first
// version built on __shfl_up spent 165 us in dependent ds_bpermute chains), then every lane replays its eight cells.  The first cell
// with run == T ends the segment, the last restart before it starts it.
template <class Op>
__device__ __forceinline__ int wave_scan_incl(int v, const int identity, Op op)
{
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));      // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));      // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));      // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));      // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));      // row_bcast:15 -> rows 1, 3
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));      // row_bcast:31 -> rows 2, 3
    return v;
}

__global__ __launch_bounds__(256) void gapless_check(const uint32_t *__restrict__ d_n_sel, const uint64_t *__restrict__ cands, const int32_t *__restrict__ known,
                                                     const int32_t *__restrict__ end_lane, const uint8_t *__restrict__ q_res, const uint32_t *__restrict__ q_off,
                                                     const uint32_t *__restrict__ q_len, const uint8_t *__restrict__ t_res, const uint32_t *__restrict__ t_off,
                                                     const uint32_t *__restrict__ t_len, const int8_t *__restrict__ sub_g, int4 *__restrict__ out,
                                                     int32_t *__restrict__ mode)
{
    __shared__ int8_t sub[1024];
    reinterpret_cast<uint32_t *>(sub)[threadIdx.x] = reinterpret_cast<const uint32_t *>(sub_g)[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= *d_n_sel) return;                                  // (whole wavefronts leave together; the grid is sized from an upper bound)
    const uint64_t key = cands[s];
    const uint32_t q = key_q(key), t = key_t(key);
    const int T = known[s];
    const int Lq = (int)q_len[q], Lt = (int)t_len[t];
    const uint8_t *qs = q_res + q_off[q], *ts = t_res + t_off[t];
    constexpr int INF = 0x3fffffff;
    auto add = [](int a, int b) { return a + b; };
    auto mn = [](int a, int b) { return min(a, b); };
    int found_is = -1, found_ie = -1, found_d = 0;
    for (int x = 0; x < 2 && found_is < 0 && T > 0; ++x) {
        const int d = key_dlo(key) + 2 * end_lane[s] + x;
        const int i0 = max(0, -d), n = min(Lq - 1, Lt - 1 - d) - i0 + 1;        // cell k of the diagonal is (i0 + k, i0 + k + d)
        int carry_p = 0, carry_m = 0, last_reset = -1;
        for (int base = 0; base < n; base += 512) {             // wave-uniform trip count
            const int k0 = base + lane * 8;
            uint32_t qw[2] = {0u, 0u}, tw[2] = {0u, 0u};
            if (k0 < n) { __builtin_memcpy(qw, qs + i0 + k0, 8); __builtin_memcpy(tw, ts + i0 + k0 + d, 8); }
            int sc[8], S = 0, lmin = INF;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int qc = (qw[u >> 2] >> ((u & 3) * 8)) & 31, tc = (tw[u >> 2] >> ((u & 3) * 8)) & 31;
                sc[u] = k0 + u < n ? (int)sub[qc * 32 + tc] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { S += sc[u]; lmin = min(lmin, S); }
            const int p_start = carry_p + wave_scan_incl(S, 0, add) - S;
            const int m_incl = wave_scan_incl(p_start + lmin, INF, mn);
            const int m_start = min(carry_m, __builtin_amdgcn_update_dpp(INF, m_incl, 0x138, 0xf, 0xf, false));      // lane l <- lane l - 1
            int P = p_start, M = m_start, first = -1, reset = -1;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                P += sc[u]; M = min(M, P);
                const bool live = k0 + u < n && first < 0;
                if (live && P - M == T) first = u;
                if (live && P == M) reset = u;
            }
            const unsigned long long hit = __ballot(first >= 0);
            const unsigned long long resets = __ballot(reset >= 0);
            if (hit) {
                const int l = __builtin_ctzll(hit);
                const int f = __builtin_amdgcn_readlane(first, l), r_own = __builtin_amdgcn_readlane(reset, l);
                int start_k = last_reset + 1;
                if (r_own >= 0) start_k = base + l * 8 + r_own + 1;
                else {
                    const unsigned long long before = resets & ((1ull << l) - 1ull);
                    if (before) { const int r = 63 - __builtin_clzll(before); start_k = base + r * 8 + __builtin_amdgcn_readlane(reset, r) + 1; }
                }
                found_is = i0 + start_k; found_ie = i0 + base + l * 8 + f; found_d = d;
                break;
            }
            if (resets) { const int r = 63 - __builtin_clzll(resets); last_reset = base + r * 8 + __builtin_amdgcn_readlane(reset, r); }
            carry_p = __builtin_amdgcn_readlane(P, 63);
            carry_m = __builtin_amdgcn_readlane(M, 63);
        }
    }
    if (lane == 0) {
        if (found_is >= 0) {
            out[s] = make_int4(T, found_ie, found_ie + found_d, found_ie - found_is + 1);       // score, end cell, run length
            mode[s] = -2;
        } else mode[s] = -1;
    }
}

// One THREAD per selected pair.  The walk is a serial chain of tiny decisions; spread over a wavefront (one alignment per wave, as the
// first version did) every decision costs an issue slot of the whole CU - vector or scalar unit alike - and 40 k alignments took
// 0.48 ms.  One alignment per lane instead; with all 64 lanes of a wave in use there are too few waves to hide the load latency
// of the chain (0.15 ms), so a wave carries WALK_LANES alignments and the chip runs n / WALK_LANES waves.  A lane reads the code word of 8 consecutive
// cells of its current diagonal (4 bytes) and fetches WALK_AHEAD words down that diagonal at once, so the latency of the scattered
// 4-byte loads is paid once per 8 * WALK_AHEAD cells of a diagonal run, not once per word.  (Keeping single loads in flight across
// iterations does not work: rotating the word registers reads them, which makes the compiler wait for every outstanding load.)
#define WALK_AHEAD 8
#ifndef WALK_LANES               // (16 and 32 measured at the end of round 3 with -DWALK_LANES: no difference, tools/ab/variants.sh)
#define WALK_LANES 64         // alignments per wavefront: few, so that many wavefronts overlap their load latencies (see above)
#endif
__global__ __launch_bounds__(WALK_LANES) void walk(const uint32_t *__restrict__ d_n_sel, SelInfo *__restrict__ sel, const uint64_t *__restrict__ cands, const int4 *__restrict__ sw,
                                           const uint64_t *__restrict__ dir_off, const uint32_t *__restrict__ dirs, const int32_t *__restrict__ mode,
                                           const uint64_t *__restrict__ run_off, uint32_t *__restrict__ runs)
{
    const uint64_t s = (uint64_t)blockIdx.x * WALK_LANES + threadIdx.x;
    if (s >= *d_n_sel) return;
    SelInfo info = sel[s];
    const uint64_t key = cands[info.cand];
    // the codes cover the full band (64 lanes, mode -1) or the sub-band that starts at lane mode[] (32 lanes): the row of one 16-step
    // block holds 2 dwords per lane
    const int md = mode[info.cand];
    if (md == -2) {
        // settled by the gapless shortcut (gapless_check): sw[] holds score, end cell and the length of the one M run
        const int4 g = sw[info.cand];
        info.iend = g.y; info.jend = g.z; info.istart = g.y - g.w + 1; info.jstart = g.z - g.w + 1;
        info.n_runs = 1; info.aln_len = (uint32_t)g.w;
        runs[run_off[s]] = ((uint32_t)g.w << 2);
        sel[s] = info;
        return;
    }
    const int dlo = key_dlo(key) + (md > 0 ? 2 * md : 0);
    const size_t row = md < 0 ? 128 : 64;
    const int4 cell = sw[info.cand];
    const int a0 = cell.w;
    // word of (block b, diagonal rel): dword ((b * lanes + (rel >> 1)) * 2 + (rel & 1)) of this pair's traceback area
    const uint32_t *dir = dirs + dir_off[info.cand] * 128;
    uint32_t *out = runs + run_off[s];

    info.iend = cell.y; info.jend = cell.z;     // (the trace pass recomputes the same score)
    int i = info.iend, j = info.jend, state = 0;
    int istart = i, jstart = j;
    uint32_t n_runs = 0, aln_len = 0, cur_op = 3, cur_len = 0;
    int cur_rel = -1, cur_blk = -1, have = 0;
    uint32_t w[WALK_AHEAD];                                    // words of blocks cur_blk, -1, ... of diagonal cur_rel
#pragma unroll
    for (int k = 0; k < WALK_AHEAD; ++k) w[k] = 0;
    for (;;) {
        // cell (i, j) is the (m & 7)-th cell of block m >> 3 of diagonal rel
        const int rel = j - i - dlo;
        const int m = i - a0 + (rel >> 1);
        const int blk = m >> 3;
        if (rel != cur_rel || blk != cur_blk) {
            const uint32_t *col = dir + rel;
            if (rel == cur_rel && blk == cur_blk - 1 && have > 1) {
                // the next word down the same diagonal is already here (no load in flight: a refill waits for all of its loads)
#pragma unroll
                for (int k = 0; k + 1 < WALK_AHEAD; ++k) w[k] = w[k + 1];
                --have;
            } else {
#pragma unroll
                for (int k = 0; k < WALK_AHEAD; ++k) w[k] = blk >= k ? col[(size_t)(blk - k) * row] : 0u;
                have = WALK_AHEAD;
            }
            cur_rel = rel; cur_blk = blk;
        }
        const uint32_t word = w[0];
        const int nidx = m & 7;
        const uint32_t nib = (word >> (nidx * 4)) & 15u;
        uint32_t op, cnt = 1;
        if (state == 0) {
            const uint32_t src = nib & 3u;
            if (src == 0) break;
            if (src != 1) { state = (src == 2) ? 1 : 2; continue; }
            // follow the diagonal inside this word: up to nidx + 1 cells, never past row 0 / column 0
            // (a zero nibble of x = a diagonal code; the cells of the word below this one are shifted to the top and counted at once)
            const int lim = min(i, j);
            const uint32_t x = ((word & 0x33333333u) ^ 0x11111111u) << ((7 - nidx) * 4);
            const int run = x ? (__clz(x) >> 2) : 8;
            int r = min(min(run, nidx + 1), lim + 1);
            // whole words of diagonal codes further down the same diagonal, as far as they are already here: a long ungapped stretch
            // costs a handful of instructions per 8 cells instead of a trip through the general step
            if (r == nidx + 1)
                while (have > 1 && cur_blk > 0 && r + 8 <= lim + 1 && ((w[1] & 0x33333333u) ^ 0x11111111u) == 0u) {
                    r += 8;
#pragma unroll
                    for (int k = 0; k + 1 < WALK_AHEAD; ++k) w[k] = w[k + 1];
                    --have; --cur_blk;
                }
            op = 0; cnt = (uint32_t)r;
            istart = i - r + 1; jstart = j - r + 1;
        } else if (state == 1) {
            op = 2; state = (nib & 4u) ? 1 : 0;
        } else {
            op = 1; state = (nib & 8u) ? 2 : 0;
        }
        aln_len += cnt;
        if (op == cur_op) cur_len += cnt;
        else {
            if (cur_len) out[n_runs++] = (cur_len << 2) | cur_op;
            cur_op = op; cur_len = cnt;
        }
        if (op == 0) {
            if ((int)cnt > min(i, j)) break;            // the last cell of the run lies on row 0 or column 0
            i -= (int)cnt; j -= (int)cnt;
        } else if (op == 2) --j;
        else --i;
    }
    if (cur_len) out[n_runs++] = (cur_len << 2) | cur_op;
    info.istart = istart; info.jstart = jstart; info.n_runs = n_runs; info.aln_len = aln_len;
    sel[s] = info;
}

// identities + filters; FIN_LANES lanes per selected pair, parallel over the columns of each M run (a whole wavefront per pair meant
// 38 k wavefronts that each lived for a chain of dependent loads: 0.039 ms; two pairs per wavefront: 0.031 ms, four: 0.030 ms but
// slower once alignments are long)
constexpr int FIN_LANES = 32;
__global__ __launch_bounds__(256) void finalize(const uint32_t *__restrict__ d_n_sel, SelInfo *__restrict__ sel, const uint64_t *__restrict__ cands,
                                                const uint8_t *__restrict__ q_res, const uint32_t *__restrict__ q_off, const uint32_t *__restrict__ q_len,
                                                const uint8_t *__restrict__ t_res, const uint32_t *__restrict__ t_off,
                                                const uint64_t *__restrict__ run_off, const uint32_t *__restrict__ runs, double min_id_pct, double min_qcov_pct)
{
    const int lane = threadIdx.x & (FIN_LANES - 1);
    const uint64_t s = (uint64_t)blockIdx.x * (256 / FIN_LANES) + threadIdx.x / FIN_LANES;
    if (s >= *d_n_sel) return;                                // (whole lane groups leave together)
    const SelInfo info = sel[s];
    const uint64_t key = cands[info.cand];
    const uint32_t q = key_q(key), t = key_t(key);
    const uint8_t *qs = q_res + q_off[q], *ts = t_res + t_off[t];
    const uint32_t *rv = runs + run_off[s];
    int i = info.istart, j = info.jstart;
    uint32_t ident = 0;
    for (uint32_t r = 0; r < info.n_runs; ++r) {
        const uint32_t run = rv[info.n_runs - 1 - r];     // the walk stored them end -> start
        const int len = (int)(run >> 2);
        const uint32_t op = run & 3u;
        if (op == 0) {
            for (int x = lane; x < len; x += FIN_LANES) ident += (qs[i + x] == ts[j + x]) ? 1u : 0u;
            i += len; j += len;
        } else if (op == 1) i += len;
        else j += len;
    }
    for (int d = FIN_LANES / 2; d > 0; d >>= 1) ident += __shfl_xor(ident, d, 64);
    if (lane == 0) {
        const double idp = (double)ident * 100.0 / (double)info.aln_len;
        const double qcov = (double)(info.iend - info.istart + 1) * 100.0 / (double)q_len[q];
        sel[s].n_ident = ident;
        sel[s].pass = (idp >= min_id_pct && qcov >= min_qcov_pct) ? 1u : 0u;
    }
}

// hsp_mode 1: two bands of one (q, t) that end in the same cell found the same alignment: keep the higher score, then the
// lower bin (= lower index, sel is ordered by (q, t, bin)); the decision ignores whether the other one survives its filters
__global__ __launch_bounds__(256) void dedupe_bands(const uint32_t *__restrict__ d_n_sel, SelInfo *__restrict__ sel, const uint64_t *__restrict__ cands)
{
    const uint64_t n_sel = *d_n_sel;
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_sel) return;
    const SelInfo me = sel[s];
    const uint64_t g = cands[me.cand] >> 18;
    bool dead = false;
    for (int dir = -1; dir <= 1 && !dead; dir += 2) {
        for (int64_t x = (int64_t)s + dir; x >= 0 && x < (int64_t)n_sel; x += dir) {
            const SelInfo o = sel[x];
            if ((cands[o.cand] >> 18) != g) break;
            if (o.iend == me.iend && o.jend == me.jend && (o.score > me.score || (o.score == me.score && x < (int64_t)s))) { dead = true; break; }
        }
    }
    if (dead) sel[s].pad = 1u;
}

// hsp_mode 2 (BLAST's way with the HSPs of one subject strand, oracle/align_oracle.c align_group): the alignments of a (q, t) group in the order score
// descending, position ascending; one is dropped when an ACCEPTED one in front of it shares its start cell or its end cell, or holds its query range and
// its subject range inside its own.  Order-dependent, so the group's first thread works the group off alone; groups are a handful of bands (a repeat
// family makes dozens).  pad: 0 not looked at yet, 2 accepted (back to 0 at the end), 1 dropped.
__global__ __launch_bounds__(256) void cull_hsps(const uint32_t *__restrict__ d_n_sel, SelInfo *__restrict__ sel, const uint64_t *__restrict__ cands)
{
    const int64_t n_sel = (int64_t)*d_n_sel;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_sel) return;
    const uint64_t g = cands[sel[s].cand] >> 18;
    if (s > 0 && (cands[sel[s - 1].cand] >> 18) == g) return;
    int64_t e = s + 1;
    while (e < n_sel && (cands[sel[e].cand] >> 18) == g) ++e;
    if (e - s == 1) return;
    for (int64_t it = s; it < e; ++it) {
        int64_t best = -1;
        int32_t best_score = 0;
        for (int64_t x = s; x < e; ++x)
            if (sel[x].pad == 0u && (best < 0 || sel[x].score > best_score)) { best = x; best_score = sel[x].score; }
        const SelInfo me = sel[best];
        bool dead = false;
        for (int64_t y = s; y < e && !dead; ++y) {
            if (sel[y].pad != 2u) continue;
            const SelInfo o = sel[y];
            dead = (o.iend == me.iend && o.jend == me.jend) || (o.istart == me.istart && o.jstart == me.jstart) ||
                   (o.istart <= me.istart && me.iend <= o.iend && o.jstart <= me.jstart && me.jend <= o.jend);
        }
        sel[best].pad = dead ? 1u : 2u;
    }
    for (int64_t x = s; x < e; ++x)
        if (sel[x].pad == 2u) sel[x].pad = 0u;
}

// hsp_mode 2, in front of topk: keep = 1 for the best alignment of its subject among the query's passing alignments of one competition class
// (score descending, position ascending - the order topk ranks in)
__global__ __launch_bounds__(256) void subject_best(const uint32_t *__restrict__ d_n_sel, SelInfo *__restrict__ sel, const uint64_t *__restrict__ cands, int n_splits, uint32_t t_base,
                                                    const uint32_t *__restrict__ t_class, const uint32_t *__restrict__ t_subject)
{
    const int64_t n_sel = (int64_t)*d_n_sel;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_sel) return;
    const SelInfo me = sel[s];
    uint32_t best = 0;
    if (me.pass) {
        const uint64_t key = cands[me.cand];
        const uint32_t q = key_q(key), t = key_t(key), split = t_class ? t_class[t] : (t + t_base) % (uint32_t)n_splits, sj = t_subject[t];
        best = 1;
        for (int dir = -1; dir <= 1 && best; dir += 2) {
            for (int64_t x = s + dir; x >= 0 && x < n_sel; x += dir) {
                const SelInfo o = sel[x];
                const uint64_t ko = cands[o.cand];
                if (key_q(ko) != q) break;
                const uint32_t to = key_t(ko);
                if (!o.pass || t_subject[to] != sj || (t_class ? t_class[to] : (to + t_base) % (uint32_t)n_splits) != split) continue;
                if (o.score > me.score || (o.score == me.score && x < s)) { best = 0; break; }
            }
        }
    }
    sel[s].keep = best;
}

__global__ __launch_bounds__(256) void apply_dedupe(const uint32_t *__restrict__ d_n_sel, SelInfo *__restrict__ sel)
{
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < *d_n_sel && sel[s].pad) sel[s].pass = 0u;
}

// rank inside (query, target mod n_splits): score desc, target asc, band asc.  sel is ordered by (q, t, bin).
// t_class (optional): competition class of every target = group * n_splits + (index inside the group) % n_splits, so that a
// batch of reference sets (genomes) searched at once ranks exactly as if each had been searched alone
__global__ __launch_bounds__(256) void topk(const uint32_t *__restrict__ d_n_sel, SelInfo *__restrict__ sel, const uint64_t *__restrict__ cands, int top_k, int n_splits, uint32_t t_base,
                                            const uint32_t *__restrict__ t_class, uint32_t *__restrict__ keep_flag, uint32_t *__restrict__ hit_pos, uint64_t *__restrict__ cig_pos,
                                            const unsigned long long *__restrict__ score_hdr, const unsigned long long *__restrict__ trace_hdr, unsigned long long *__restrict__ mail,
                                            SelectState st, const uint32_t *__restrict__ t_subject = nullptr)
{
    // top-k decision per selected pair AND the two compactions behind it (slot of the hit record, slot of its CIGAR runs) in one launch:
    // the tile scans (kept, runs of the kept) and finds the totals in front of it by look-back (lookback.h)
    __shared__ uint32_t s_tile, lds32[4];
    __shared__ uint64_t lds64[4], s_pre[2];
    const uint32_t tile = lb_take_tile(st.count_state, st.ticket_base, &s_tile);
    const uint64_t n_sel = *d_n_sel;
    const uint64_t s = (uint64_t)tile * 256 + threadIdx.x;
    if (s == 0) {
        // the totals of the two Smith-Waterman passes into the block the host reads with its last synchronisation (one copy for everything)
        mail[4] = score_hdr[0]; mail[5] = score_hdr[1];
        mail[6] = trace_hdr ? trace_hdr[0] : 0ull; mail[7] = trace_hdr ? trace_hdr[1] : 0ull; mail[8] = trace_hdr ? trace_hdr[5] : 0ull;
        mail[9] = score_hdr[6];                                                  // cells of the candidates the score pass settled without a sweep,
        reinterpret_cast<uint32_t *>(mail)[3] = (uint32_t)score_hdr[7];          // their number
    }
    uint32_t keep = 0;
    uint64_t n_runs = 0;
    if (s < n_sel) {
        const SelInfo me = sel[s];
        if (me.pass) {
            const uint64_t key = cands[me.cand];
            const uint32_t q = key_q(key), t = key_t(key), split = t_class ? t_class[t] : (t + t_base) % (uint32_t)n_splits;
            uint32_t rank = 0;
            if (t_subject) {
                // hsp_mode 2: top_k counts subjects.  This alignment's subject is as good as its best alignment (subject_best marked it: keep = 1);
                // its rank = the subjects of the same query and class whose best alignment stands in front of that one
                const uint32_t sj = t_subject[t];
                int64_t mine = (int64_t)s;
                int32_t mine_score = me.score;
                if (!me.keep)
                    for (int dir = -1; dir <= 1 && mine == (int64_t)s; dir += 2)
                        for (int64_t x = (int64_t)s + dir; x >= 0 && x < (int64_t)n_sel; x += dir) {
                            const SelInfo o = sel[x];
                            const uint64_t ko = cands[o.cand];
                            if (key_q(ko) != q) break;
                            const uint32_t to = key_t(ko);
                            if (o.pass && o.keep && t_subject[to] == sj && (t_class ? t_class[to] : (to + t_base) % (uint32_t)n_splits) == split) { mine = x; mine_score = o.score; break; }
                        }
                for (int dir = -1; dir <= 1; dir += 2) {
                    for (int64_t x = (int64_t)s + dir; x >= 0 && x < (int64_t)n_sel; x += dir) {
                        const SelInfo o = sel[x];
                        const uint64_t ko = cands[o.cand];
                        if (key_q(ko) != q) break;
                        const uint32_t to = key_t(ko);
                        if (!o.pass || !o.keep || t_subject[to] == sj || (t_class ? t_class[to] : (to + t_base) % (uint32_t)n_splits) != split) continue;
                        if (o.score > mine_score || (o.score == mine_score && x < mine)) ++rank;
                    }
                }
            } else
            for (int dir = -1; dir <= 1; dir += 2) {
                for (int64_t x = (int64_t)s + dir; x >= 0 && x < (int64_t)n_sel; x += dir) {
                    const SelInfo o = sel[x];
                    const uint64_t ko = cands[o.cand];
                    if (key_q(ko) != q) break;
                    const uint32_t to = key_t(ko);
                    if (!o.pass || (t_class ? t_class[to] : (to + t_base) % (uint32_t)n_splits) != split) continue;
                    if (o.score > me.score || (o.score == me.score && (to < t || (to == t && x < (int64_t)s)))) ++rank;
                }
            }
            keep = rank < (uint32_t)top_k ? 1u : 0u;
        }
        n_runs = keep ? me.n_runs : 0;
    }
    uint32_t tot_k;
    uint64_t tot_r;
    const uint32_t ex_k = block_excl_scan_256<uint32_t>(keep, &tot_k, lds32);
    const uint64_t ex_r = block_excl_scan_256<uint64_t>(n_runs, &tot_r, lds64);
    if (threadIdx.x < 64) {
        const uint64_t p0 = lb_tile_prefix<32>(st.count_state + 1, tile, tot_k, st.epoch_count, (int)threadIdx.x);
        const uint64_t p1 = lb_tile_prefix<48>(st.run_state + 1, tile, tot_r, st.epoch_run, (int)threadIdx.x);
        if (threadIdx.x == 0) { s_pre[0] = p0; s_pre[1] = p1; }
    }
    __syncthreads();
    if (s < n_sel) {
        keep_flag[s] = keep;
        hit_pos[s] = (uint32_t)(s_pre[0] + ex_k);
        cig_pos[s] = s_pre[1] + ex_r;
    }
    if (tile == gridDim.x - 1 && threadIdx.x == 255) {          // the last tile: number of hits, number of their CIGAR runs
        reinterpret_cast<uint32_t *>(mail)[2] = (uint32_t)(s_pre[0] + tot_k);
        mail[2] = s_pre[1] + tot_r;
    }
}

__global__ __launch_bounds__(256) void emit(const uint32_t *__restrict__ d_n_sel, const SelInfo *__restrict__ sel, const uint64_t *__restrict__ cands,
                                            const uint32_t *__restrict__ keep_flag, const uint32_t *__restrict__ hit_pos, const uint64_t *__restrict__ cig_pos,
                                            const uint64_t *__restrict__ run_off, const uint32_t *__restrict__ runs, const uint32_t *__restrict__ nblk,
                                            const uint32_t *__restrict__ q_len, const uint32_t *__restrict__ t_len,
                                            pep_hit *__restrict__ hits, uint32_t *__restrict__ cigar,
                                            const unsigned long long *__restrict__ mail = nullptr, unsigned long long *__restrict__ mail_copy = nullptr)
{
    const int lane = threadIdx.x & 63;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // the counter block of the stage (final since topk) once more, outside the search's zeroed region: pack_out and K10 read it there, so that
    // pack_out may clear that region for the next search
    if (mail_copy && blockIdx.x == 0 && threadIdx.x < 10) mail_copy[threadIdx.x] = mail[threadIdx.x];
    if (s >= *d_n_sel || !keep_flag[s]) return;
    const SelInfo info = sel[s];
    const uint64_t key = cands[info.cand];
    const uint64_t co = cig_pos[s];
    const uint32_t *rv = runs + run_off[s];
    for (uint32_t r = lane; r < info.n_runs; r += 64) cigar[co + r] = rv[info.n_runs - 1 - r];
    // exact in-band in-matrix cell count of the winning band: the 128 diagonals are spread over the lanes
    const uint32_t q = key_q(key), t = key_t(key);
    const int dlo = key_dlo(key), Lq = (int)q_len[q], Lt = (int)t_len[t];
    const int dl = max(dlo, -(Lq - 1)), dh = min(dlo + 127, Lt - 1);
    unsigned long long cells = 0;
    for (int d = dl + lane; d <= dh; d += 64) cells += (unsigned long long)(min(Lq - 1, Lt - 1 - d) - max(0, -d) + 1);
    for (int x = 32; x > 0; x >>= 1) cells += __shfl_xor(cells, x, 64);
    if (lane == 0) {
        pep_hit h;
        h.q = q; h.t = t;
        h.q_start = (uint32_t)info.istart + 1; h.q_end = (uint32_t)info.iend + 1;
        h.t_start = (uint32_t)info.jstart + 1; h.t_end = (uint32_t)info.jend + 1;
        h.score = info.score; h.n_ident = info.n_ident; h.aln_len = info.aln_len; h.nm = info.aln_len - info.n_ident;
        h.cigar_runs = info.n_runs; h.bin = (int32_t)(key & ((1u << 18) - 1)); h.cigar_off = co;
        h.cells = cells;
        hits[hit_pos[s]] = h;
    }
    (void)nblk;
}

// The result on its way to the host WITHOUT the host knowing its size: the counter block (72 bytes: hits, CIGAR runs, statistics), the
// hit records and the CIGAR arena are written into the context's pinned staging area by this kernel - header (PACK_HEADER bytes), hits,
// arena back to back - so that the search needs no synchronisation between its candidate count and its end.  A staging area that is too
// small (the first search of a context, a result that grew) gets the header only, with the overflow word set: the host then sizes it
// and copies the classic way.
constexpr size_t PACK_HEADER = 128;
// Two chores ride along (a launch of their own costs 4 - 5 us each): the counters of the NEXT search are cleared (zero_words: the region every
// search needs zeroed - this is its last reader), and the union-find that follows gets its parent array initialised (iota).
__global__ __launch_bounds__(256) void pack_out(const unsigned long long *__restrict__ mail, const pep_hit *__restrict__ hits, const uint32_t *__restrict__ cigar,
                                                unsigned char *__restrict__ pinned, unsigned long long cap, uint32_t *__restrict__ zero_words, uint32_t n_zero_words,
                                                uint32_t *__restrict__ iota, uint32_t n_iota)
{
    for (uint32_t x = blockIdx.x * 256 + threadIdx.x; x < n_zero_words; x += gridDim.x * 256) zero_words[x] = 0u;
    for (uint32_t x = blockIdx.x * 256 + threadIdx.x; x < n_iota; x += gridDim.x * 256) iota[x] = x;
    const uint32_t n_hits = reinterpret_cast<const uint32_t *>(mail)[2];
    const unsigned long long n_cig = mail[2];
    const bool fits = PACK_HEADER + (unsigned long long)n_hits * sizeof(pep_hit) + n_cig * 4 <= cap;
    unsigned long long *head = reinterpret_cast<unsigned long long *>(pinned);
    if (blockIdx.x == 0 && threadIdx.x < 16) head[threadIdx.x] = threadIdx.x < 10 ? mail[threadIdx.x] : (threadIdx.x == 15 ? (fits ? 0ull : 1ull) : 0ull);
    if (!fits) return;
    const uint64_t stride = (uint64_t)gridDim.x * 256, t0 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint4 *src = reinterpret_cast<const uint4 *>(hits);
    uint4 *dst = reinterpret_cast<uint4 *>(pinned + PACK_HEADER);
    for (uint64_t x = t0; x < (uint64_t)n_hits * (sizeof(pep_hit) / 16); x += stride) dst[x] = src[x];
    uint32_t *cdst = reinterpret_cast<uint32_t *>(pinned + PACK_HEADER + (size_t)n_hits * sizeof(pep_hit));
    for (uint64_t x = t0; x < n_cig; x += stride) cdst[x] = cigar[x];
}

}  // namespace

// workspace slots: ws[16] flag, ws[17] pos, ws[18] best_idx, ws[19] sel, ws[20] run_cap/run_off (u64 x2) + selected keys,
//                  ws[21] runs, ws[22] keep arrays, ws[23] output hits + cigar, ws[9] small counters
//
// Host round trips.  Between the seed stage's count of candidates and the sizes of the result (hits, CIGAR runs) the host does not need
// to see anything: the number of selected pairs stays on the device (every kernel behind the selection reads it there and is launched on
// a grid sized for all n candidates), the traceback area and the run arena are sized from upper bounds (every candidate selected, every
// pair as long as the longest), and the statistics ride on the one read-back at the end.  When those bounds would cost more memory than
// is sensible (FAST_DIR_BYTES / FAST_RUN_BYTES: long sequences times many candidates - searches that are long enough not to care about
// two host round trips), or with params.reserved2 = 1 (tests), the stage synchronises after the selection and sizes everything exactly.
struct HostMail { uint32_t n_pairs, n_sel, n_hits, n_settled; unsigned long long n_cig, run_cap, score_cells, score_blocks, trace_cells, trace_blocks, trace_swept, settled_cells; };
static_assert(sizeof(HostMail) == 80, "layout of the selection stage's counter block");

// the host's half of a search whose result left through pack_out: sizes, statistics and the views of the staging area - called once the
// stream has been waited for (by pep_search behind its one final wait, or by pep_extend itself when the caller wants the result at once)
int pep_extend_finish(pep_ctx *ctx)
{
    if (!ctx->ext.pending) return PEP_OK;
    ctx->ext.pending = false;
    pep_result *res = ctx->ext.res;
    HostMail h;
    std::memcpy(&h, ctx->pin_stage.p, sizeof(h));
    const bool overflow = reinterpret_cast<const unsigned long long *>(ctx->pin_stage.p)[15] != 0;
    ctx->stats.cells += h.score_cells;
    ctx->stats.cells_swept += h.score_blocks * 16 * 64;
    ctx->stats.candidates_settled += h.n_settled;
    ctx->stats.cells_settled += h.settled_cells;
    ctx->stats.cells_trace += h.trace_cells;
    ctx->stats.cells_swept_trace += h.trace_blocks * 16 * 64;
    ctx->stats.dir_bytes += h.trace_blocks * 512;
    ctx->trace_swept = h.trace_swept;
    ctx->stats.tracebacks_gapless = h.n_sel - h.trace_swept;
    ctx->stats.pairs = h.n_pairs;
    ctx->stats.tracebacks = h.n_sel;
    ctx->stats.hits = h.n_hits;
    if (h.n_hits == 0) return PEP_OK;
    const size_t hb = (size_t)h.n_hits * sizeof(pep_hit);
    const pep_hit *d_hits = reinterpret_cast<const pep_hit *>(ctx->ext.d_hits);
    const uint32_t *d_cig = reinterpret_cast<const uint32_t *>(ctx->ext.d_cig);
    if (!overflow) {
        res->st_hits = reinterpret_cast<const pep_hit *>(ctx->pin_stage.p + PACK_HEADER);
        res->st_cigar = reinterpret_cast<const uint32_t *>(ctx->pin_stage.p + PACK_HEADER + hb);
    } else if (pin_reserve(ctx, ctx->pin_stage, PACK_HEADER + hb + (h.n_cig + 1) * 4) == PEP_OK) {
        // the staging area was too small for this result (now it is not: the next search of this size goes through pack_out)
        PEP_HIP(ctx, hipMemcpyAsync(ctx->pin_stage.p + PACK_HEADER, d_hits, hb, hipMemcpyDeviceToHost, ctx->stream));
        if (h.n_cig) PEP_HIP(ctx, hipMemcpyAsync(ctx->pin_stage.p + PACK_HEADER + hb, d_cig, h.n_cig * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, pep_stream_wait(ctx));
        res->st_hits = reinterpret_cast<const pep_hit *>(ctx->pin_stage.p + PACK_HEADER);
        res->st_cigar = reinterpret_cast<const uint32_t *>(ctx->pin_stage.p + PACK_HEADER + hb);
    } else {
        res->hits.resize(h.n_hits);
        res->cigar.resize(h.n_cig);
        PEP_HIP(ctx, hipMemcpyAsync(res->hits.data(), d_hits, hb, hipMemcpyDeviceToHost, ctx->stream));
        if (h.n_cig) PEP_HIP(ctx, hipMemcpyAsync(res->cigar.data(), d_cig, h.n_cig * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    res->n_hits = h.n_hits; res->n_cigar = h.n_cig;
    res->d_hits = d_hits; res->d_cigar = d_cig;
    return PEP_OK;
}

int pep_extend(pep_ctx *ctx, const uint64_t *d_cands, uint64_t n, const int32_t *h_min_score, pep_result *res, bool defer)
{
    const pep_search_params &P = ctx->params;
    pep_drop_dev_result(ctx);                     // ws[23] is about to be rewritten
    res->hits.clear();
    res->cigar.clear();
    ctx->stats.candidates = n;
    ctx->stats.pairs = ctx->stats.tracebacks = ctx->stats.hits = 0;
    ctx->stats.cells = ctx->stats.cells_swept = ctx->stats.dir_bytes = 0;
    ctx->stats.candidates_settled = ctx->stats.cells_settled = 0;
    if (n == 0) return PEP_OK;
    hipStream_t st = ctx->stream;
    // ---- pass 1: score-only banded SW over every candidate
    // per-query score thresholds: uploaded from pinned memory BEFORE the score pass is queued (a copy from pageable memory makes the
    // host wait for everything queued ahead of it - behind the score pass that was the whole pass)
    DevBuf &dms = ctx->d_min_score;
    if (h_min_score) {                            // (nullptr: the caller has uploaded them already)
        PEP_TRY(dev_reserve(ctx, dms, (size_t)(ctx->q.n + 1) * 4));
        PEP_TRY(pin_reserve(ctx, ctx->pin_ms, (size_t)(ctx->q.n + 1) * 4));
        std::memcpy(ctx->pin_ms.p, h_min_score, (size_t)ctx->q.n * 4);
        PEP_HIP(ctx, hipMemcpyAsync(dms.p, ctx->pin_ms.p, (size_t)ctx->q.n * 4, hipMemcpyHostToDevice, st));
    }
    unsigned long long *score_hdr = nullptr, *trace_hdr = nullptr;
    PEP_TRY(pep_sw_run(ctx, d_cands, n, false, nullptr, nullptr, nullptr, nullptr, 0, &score_hdr));      // (the pass times come from the context's phase timers, read after the search)

    pep_timer_begin(ctx, TM_TRACE);

    const int4 *sw = ctx->ws[12].as<const int4>();
    // the block the host reads: u32 [0] (q, t) pairs, [1] selected pairs, [2] hits; u64 [2] CIGAR runs of the hits, [3] run capacity of the selected
    // pairs, [4] [5] cells / 16-step blocks of the score pass, [6] [7] [8] cells / blocks / swept pairs of the traceback pass
    void *zb = nullptr;
    PEP_TRY(pep_zero_block(ctx, PEP_ZC_SELECT, PEP_ZERO_SELECT, 256, &zb));
    uint32_t *counters = reinterpret_cast<uint32_t *>(zb);
    unsigned long long *mail = reinterpret_cast<unsigned long long *>(zb);
    const uint32_t *d_n_sel = counters + 1;
    HostMail h_mail;
    std::memset(&h_mail, 0, sizeof(h_mail));
    ctx->ext.pending = false;
    const unsigned gb = (unsigned)ceil_div(n, 256);
    // selection, compaction and the run slots of the selected pairs in one launch (select_gather); the arrays of the selected pairs are laid
    // out for n entries (every candidate selected) whatever the count turns out to be
    PEP_TRY(dev_reserve(ctx, ctx->ws[19], (size_t)n * sizeof(SelInfo)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[20], ((size_t)n + 2) * 8 * 2));
    PEP_TRY(dev_reserve(ctx, ctx->ws[8], ((size_t)n + 2) * 4 * 2));      // ws[8]: raw seed hits of K4, free again
    int32_t *known = ctx->ws[8].as<int32_t>(), *end_lane = known + n + 2;
    SelInfo *sel = ctx->ws[19].as<SelInfo>();
    uint64_t *run_off = ctx->ws[20].as<uint64_t>(), *sel_keys = run_off + n + 2;
    {
        SelectState ss;
        uint64_t ep = 0;
        uint32_t tb2 = 0;
        PEP_TRY(pep_lookback_begin(ctx, ctx->fused_state[0], gb, (1u << 30) - 1, &ss.count_state, &ss.ticket_base, &ss.epoch_count));
        PEP_TRY(pep_lookback_begin(ctx, ctx->fused_state[1], gb, (1u << 14) - 1, &ss.run_state, &tb2, &ep));
        ss.epoch_run = ep;
        hipLaunchKernelGGL(select_gather, dim3(gb), dim3(256), 0, st, d_cands, n, sw, dms.as<const int32_t>(), P.hsp_mode, ctx->q.len.as<const uint32_t>(),
                           ctx->t.len.as<const uint32_t>(), sel, run_off, sel_keys, known, end_lane, counters, mail + 3, ss);
    }
    // upper bounds for the buffers of the traceback stage
    constexpr uint64_t FAST_DIR_BYTES = 8ull << 30, FAST_RUN_BYTES = 2ull << 30;
    const uint64_t max_blk = ((uint64_t)ctx->q.max_len + ctx->t.max_len) / 16 + 3;
    const uint64_t dir_blocks_bound = n * max_blk, run_bound = n * (2ull * std::min(ctx->q.max_len, ctx->t.max_len) + 2);
    const bool fast = (P.reserved2 & 1) == 0 && dir_blocks_bound * 512 <= FAST_DIR_BYTES && run_bound * 4 <= FAST_RUN_BYTES;
    uint64_t n_b = n;                            // what the buffers and grids behind the selection are sized for
    if (!fast) {
        PEP_TRY(pep_read_back(ctx, &h_mail, mail, 8));
        PEP_TRY(pep_sync_reads(ctx));
        n_b = h_mail.n_sel;
    }
    uint32_t n_hits = 0;
    uint64_t n_cig = 0;
    uint64_t *cig_pos = nullptr;
    uint32_t *runs = nullptr, *keep_flag = nullptr, *hit_pos = nullptr;
    if (n_b) {
        uint64_t total_runs = run_bound;
        if (!fast) PEP_TRY(pep_read_back(ctx, &total_runs, mail + 3, 8));         // arrives with the synchronisation inside pep_sw_run (block total)
        // ---- rule 5a: pairs whose alignment is one ungapped run are settled without a sweep (the score-pass results in ws[12] have been
        // consumed by gather_sel: the slots of the selected pairs are written from here on)
        PEP_TRY(dev_reserve(ctx, ctx->d_trace_mode, ((size_t)n_b + 1) * sizeof(int32_t)));
        PEP_TRY(pep_upload_sub_table(ctx));                                 // (K9 drives this stage without the seed stage in front)
        hipLaunchKernelGGL(gapless_check, dim3((unsigned)ceil_div(n_b, 4)), dim3(256), 0, st, d_n_sel, (const uint64_t *)sel_keys, (const int32_t *)known,
                           (const int32_t *)end_lane, ctx->q.res.as<const uint8_t>(), ctx->q.off.as<const uint32_t>(), ctx->q.len.as<const uint32_t>(),
                           ctx->t.res.as<const uint8_t>(), ctx->t.off.as<const uint32_t>(), ctx->t.len.as<const uint32_t>(), ctx->d_params.as<const int8_t>(),
                           ctx->ws[12].as<int4>(), ctx->d_trace_mode.as<int32_t>());
        // ---- pass 2: the same DP with traceback codes, the remaining selected pairs only
        PEP_TRY(pep_sw_run(ctx, sel_keys, n_b, true, known, end_lane, ctx->d_trace_mode.as<const int32_t>(), d_n_sel, fast ? dir_blocks_bound : 0, &trace_hdr));
        const int4 *sw2 = ctx->ws[12].as<const int4>();
        PEP_TRY(dev_reserve(ctx, ctx->ws[21], (total_runs + 1) * 4));
        runs = ctx->ws[21].as<uint32_t>();
        const unsigned gfin = (unsigned)ceil_div(n_b, 256 / FIN_LANES);
        hipLaunchKernelGGL(walk, dim3((unsigned)ceil_div(n_b, WALK_LANES)), dim3(WALK_LANES), 0, st, d_n_sel, sel, (const uint64_t *)sel_keys, sw2, ctx->ws[11].as<const uint64_t>(),
                           ctx->ws[13].as<const uint32_t>(), ctx->d_trace_mode.as<const int32_t>(), (const uint64_t *)run_off, runs);
        hipLaunchKernelGGL(finalize, dim3(gfin), dim3(256), 0, st, d_n_sel, sel, (const uint64_t *)sel_keys, ctx->q.res.as<const uint8_t>(),
                           ctx->q.off.as<const uint32_t>(), ctx->q.len.as<const uint32_t>(), ctx->t.res.as<const uint8_t>(),
                           ctx->t.off.as<const uint32_t>(), (const uint64_t *)run_off, (const uint32_t *)runs, P.min_id_pct, P.min_qcov_pct);
        const uint32_t *t_subject = nullptr;
        if (P.hsp_mode == 1) {
            hipLaunchKernelGGL(dedupe_bands, dim3((unsigned)ceil_div(n_b, 256)), dim3(256), 0, st, d_n_sel, sel, (const uint64_t *)sel_keys);
            hipLaunchKernelGGL(apply_dedupe, dim3((unsigned)ceil_div(n_b, 256)), dim3(256), 0, st, d_n_sel, sel);
        } else if (P.hsp_mode == 2) {
            hipLaunchKernelGGL(cull_hsps, dim3((unsigned)ceil_div(n_b, 256)), dim3(256), 0, st, d_n_sel, sel, (const uint64_t *)sel_keys);
            hipLaunchKernelGGL(apply_dedupe, dim3((unsigned)ceil_div(n_b, 256)), dim3(256), 0, st, d_n_sel, sel);
            // the subject of every target: the reference sequence it is a strand / frame / chunk of (K1's table)
            PEP_TRY(pep_k1_host_tables(ctx));
            if (ctx->t_meta.size() != ctx->t.n) return pep_fail(ctx, PEP_ERR_STATE, "hsp_mode 2 needs the targets' sequence table (targets made by pep_translate / pep_use_nt_as_residues)");
            std::vector<uint32_t> subj(ctx->t.n + 1, 0u);
            for (uint32_t t = 0; t < ctx->t.n; ++t) subj[t] = ctx->t_meta[t].seq;
            PEP_TRY(dev_reserve(ctx, ctx->d_t_subject, ((size_t)ctx->t.n + 1) * 4));
            PEP_TRY(pep_h2d(ctx, ctx->d_t_subject.p, subj.data(), ((size_t)ctx->t.n + 1) * 4));
            t_subject = ctx->d_t_subject.as<const uint32_t>();
            hipLaunchKernelGGL(subject_best, dim3((unsigned)ceil_div(n_b, 256)), dim3(256), 0, st, d_n_sel, sel, (const uint64_t *)sel_keys, P.n_splits, (uint32_t)(P.t_index_base % P.n_splits),
                               ctx->t_class_ready ? ctx->d_t_class.as<const uint32_t>() : (const uint32_t *)nullptr, t_subject);
        }
        // top-k, then compaction of hits and CIGAR runs
        PEP_TRY(dev_reserve(ctx, ctx->ws[22], ((size_t)n_b + 2) * (4 + 4 + 8)));
        keep_flag = ctx->ws[22].as<uint32_t>(); hit_pos = keep_flag + n_b + 2;
        cig_pos = reinterpret_cast<uint64_t *>(hit_pos + n_b + 2);
        {
            SelectState ts;
            uint64_t ep = 0;
            uint32_t tb2 = 0;
            const uint64_t tiles = ceil_div(n_b, 256);
            PEP_TRY(pep_lookback_begin(ctx, ctx->fused_state[2], tiles, (1u << 30) - 1, &ts.count_state, &ts.ticket_base, &ts.epoch_count));
            PEP_TRY(pep_lookback_begin(ctx, ctx->fused_state[3], tiles, (1u << 14) - 1, &ts.run_state, &tb2, &ep));
            ts.epoch_run = ep;
            hipLaunchKernelGGL(topk, dim3((unsigned)tiles), dim3(256), 0, st, d_n_sel, sel, (const uint64_t *)sel_keys, P.top_k, P.n_splits,
                               (uint32_t)(P.t_index_base % P.n_splits), ctx->t_class_ready ? ctx->d_t_class.as<const uint32_t>() : (const uint32_t *)nullptr, keep_flag, hit_pos, cig_pos,
                               (const unsigned long long *)score_hdr, (const unsigned long long *)trace_hdr, mail, ts, t_subject);
        }
        if (fast && !ctx->device_results && pin_reserve(ctx, ctx->pin_stage, PACK_HEADER) == PEP_OK) {
            // the result leaves through pack_out: output buffers from the same upper bounds, no look at the sizes, no synchronisation here
            const size_t hb_bound = (size_t)n_b * sizeof(pep_hit);
            PEP_TRY(dev_reserve(ctx, ctx->ws[23], hb_bound + (run_bound + 1) * 4));
            pep_hit *d_hits = ctx->ws[23].as<pep_hit>();
            uint32_t *d_cig = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ctx->ws[23].p) + hb_bound);
            PEP_TRY(dev_reserve(ctx, ctx->d_mail_copy, 128));
            unsigned long long *mail_copy = ctx->d_mail_copy.as<unsigned long long>();
            uint32_t *parent = nullptr;
            if (ctx->grp_nodes) { PEP_TRY(dev_reserve(ctx, ctx->ws[0], (size_t)ctx->grp_nodes * 4)); parent = ctx->ws[0].as<uint32_t>(); }      // (K10 follows: pep_k10_queue)
            hipLaunchKernelGGL(emit, dim3((unsigned)ceil_div(n_b, 4)), dim3(256), 0, st, d_n_sel, (const SelInfo *)sel, (const uint64_t *)sel_keys, (const uint32_t *)keep_flag,
                               (const uint32_t *)hit_pos, (const uint64_t *)cig_pos, (const uint64_t *)run_off, (const uint32_t *)runs,
                               ctx->ws[10].as<const uint32_t>(), ctx->q.len.as<const uint32_t>(), ctx->t.len.as<const uint32_t>(), d_hits, d_cig,
                               (const unsigned long long *)mail, mail_copy);
            hipLaunchKernelGGL(pack_out, dim3(1024), dim3(256), 0, st, (const unsigned long long *)mail_copy, (const pep_hit *)d_hits, (const uint32_t *)d_cig,
                               ctx->pin_stage.p, (unsigned long long)ctx->pin_stage.cap, ctx->d_zero.as<uint32_t>(), (uint32_t)(PEP_ZERO_TOTAL / 4), parent, ctx->grp_nodes);
            PEP_HIP(ctx, hipGetLastError());
            ctx->ext.pending = true; ctx->ext.res = res; ctx->ext.d_hits = d_hits; ctx->ext.d_cig = d_cig;
            ctx->ext.d_n_hits = reinterpret_cast<const uint32_t *>(mail_copy) + 2; ctx->ext.n_bound = n_b;
            ctx->ext.parent_ready = parent != nullptr;
            for (bool &f : ctx->zero_ok) f = true;   // as of this point of the stream every counter block is zero again (pack_out): whatever is queued
            ctx->zero_clean = true;                  // behind it - the next search, K9's alignment stage - needs no fill
            pep_timer_end(ctx, TM_TRACE);
            if (defer) return PEP_OK;
            PEP_HIP(ctx, pep_stream_wait(ctx));
            return pep_extend_finish(ctx);
        }
        PEP_TRY(pep_read_back(ctx, &h_mail, mail, sizeof(h_mail)));
        PEP_TRY(pep_sync_reads(ctx));
        n_hits = h_mail.n_hits; n_cig = h_mail.n_cig;
        ctx->stats.cells += h_mail.score_cells;
        ctx->stats.cells_swept += h_mail.score_blocks * 16 * 64;
        ctx->stats.candidates_settled += h_mail.n_settled;
        ctx->stats.cells_settled += h_mail.settled_cells;
        if (fast) {                                  // (the exactly sized pass has added its totals itself)
            ctx->stats.cells_trace += h_mail.trace_cells;
            ctx->stats.cells_swept_trace += h_mail.trace_blocks * 16 * 64;
            ctx->stats.dir_bytes += h_mail.trace_blocks * 512;
            ctx->trace_swept = h_mail.trace_swept;
        }
        ctx->stats.tracebacks_gapless = h_mail.n_sel - ctx->trace_swept;   // (counted by the pass's set-up kernel: the pairs that entered the sweep)
    } else {
        unsigned long long h_score[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        PEP_TRY(pep_read_back(ctx, h_score, score_hdr, sizeof(h_score)));
        PEP_TRY(pep_sync_reads(ctx));
        ctx->stats.cells += h_score[0];
        ctx->stats.cells_swept += h_score[1] * 16 * 64;
        ctx->stats.cells_settled += h_score[6];
        ctx->stats.candidates_settled += h_score[7];
    }
    ctx->stats.pairs = h_mail.n_pairs;
    ctx->stats.tracebacks = h_mail.n_sel;
    ctx->stats.hits = n_hits;
    if (n_hits) {
        const uint32_t n_sel = h_mail.n_sel;
        const size_t hb = (size_t)n_hits * sizeof(pep_hit);
        PEP_TRY(dev_reserve(ctx, ctx->ws[23], hb + (n_cig + 1) * 4));
        pep_hit *d_hits = ctx->ws[23].as<pep_hit>();
        uint32_t *d_cig = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ctx->ws[23].p) + hb);
        hipLaunchKernelGGL(emit, dim3((unsigned)ceil_div(n_sel, 4)), dim3(256), 0, st, d_n_sel, (const SelInfo *)sel, (const uint64_t *)sel_keys, (const uint32_t *)keep_flag,
                           (const uint32_t *)hit_pos, (const uint64_t *)cig_pos, (const uint64_t *)run_off, (const uint32_t *)runs,
                           ctx->ws[10].as<const uint32_t>(), ctx->q.len.as<const uint32_t>(), ctx->t.len.as<const uint32_t>(), d_hits, d_cig);
        PEP_HIP(ctx, hipGetLastError());
        // the table goes to the context's pinned staging area (DMA speed, no page faults); pep_result_copy reads it from there.
        // If the host refuses that much pinned memory the result owns ordinary vectors instead.  In device-result mode
        // (pep_set_result_mode) nothing is copied here: whoever wants the host copy fetches it later (pep_fetch_result).
        if (ctx->device_results) {
        } else if (pin_reserve(ctx, ctx->pin_stage, hb + (n_cig + 1) * 4) == PEP_OK) {
            PEP_HIP(ctx, hipMemcpyAsync(ctx->pin_stage.p, d_hits, hb + n_cig * 4, hipMemcpyDeviceToHost, st));       // hits and arena are adjacent in ws[23]
            res->st_hits = reinterpret_cast<const pep_hit *>(ctx->pin_stage.p);
            res->st_cigar = reinterpret_cast<const uint32_t *>(ctx->pin_stage.p + hb);
        } else {
            res->hits.resize(n_hits);
            res->cigar.resize(n_cig);
            PEP_HIP(ctx, hipMemcpyAsync(res->hits.data(), d_hits, hb, hipMemcpyDeviceToHost, st));
            if (n_cig) PEP_HIP(ctx, hipMemcpyAsync(res->cigar.data(), d_cig, n_cig * 4, hipMemcpyDeviceToHost, st));
        }
        res->n_hits = n_hits; res->n_cigar = n_cig;
        res->d_hits = d_hits; res->d_cigar = d_cig;
    }
    pep_timer_end(ctx, TM_TRACE);
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}
