// K14 + host pass: the consumer of the all-vs-all table, get_similar_pairs / get_similar of PEPPAN.py:194-294.
//
//   pep_similar_scan     (host C++)  the ORDER-DEPENDENT part: one pass over the sorted table with the alive / pending state of
//                                    PEPPAN.py:231-276 - which genes are absorbed or repetitive, which pairs conflict, and which groups
//                                    of forward rows have to be judged by get_similar.  Nothing in it depends on what get_similar
//                                    returns, so it runs first and hands the groups over as a list of events.
//   pep_pair_support     (K14, GPU)  get_similar (PEPPAN.py:195-224) for all those groups at once, one wavefront per group: the walk
//                                    over the CIGAR runs of the group's rows with a coverage map over the query's nucleotides (last
//                                    writer's identity per position, positions remembered in first-cover order), the decision test
//                                    after every M run and the mean identity exactly as numpy computes it (below).
//   pep_similar_resolve  (host C++)  the dictionary semantics of ortho_pairs (first writer wins for get_similar, a conflict overwrites,
//                                    insertion order kept) over the events and their values.
//
// The mean.  The reference computes int(np.mean(list(matched_aa.values())) * 10000): a float64 sum over the positions in dictionary
// (first-insertion) order by numpy's pairwise summation - blocks of at most 128 elements summed through 8 interleaved accumulators,
// combined by a fixed binary tree (n2 = n / 2 - (n / 2) % 8) - divided by n.  A plain running sum differs from it in the last bit for
// almost every n (300 of 300 random cases), and int() truncation turns a last-bit difference at x.9999999999 into a different integer, so
// K14 reproduces the summation tree operation for operation: 8 lanes per leaf block hold the 8 accumulators, their combination is the
// same ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) + tail, and one lane folds the leaf sums along the tree.  IEEE float64 additions and one
// division give the same bits on the GPU as in numpy (tests/test_gpu_parity.py compares with np.mean itself through the oracle).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "common.h"
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <unordered_map>

namespace {

constexpr int LEAF_MAX = 512;            // leaf blocks (<= 128 elements, >= 64 once there are several) kept in LDS: n <= 32768 positions
constexpr int ROW_MAX = 255;             // rows per group (the reference never judges 50 or more, PEPPAN.py:266)
constexpr int PW_BLOCK = 128;

struct SupportArgs {
    const pep_support_row *rows;
    const uint32_t *cigar;
    const uint64_t *grp_off;
    const uint32_t *grp_qlen, *grp_rlen;
    const uint64_t *scr_off;             // per group: first entry of its scratch (qlen + 2 entries)
    uint8_t *last_row;                   // scratch: 1 + index (inside the group) of the last row that covered the position, 0 = not covered
    uint32_t *seq;                       // scratch: positions in first-cover order
    uint8_t *val_row;                    // scratch: last_row of seq[k]
    int32_t *value;
    uint64_t n_groups;
    uint64_t g_base;                     // first group of this launch (the groups go out in batches bounded by a scratch budget)
    pep_support_limits lim;
};

__device__ __forceinline__ int split_left(int n) { const int h = n / 2; return h - h % 8; }

// numpy's pairwise sum of v[0 .. n) where v(k) = iden[val_row[k] - 1]; every lane returns the sum
__device__ double numpy_sum(const uint8_t *val_row, const double *iden, int n, int lane, uint32_t *leaf_off, uint32_t *leaf_n, double *leaf_sum)
{
    auto v = [&](int k) { return iden[val_row[k] - 1]; };
    auto leaf_serial = [&](int off, int m) {
        if (m < 8) { double r = 0.; for (int i = 0; i < m; ++i) r += v(off + i); return r; }
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = v(off + j);
        int i = 8;
        for (; i < m - (m % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += v(off + i + j);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < m; ++i) res += v(off + i);
        return res;
    };
    // phase A (lane 0): the leaves of the summation tree, left to right
    int n_leaf = 0;
    if (lane == 0) {
        int st_off[24], st_n[24], sp = 0;
        st_off[0] = 0; st_n[0] = n; sp = 1;
        while (sp > 0) {
            --sp;
            const int off = st_off[sp], m = st_n[sp];
            if (m <= PW_BLOCK) { if (n_leaf < LEAF_MAX) { leaf_off[n_leaf] = (uint32_t)off; leaf_n[n_leaf] = (uint32_t)m; } ++n_leaf; }
            else { const int l = split_left(m); st_off[sp] = off + l; st_n[sp] = m - l; ++sp; st_off[sp] = off; st_n[sp] = l; ++sp; }
        }
    }
    n_leaf = __shfl(n_leaf, 0, 64);
    const bool in_lds = n_leaf <= LEAF_MAX;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // phase B: leaf sums, 8 lanes per leaf = its 8 accumulators
    if (in_lds) {
        const int j = lane & 7, w = lane >> 3;
        for (int l0 = 0; l0 < n_leaf; l0 += 8) {                // wave-uniform trip count
            const int l = l0 + w;
            const bool have = l < n_leaf;
            const int off = have ? (int)leaf_off[l] : 0, m = have ? (int)leaf_n[l] : 0;
            double r = 0.;
            if (m >= 8) {
                r = v(off + j);
                for (int i = 8; i < m - (m % 8); i += 8) r += v(off + i + j);
            }
            const double t = r + __shfl_down(r, 1, 64);
            const double u = t + __shfl_down(t, 2, 64);
            double res = u + __shfl_down(u, 4, 64);
            if (j == 0 && have) {
                if (m < 8) { res = 0.; for (int i = 0; i < m; ++i) res += v(off + i); }
                else for (int i = m - (m % 8); i < m; ++i) res += v(off + i);
                leaf_sum[l] = res;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // phase C (lane 0): fold the leaf sums along the tree (post-order, explicit stack; leaves are met left to right)
    double total = 0.;
    if (lane == 0) {
        int fr_off[24], fr_n[24], fr_stage[24], sp = 0, li = 0;
        double fr_left[24];
        fr_off[0] = 0; fr_n[0] = n; fr_stage[0] = 0; sp = 1;
        bool have_val = false;
        double val = 0.;
        while (sp > 0) {
            const int top = sp - 1;
            if (!have_val) {
                if (fr_n[top] <= PW_BLOCK) { val = in_lds ? leaf_sum[li] : leaf_serial(fr_off[top], fr_n[top]); ++li; have_val = true; --sp; }
                else { fr_stage[top] = 1; const int l = split_left(fr_n[top]); fr_off[sp] = fr_off[top]; fr_n[sp] = l; fr_stage[sp] = 0; ++sp; }
            } else if (fr_stage[top] == 1) {
                fr_left[top] = val; fr_stage[top] = 2; have_val = false;
                const int l = split_left(fr_n[top]);
                fr_off[sp] = fr_off[top] + l; fr_n[sp] = fr_n[top] - l; fr_stage[sp] = 0; ++sp;
            } else { val = fr_left[top] + val; --sp; }
        }
        total = val;
    }
    return __shfl(total, 0, 64);
}

__global__ __launch_bounds__(64) void k14_pair_support(SupportArgs a)
{
    __shared__ uint32_t leaf_off[LEAF_MAX], leaf_n[LEAF_MAX];
    __shared__ double leaf_sum[LEAF_MAX];
    __shared__ double iden[ROW_MAX + 1];
    const int lane = threadIdx.x;
    const uint64_t g = a.g_base + blockIdx.x;
    if (g >= a.n_groups) return;
    const uint64_t r0 = a.grp_off[g], r1 = a.grp_off[g + 1];
    const int ql = (int)a.grp_qlen[g], sl = (int)a.grp_rlen[g];
    const int n_rows = (int)(r1 - r0);
    int result = PEP_SUPPORT_NONE;
    bool decided = n_rows == 0 || 20ll * min(ql, sl) <= (long long)max(ql, sl);          // PEPPAN.py:199-200
    uint8_t *last_row = a.last_row + a.scr_off[g];
    uint32_t *seq = a.seq + a.scr_off[g];
    uint8_t *val_row = a.val_row + a.scr_off[g];
    if (!decided) {
        for (int p = lane; p <= ql + 1; p += 64) last_row[p] = 0;
        for (int x = lane; x < n_rows; x += 64) iden[x] = a.rows[r0 + x].identity;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const double min_len = fmin(a.lim.match_len[0], fmin(a.lim.match_len[1], a.lim.match_len[2]));
    const double min_prop = fmin(a.lim.match_prop[0], fmin(a.lim.match_prop[1], a.lim.match_prop[2]));
    int n_cov = 0;
    for (int x = 0; x < n_rows && !decided; ++x) {
        const pep_support_row row = a.rows[r0 + x];
        long long qpos = row.q_start, rpos = row.r_start;
        const uint32_t *cg = a.cigar + row.cigar_off;
        for (uint32_t k = 0; k < row.cigar_runs && !decided; ++k) {
            const uint32_t run = cg[k];
            const long long len = run >> 2;
            const uint32_t op = run & 3u;
            if (op == 1) { qpos += len; continue; }
            if (op != 0) { rpos += len; continue; }
            // an in-frame M run covers the query positions from the first codon start inside it (PEPPAN.py:206-208)
            const long long lo = qpos + ((1 - qpos) % 3 + 3) % 3, hi = min(qpos + len, (long long)ql + 1);
            if (lo < hi && (a.lim.any_frame || qpos % 3 == rpos % 3)) {
                for (long long p0 = lo; p0 < hi; p0 += 64) {                             // wave-uniform trip count
                    const long long p = p0 + lane;
                    const bool in = p < hi;
                    const bool fresh = in && last_row[p] == 0;
                    const unsigned long long fm = __ballot(fresh);
                    if (fresh) seq[n_cov + __popcll(fm & ((1ull << lane) - 1ull))] = (uint32_t)p;
                    if (in) last_row[p] = (uint8_t)(x + 1);
                    n_cov += __popcll(fm);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            qpos += len; rpos += len;
            const double n_nt = (double)(3ll * n_cov);
            if (n_cov > 0 && n_nt >= min_len && n_nt >= min_prop * (double)ql) {                       // PEPPAN.py:211
                for (int i = lane; i < n_cov; i += 64) val_row[i] = last_row[seq[i]];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const double sum = numpy_sum(val_row, iden, n_cov, lane, leaf_off, leaf_n, leaf_sum);
                const double mean = sum / (double)n_cov;
                const int ave = (int)(mean * 10000.0);                                    // PEPPAN.py:212
                if ((double)ave >= a.lim.identity_x1e4) {
                    const double shorter = (double)min(ql, sl);
                    const double need = fmin(fmax(a.lim.match_len[0], a.lim.match_prop[0] * shorter),
                                             fmin(fmax(a.lim.match_len[1], a.lim.match_prop[1] * shorter), fmax(a.lim.match_len[2], a.lim.match_prop[2] * shorter)));
                    result = n_nt >= need ? ave : 0;                                      // PEPPAN.py:214-219
                    decided = true;
                }
            }
        }
    }
    if (lane == 0) a.value[g] = result;
}

}  // namespace

int pep_k14_pair_support(pep_ctx *ctx, uint64_t n_rows, const pep_support_row *h_rows, const uint32_t *h_cigar, uint64_t n_cigar, uint64_t n_groups,
                         const uint64_t *h_grp_off, const uint32_t *h_qlen, const uint32_t *h_rlen, const pep_support_limits *lim, int32_t *h_value)
{
    if (n_groups == 0) return PEP_OK;
    if (n_groups > 0x7FFFFFFFull) return pep_fail(ctx, PEP_ERR_LIMIT, "pep_pair_support: more than 2^31 - 1 groups");
    if (h_grp_off[0] != 0 || h_grp_off[n_groups] != n_rows) return pep_fail(ctx, PEP_ERR_ARG, "pep_pair_support: grp_off must run from 0 to n_rows");
    // scratch (query length + 2 entries of 6 bytes) only for the groups the kernel works on - those with rows whose two genes pass the length test of
    // PEPPAN.py:199-200 - and in batches of at most SCR_BUDGET entries: a self search with millions of settled pairs of kb-long genes would otherwise
    // ask for tens of GB in one reservation.  scr[g] = offset of group g inside ITS batch; a batch is a run of consecutive groups.
    constexpr uint64_t SCR_BUDGET = 160ull << 20;                     // entries: 1 GiB of scratch
    std::vector<uint64_t> scr(n_groups + 1, 0), batch_end;
    uint64_t used = 0, cap = 0;
    for (uint64_t g = 0; g < n_groups; ++g) {
        if (h_grp_off[g + 1] < h_grp_off[g]) return pep_fail(ctx, PEP_ERR_ARG, "pep_pair_support: grp_off must be non-decreasing");
        if (h_grp_off[g + 1] - h_grp_off[g] > (uint64_t)ROW_MAX) return pep_fail(ctx, PEP_ERR_LIMIT, "pep_pair_support: more than 255 alignments in one group");
        const uint64_t ql = h_qlen[g], sl = h_rlen[g];
        const bool works = h_grp_off[g + 1] > h_grp_off[g] && 20ull * std::min(ql, sl) > std::max(ql, sl);
        const uint64_t need = works ? ((ql + 2 + 3) & ~3ull) : 0;
        if (used && used + need > SCR_BUDGET) { batch_end.push_back(g); used = 0; }
        scr[g] = used;
        used += need;
        cap = std::max(cap, used);
        for (uint64_t x = h_grp_off[g]; x < h_grp_off[g + 1]; ++x) {
            const pep_support_row &r = h_rows[x];
            if (r.cigar_off + r.cigar_runs > n_cigar) return pep_fail(ctx, PEP_ERR_ARG, "pep_pair_support: CIGAR slice out of range");
            if (r.q_start < 1) return pep_fail(ctx, PEP_ERR_ARG, "pep_pair_support: coordinates are 1-based");
        }
    }
    batch_end.push_back(n_groups);
    const uint64_t total = cap;
    // ws[0] rows, ws[1] cigar, ws[2] groups (off u64, scr u64, qlen u32, rlen u32, value i32), ws[3] scratch bytes x2, ws[4] scratch u32
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], (n_rows + 1) * sizeof(pep_support_row)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[1], (n_cigar + 1) * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[2], (n_groups + 1) * (8 + 8 + 4 + 4 + 4) + 64));
    PEP_TRY(dev_reserve(ctx, ctx->ws[3], 2 * total + 64));
    PEP_TRY(dev_reserve(ctx, ctx->ws[4], total * 4 + 64));
    char *gb = ctx->ws[2].as<char>();
    uint64_t *d_off = reinterpret_cast<uint64_t *>(gb), *d_scr = d_off + n_groups + 1;
    uint32_t *d_ql = reinterpret_cast<uint32_t *>(d_scr + n_groups + 1), *d_rl = d_ql + n_groups + 1;
    int32_t *d_val = reinterpret_cast<int32_t *>(d_rl + n_groups + 1);
    hipStream_t st = ctx->stream;
    if (n_rows) PEP_TRY(pep_h2d(ctx, ctx->ws[0].p, h_rows, n_rows * sizeof(pep_support_row)));
    if (n_cigar) PEP_TRY(pep_h2d(ctx, ctx->ws[1].p, h_cigar, n_cigar * 4));
    PEP_TRY(pep_h2d(ctx, d_off, h_grp_off, (n_groups + 1) * 8));
    PEP_TRY(pep_h2d(ctx, d_scr, scr.data(), (n_groups + 1) * 8));
    PEP_TRY(pep_h2d(ctx, d_ql, h_qlen, n_groups * 4));
    PEP_TRY(pep_h2d(ctx, d_rl, h_rlen, n_groups * 4));
    SupportArgs a;
    a.rows = ctx->ws[0].as<const pep_support_row>(); a.cigar = ctx->ws[1].as<const uint32_t>();
    a.grp_off = d_off; a.grp_qlen = d_ql; a.grp_rlen = d_rl; a.scr_off = d_scr;
    a.last_row = ctx->ws[3].as<uint8_t>(); a.val_row = a.last_row + total; a.seq = ctx->ws[4].as<uint32_t>();
    a.value = d_val; a.n_groups = n_groups; a.lim = *lim;
    for (uint64_t b = 0, g0 = 0; b < batch_end.size(); g0 = batch_end[b++]) {             // (one launch unless the scratch budget splits the groups)
        a.g_base = g0;
        hipLaunchKernelGGL(k14_pair_support, dim3((unsigned)(batch_end[b] - g0)), dim3(64), 0, st, a);
    }
    PEP_HIP(ctx, hipGetLastError());
    PEP_TRY(pep_d2h_queue(ctx, h_value, d_val, n_groups * 4));
    PEP_HIP(ctx, hipStreamSynchronize(st));
    pep_d2h_finish(ctx);
    return PEP_OK;
}

extern "C" {

int pep_pair_support(pep_ctx *ctx, uint64_t n_rows, const pep_support_row *rows, const uint32_t *cigar, uint64_t n_cigar, uint64_t n_groups,
                     const uint64_t *grp_off, const uint32_t *grp_qlen, const uint32_t *grp_rlen, const pep_support_limits *lim, int32_t *value)
{
    if (!ctx || !lim || (n_groups && (!grp_off || !grp_qlen || !grp_rlen || !value)) || (n_rows && !rows) || (n_cigar && !cigar)) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    return pep_k14_pair_support(ctx, n_rows, rows, cigar, n_cigar, n_groups, grp_off, grp_qlen, grp_rlen, lim, value);
}

// Host side, no context.  Rows in the order of the table RunBlast.run returns (query, reference, score).  q / r: gene codes in
// [0, n_genes) that keep the order of the gene ids (the pair keys are (smaller, larger)).  action / forward: the row-local tests of
// PEPPAN.py:246-263, evaluated by the caller for all rows at once.
int pep_similar_scan(uint64_t n, const int64_t *q, const int64_t *r, const uint8_t *action, const uint8_t *forward, const int32_t *iden4, uint64_t n_genes,
                     uint8_t *alive, uint8_t *seen_as_query, int64_t *absorbed, uint64_t *n_absorbed,
                     uint8_t *ev_kind, int64_t *ev_a, int64_t *ev_b, uint64_t *ev_row_off, uint64_t *ev_rows, uint64_t *n_events)
{
    if (!n_absorbed || !n_events || (n && (!q || !r || !action || !forward || !iden4 || !absorbed || !ev_kind || !ev_a || !ev_b || !ev_row_off || !ev_rows)) ||
        (n_genes && (!alive || !seen_as_query))) return PEP_ERR_ARG;
    for (uint64_t g = 0; g < n_genes; ++g) { alive[g] = 1; seen_as_query[g] = 0; }
    uint64_t na = 0, ne = 0, nr = 0;
    std::vector<uint64_t> pending;
    if (ev_row_off) ev_row_off[0] = 0;
    auto settle = [&]() {
        const int64_t a = q[pending[0]], b = r[pending[0]];
        if (pending.size() >= 50) alive[b] = 0;                              // fifty or more hits between two genes: the reference gene is a repeat (PEPPAN.py:266-267)
        else if (a != b) {
            ev_kind[ne] = PEP_EVENT_SUPPORT; ev_a[ne] = std::min(a, b); ev_b[ne] = std::max(a, b);
            for (uint64_t k : pending) ev_rows[nr++] = k;
            ev_row_off[++ne] = nr;
        }
        pending.clear();
    };
    for (uint64_t k = 0; k < n; ++k) {
        const int64_t a = q[k], b = r[k];
        if (a < 0 || b < 0 || (uint64_t)a >= n_genes || (uint64_t)b >= n_genes) return PEP_ERR_ARG;
        seen_as_query[a] = 1;
        if (!alive[a] || !alive[b]) continue;                                // PEPPAN.py:237-243
        switch (action[k]) {
            case PEP_ROW_CONFLICT:
                ev_kind[ne] = PEP_EVENT_CONFLICT; ev_a[ne] = std::min(a, b); ev_b[ne] = std::max(a, b);
                ev_row_off[++ne] = nr;
                break;
            case PEP_ROW_ABSORB_QUERY:
                absorbed[3 * na] = b; absorbed[3 * na + 1] = a; absorbed[3 * na + 2] = iden4[k]; ++na;
                alive[a] = 0;
                break;
            case PEP_ROW_ABSORB_REF:
                absorbed[3 * na] = a; absorbed[3 * na + 1] = b; absorbed[3 * na + 2] = iden4[k]; ++na;
                alive[b] = 0;
                break;
            default:
                if (!forward[k]) break;                                      // PEPPAN.py:262-263
                if (!pending.empty() && (q[pending[0]] != a || r[pending[0]] != b)) settle();
                pending.push_back(k);
        }
    }
    if (!pending.empty()) settle();
    *n_absorbed = na; *n_events = ne;
    return PEP_OK;
}

// ortho_pairs as the reference's dictionary builds it: a conflict sets -2 whatever was there (PEPPAN.py:249), get_similar writes only
// pairs that have no entry yet and only when it reaches a decision (PEPPAN.py:196-198, 216-219); entries keep the position of their first
// insertion; zero values are left out at the end (PEPPAN.py:294).  out: (a, b, value) triples.
int pep_similar_resolve(uint64_t n_events, const uint8_t *ev_kind, const int64_t *ev_a, const int64_t *ev_b, const int32_t *ev_value,
                        int64_t *out, uint64_t *n_out)
{
    if (!n_out || (n_events && (!ev_kind || !ev_a || !ev_b || !ev_value || !out))) return PEP_ERR_ARG;
    struct KeyHash { size_t operator()(const std::pair<int64_t, int64_t> &k) const { return (size_t)((uint64_t)k.first * 0x9E3779B97F4A7C15ull ^ (uint64_t)k.second); } };
    std::unordered_map<std::pair<int64_t, int64_t>, size_t, KeyHash> at;
    std::vector<int64_t> ent;                       // (a, b, value) in first-insertion order
    at.reserve((size_t)n_events);
    for (uint64_t e = 0; e < n_events; ++e) {
        const std::pair<int64_t, int64_t> key(ev_a[e], ev_b[e]);
        auto it = at.find(key);
        if (ev_kind[e] == PEP_EVENT_CONFLICT) {
            if (it != at.end()) ent[3 * it->second + 2] = -2;
            else { at.emplace(key, ent.size() / 3); ent.push_back(key.first); ent.push_back(key.second); ent.push_back(-2); }
        } else if (it == at.end() && ev_value[e] != PEP_SUPPORT_NONE) {
            at.emplace(key, ent.size() / 3); ent.push_back(key.first); ent.push_back(key.second); ent.push_back(ev_value[e]);
        }
    }
    uint64_t no = 0;
    for (size_t x = 0; x < ent.size(); x += 3)
        if (ent[x + 2] != 0) { out[3 * no] = ent[x]; out[3 * no + 1] = ent[x + 1]; out[3 * no + 2] = ent[x + 2]; ++no; }
    *n_out = no;
    return PEP_OK;
}

// The clusterer's input (K9, clust.py:62-66 hands a FASTA file to mmseqs): the sequences of FASTA text as one code array + offsets, in ONE pass
// over the bytes - as Python (a split, a join and an upper() per record, then a table look-up over the joined text) this was a third of
// iterClust's time at 300 k genes.  Rules as peppan_amd/clust.py readFasta: a record starts at a '>' at the start of a line; its first line is
// the header; of the other lines those starting with '#' are dropped, blanks (str.split()'s ASCII set) are removed, every other byte goes
// through `table` (256 entries; case is the table's business).  Text before the first header belongs to nobody.
// off[0 .. *n_records] = start of each record's codes; PEP_ERR_LIMIT when there are more than `cap` records (nothing useful written).
static int fasta_scan_impl(const uint8_t *data, uint64_t n, const uint8_t *table, uint8_t *codes, uint64_t *off, uint64_t *name_off, uint32_t *name_len, uint64_t cap,
                           uint64_t *n_records, int32_t *non_ascii)
{
    if ((n && !data) || !table || !off || !n_records || (n && !codes)) return PEP_ERR_ARG;
    bool blank[256] = {};
    for (int c : {9, 10, 11, 12, 13, 28, 29, 30, 31, 32}) blank[c] = true;
    uint16_t wide[256];
    for (int c = 0; c < 256; ++c) wide[c] = (uint16_t)(table[c] | (blank[c] ? 0x100 : 0) | (c & 0x80 ? 0x200 : 0));
    uint64_t i = 0, w = 0, r = 0;
    int32_t high = 0;
    bool upper_is_identity = true;
    for (int c = 'A'; c <= 'Z'; ++c) upper_is_identity = upper_is_identity && table[c] == c;
    if (n && data[0] != '>') {
        i = n;
        for (uint64_t p = 0; p + 1 < n;) {
            const void *nl = memchr(data + p, '\n', n - 1 - p);
            if (!nl) break;
            p = (uint64_t)((const uint8_t *)nl - data) + 1;
            if (data[p] == '>') { i = p; break; }
        }
    }
    while (i < n) {                               // data[i] is the '>' of a header line
        if (r >= cap) return PEP_ERR_LIMIT;
        const void *nl = memchr(data + i, '\n', n - i);
        const uint64_t hdr_end = nl ? (uint64_t)((const uint8_t *)nl - data) : n;
        if (name_off) {                           // the record's name: the first blank-delimited token behind the '>'
            uint64_t a = i + 1;
            while (a < hdr_end && blank[data[a]]) ++a;
            uint64_t b = a;
            while (b < hdr_end && !blank[data[b]]) ++b;
            name_off[r] = a;
            name_len[r] = (uint32_t)std::min<uint64_t>(b - a, 0xffffffffu);
        }
        off[r++] = w;
        if (!nl) break;
        i = hdr_end + 1;
        while (i < n && data[i] != '>') {
            const void *e = memchr(data + i, '\n', n - i);
            const uint64_t end = e ? (uint64_t)((const uint8_t *)e - data) : n;
            if (data[i] != '#') {
                uint8_t *out = codes + w;                       // the whole line through the table first (independent bytes), blanks squeezed out only if there are any
                const uint8_t *in = data + i;
                const uint64_t len = end - i;
                // a line of upper-case letters that the table leaves alone - every line of the files PEPPAN writes for its own searches - is one copy
                // (the test is a reduction the compiler vectorises: 10 MB of exemplars in 1 ms instead of 5 through the look-up loop)
                if (upper_is_identity) {
                    uint8_t odd = 0;
                    for (uint64_t j = 0; j < len; ++j) odd |= (uint8_t)((uint8_t)(in[j] - 'A') > 25u);
                    if (!odd) { memcpy(out, in, len); w += len; i = e ? end + 1 : n; continue; }
                }
                uint32_t acc = 0;                               // (one look-up per byte: code | blank << 8 | high bit << 9)
                for (uint64_t j = 0; j < len; ++j) {
                    const uint32_t t = wide[in[j]];
                    acc |= t;
                    out[j] = (uint8_t)t;
                }
                high |= (acc >> 2) & 0x80;
                if (!(acc & 0x100)) w += len;
                else {
                    uint64_t k = 0;
                    for (uint64_t j = 0; j < len; ++j) {
                        out[k] = out[j];
                        k += blank[in[j]] ? 0 : 1;
                    }
                    w += k;
                }
            }
            i = e ? end + 1 : n;
        }
    }
    off[r] = w;
    *n_records = r;
    if (non_ascii) *non_ascii = high ? 1 : 0;
    return PEP_OK;
}

int pep_fasta_scan(const uint8_t *data, uint64_t n, const uint8_t *table, uint8_t *codes, uint64_t *off, uint64_t cap, uint64_t *n_records, int32_t *non_ascii)
{
    return fasta_scan_impl(data, n, table, codes, off, nullptr, nullptr, cap, n_records, non_ascii);
}

// The row-local tests in front of pep_similar_scan (PEPPAN.py:244-263; the numpy form - fifteen passes over 70 000 rows - was 1.5 of the 4 ms the
// classification step of get_similar_pairs took).
int pep_similar_classify(uint64_t n, const int64_t *q, const int64_t *r, const double *iden, const int64_t *qs, const int64_t *qe, const int64_t *ss,
                         const int64_t *se, const int64_t *ql, const int64_t *sl, const uint8_t *rank_ge, const uint8_t *rank_le, double near_identity,
                         double cover, uint8_t *action, uint8_t *forward, int32_t *iden4)
{
    if (n && (!q || !r || !iden || !qs || !qe || !ss || !se || !ql || !sl || !rank_ge || !rank_le || !action || !forward || !iden4)) return PEP_ERR_ARG;
    const double root = std::sqrt(cover);
    for (uint64_t k = 0; k < n; ++k) {
        const double fqs = (double)qs[k], fqe = (double)qe[k], fss = (double)ss[k], fse = (double)se[k], fql = (double)ql[k], fsl = (double)sl[k];
        const double q_span = fqe - fqs + 1.0, r_span = std::fabs(fse - fss) + 1.0;
        const bool near = q[k] != r[k] && iden[k] >= near_identity;
        auto mod3 = [](int64_t a) { const int64_t m = a % 3; return m < 0 ? m + 3 : m; };          // numpy's % (sign of the divisor) - on the integers themselves: fmod of their doubles was most of this pass
        const bool same_head = mod3(qs[k]) == mod3(ss[k]), same_tail = mod3(ql[k] - qe[k]) == mod3(sl[k] - se[k]);
        const bool off_frame = fss > fse || (!same_head && same_tail);
        const bool in_frame = !off_frame && fss < fse && same_head && same_tail;
        uint8_t a = PEP_ROW_ORDINARY;
        if (near && off_frame && (q_span >= cover * fql || r_span >= cover * fsl)) a = PEP_ROW_CONFLICT;
        if (near && in_frame && fql <= fsl && q_span >= root * fsl && rank_ge[k]) a = PEP_ROW_ABSORB_QUERY;
        if (near && in_frame && fql > fsl && r_span >= root * fql && rank_le[k]) a = PEP_ROW_ABSORB_REF;
        action[k] = a;
        forward[k] = fss < fse ? 1 : 0;
        iden4[k] = (int32_t)(int64_t)(iden[k] * 10000.0);
    }
    return PEP_OK;
}

// The same pass for a reader that also wants the records' names (configure.readFasta: the exemplar file of the hot call is read afresh whenever
// the step in front rewrote it - 16 ms as one split / join / upper per record for 10 000 genes): name_off[r] / name_len[r] = the first
// blank-delimited token of record r's header line inside data (length 0: a header without a name - the caller applies its own rules).
int pep_fasta_records(const uint8_t *data, uint64_t n, const uint8_t *table, uint8_t *codes, uint64_t *off, uint64_t *name_off, uint32_t *name_len, uint64_t cap,
                      uint64_t *n_records, int32_t *non_ascii)
{
    if (!name_off || !name_len) return PEP_ERR_ARG;
    return fasta_scan_impl(data, n, table, codes, off, name_off, name_len, cap, n_records, non_ascii);
}


// The last step of get_similar_pairs (PEPPAN.py:278-288): the exemplar FASTA keeps only the records of genes in `ids` (sorted ascending) -
// header line and every line behind it, byte for byte; what precedes the first header goes.  Host code, one read and (when something
// goes) one write of the file: as Python over the file's buffer this was 10 of the 17 ms the whole decision pass took.
// A record's name is the first blank-delimited token of its header line and has to be a decimal integer ([+-]digits, as PEPPAN's encoded
// gene names are); anything else -> PEP_ERR_ARG with the file untouched, and the caller applies its own (Python int()) rules.
// which records of a FASTA buffer stay: start[] = the header lines' offsets (+ the buffer's length behind the last), keep[] per record
static int fasta_keep_plan(const char *d, size_t n, const int64_t *ids, uint64_t n_ids, std::vector<size_t> &start, std::vector<char> &keep)
{
    for (size_t p = 0; p < n;) {
        if (d[p] == '>') {
            size_t x = p + 1;
            bool neg = false;
            if (x < n && (d[x] == '+' || d[x] == '-')) { neg = d[x] == '-'; ++x; }
            const size_t first_digit = x;
            int64_t v = 0;
            while (x < n && d[x] >= '0' && d[x] <= '9') {
                if (v > (INT64_MAX - 9) / 10) return PEP_ERR_ARG;
                v = v * 10 + (d[x] - '0');
                ++x;
            }
            if (x == first_digit || (x < n && d[x] != ' ' && d[x] != '\t' && d[x] != '\n' && d[x] != '\r' && d[x] != '\f' && d[x] != '\v')) return PEP_ERR_ARG;
            if (neg) v = -v;
            start.push_back(p);
            keep.push_back(std::binary_search(ids, ids + n_ids, v) ? 1 : 0);
        }
        const void *nl = memchr(d + p, '\n', n - p);
        if (!nl) break;
        p = (size_t)((const char *)nl - d) + 1;
    }
    return PEP_OK;
}

// the kept records moved together inside the buffer (runs of kept neighbours in one piece; nothing moves in front of the first record that goes) -> the new length
static size_t fasta_keep_compact(char *w, std::vector<size_t> &start, const std::vector<char> &keep, size_t n)
{
    start.push_back(n);
    size_t at = 0;
    for (size_t k = 0; k + 1 < start.size();) {
        if (!keep[k]) { ++k; continue; }
        size_t j = k;
        while (j + 1 < start.size() - 1 && keep[j + 1]) ++j;                 // records k .. j stay
        const size_t from = start[k], len = start[j + 1] - start[k];
        if (from != at) memmove(w + at, w + from, len);
        at += len;
        k = j + 1;
    }
    return at;
}

int pep_fasta_keep(const char *path, const int64_t *ids, uint64_t n_ids, uint64_t *n_records, uint64_t *n_kept)
{
    if (!path || (n_ids && !ids)) return PEP_ERR_ARG;
    // The file's own pages, mapped: the header lines are looked at where they lie, the kept records move together inside the mapping and the file is cut to its new
    // length - no copy of the 10 MB into a buffer of this process and none back (read + write were 2 of this function's 3.2 ms at 10 000 exemplars).
    const int fd = open(path, O_RDWR);
    if (fd < 0) return PEP_ERR_ARG;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return PEP_ERR_INTERNAL; }
    const size_t n = (size_t)st.st_size;
    std::vector<size_t> start;
    std::vector<char> keep;
    auto count = [&]() {
        uint64_t kept = 0;
        for (char k : keep) kept += k;
        if (n_records) *n_records = start.size();
        if (n_kept) *n_kept = kept;
        return kept == start.size() && (start.empty() || start[0] == 0);       // every record stays: the file is left alone
    };
    void *map = n ? mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0) : MAP_FAILED;
    if (n && map != MAP_FAILED) {
        char *d = static_cast<char *>(map);
        int rc = fasta_keep_plan(d, n, ids, n_ids, start, keep);
        size_t at = n;
        const bool untouched = rc == PEP_OK && count();
        if (rc == PEP_OK && !untouched) at = fasta_keep_compact(d, start, keep, n);
        if (munmap(map, n) != 0) rc = rc == PEP_OK ? PEP_ERR_INTERNAL : rc;
        if (rc == PEP_OK && !untouched && ftruncate(fd, (off_t)at) != 0) rc = PEP_ERR_INTERNAL;
        if (close(fd) != 0 && rc == PEP_OK) rc = PEP_ERR_INTERNAL;
        return rc;
    }
    close(fd);
    // (an empty file, or one that cannot be mapped: read, compact, write)
    FILE *f = fopen(path, "rb");
    if (!f) return PEP_ERR_ARG;
    std::vector<char> data(n);
    const size_t got = data.empty() ? 0 : fread(data.data(), 1, data.size(), f);
    fclose(f);
    if (got != data.size()) return PEP_ERR_INTERNAL;
    const int rc = fasta_keep_plan(data.data(), n, ids, n_ids, start, keep);
    if (rc != PEP_OK) return rc;
    if (count()) return PEP_OK;
    const size_t at = fasta_keep_compact(data.data(), start, keep, n);
    f = fopen(path, "wb");
    if (!f) return PEP_ERR_INTERNAL;
    const size_t put = at ? fwrite(data.data(), 1, at, f) : 0;
    if (fclose(f) != 0 || put != at) return PEP_ERR_INTERNAL;
    return PEP_OK;
}

}  // extern "C"
