// K5: banded affine Smith-Waterman, anti-diagonal sweep by one wavefront (64 lanes) - over ONE candidate in the 32-bit kernels,
// over TWO candidates at once in the packed 16-bit kernels (score pass and traceback pass) that run whenever the scores fit.
//
// Mapping.  A candidate's band holds 128 diagonals d = j - i in [dlo, dlo+127].  Lane l owns the two
// adjacent diagonals A = dlo+2l and B = dlo+2l+1.  Cells of one anti-diagonal s = i+j all have the parity
// of s, so even steps update every lane's A cell and odd steps every lane's B cell: all 64 lanes work on
// every step, and the three DP dependencies are
//     diagonal  (i-1,j-1): same diagonal, two steps ago      -> the lane's own register
//     left      (i,  j-1): diagonal d-1, previous step        -> own B->A... i.e. own other register, or lane l-1 (DPP wave_shr:1)
//     up        (i-1,j  ): diagonal d+1, previous step        -> own other register, or lane l+1 (DPP wave_shl:1)
// so a step costs two DPP moves and no LDS traffic for the recurrences.  Residues of the pair are staged
// in LDS (coalesced global reads once per candidate); the 32x32 substitution table is held in PEP_TAB_REP copies
// (dword w*PEP_TAB_REP + lane mod PEP_TAB_REP) so that the per-lane gather meets at most 2-way bank conflicts.
//
// Per cell a 4-bit traceback code is produced (bits0-1 source of H: 0 none/1 diagonal/2 E/3 F, bit2 E was an
// extension, bit3 F was an extension); a lane packs 8 consecutive cells of EACH of its two diagonals in one dword
// and the wave stores 512 B (one uint2 per lane) per 16 steps, fully coalesced, into the direction workspace read
// back by the walk kernel (trace.hip), which can then follow a diagonal run inside one word.
//
// Integer DP in int32.  Algorithmic HBM bytes per candidate: Lq + Lt residues + 16 B result
// (+ 32 B per anti-diagonal step of traceback codes).  VALU-bound by construction.
#include "common.h"
#include "lookback.h"

namespace {

constexpr int LDS_TABLE_BYTES = 256 * PEP_TAB_REP * 4;   // 256 dwords x PEP_TAB_REP copies
constexpr int TAB_ROW_SHIFT = PEP_TAB_REP == 32 ? 7 : 6; // log2(bytes per table row of 4 scores)
constexpr int WAVES_PER_BLOCK = 8;

constexpr int LEN_BUCKETS = 1024;        // sw_order: candidates are ordered by min(16-step blocks, LEN_BUCKETS - 1)
__device__ __forceinline__ uint32_t len_bucket(uint32_t nb) { return nb < (uint32_t)LEN_BUCKETS ? nb : (uint32_t)LEN_BUCKETS - 1; }

struct SwArgs {
    const uint64_t *cands;
    uint64_t n;
    const uint8_t *q_res, *t_res;
    const uint32_t *q_off, *q_len, *t_off, *t_len;
    const uint32_t *sub_image;     // 8192 dwords
    const uint64_t *dir_off;       // per candidate, in 256-byte blocks
    const uint32_t *nblk;          // per candidate: 16-step blocks (8 A/B step pairs)
    uint32_t *dirs;
    int4 *out;                     // score, iend, jend, a0
    int oe, ext;
    int lds_res_bytes;             // per-wave residue staging capacity (0 = global path only)
    uint32_t *defer;               // traceback pass: the pairs that left their sub-band (counts[6] of them; counts[7]: taken by sw_trace_retry_kernel)
    int pk16;                      // sweep two candidates per wavefront in packed 16-bit (score pass; traceback pass when `known` is set)
    const int32_t *known;          // traceback pass: the score of every candidate, from the score pass
    int max_sub;                   // largest table entry: bounds the scores of a pair for the 16-bit passes
    const uint32_t *order;         // candidates by decreasing length (sw_order): item w of a launch is order[w] (order[2w], order[2w+1] packed)
    const int32_t *end_lane;       // traceback pass: per candidate the lowest lane (diagonal pair) of the score pass that reached the score
    int32_t *mode;                 // traceback pass, out: first lane L0 of the 64-diagonal sub-band the codes were written for, -1 = the full band
    int split_long;                // the pairs whose windows do not fit the staging area have a launch of their own (sw_*_kernel<true>: packed sweep in chunks of blocks)
    unsigned long long *counts;    // traceback pass: the pass's counter block, filled by sw_prep - [3] candidates too long for the four-candidate sweep (a prefix of
                                   // the order), [5] candidates in the order; [2] and [4] are the work queues of the two parts (the resident wavefronts pull their items from them)
};

// The traceback of a band is taken in the 64-diagonal sub-band around the lane in which the score pass met the band's score (lanes
// [L0, L0 + 32) of the 64) whenever that sub-band alone reaches the score, in the full band otherwise (rule and reasons: band_align in
// oracle/align_oracle.c).  Half the lanes, half the cells, half the traceback codes for nearly every pair.
constexpr int SUB_LANES = 32;
__device__ __forceinline__ int sub_band_first_lane(int end_lane) { return min(max(end_lane - SUB_LANES / 2, 0), 64 - SUB_LANES); }

// anti-diagonal geometry of the band [dlo, dlo + width) of an Lq x Lt matrix: a0 (the band row of step 0) and the number of 16-step blocks
__device__ __forceinline__ void band_geom(int Lq, int Lt, int dlo, int width, int &a0, int &nblk)
{
    const int dl = max(dlo, -(Lq - 1)), dh = min(dlo + width - 1, Lt - 1);
    const int s_lo = (dl <= 0 && dh >= 0) ? 0 : (dl > 0 ? dl : -dh);
    const int s0 = s_lo - ((s_lo - dlo) & 1);
    a0 = (s0 - dlo) / 2;
    nblk = 0;
    if (dl <= dh) {
        const int dstar = min(max(Lt - Lq, dl), dh);
        const int s_hi = 2 * min(Lq - 1, Lt - 1 - dstar) + dstar;
        nblk = (s_hi - s0 + 1 + 15) / 16;
    }
}

__device__ __forceinline__ int shr1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }   // lane l <- lane l-1
__device__ __forceinline__ int shl1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }   // lane l <- lane l+1
// band-edge lanes read 0: for H that is the local-alignment boundary; for E/F it stands in for -inf, which is
// equivalent because a gap state <= 0 can never be selected over H >= 0 and only decays further
__device__ __forceinline__ int shr1z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int shl1z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }

// LDS images of the residues, ready to be added into a table address: q -> qc * 8 rows, t -> (tc >> 2) rows + (tc & 3)  (row = PEP_TAB_REP dwords)
__device__ __forceinline__ uint16_t q_addr_part(int qc) { return (uint16_t)(qc << (TAB_ROW_SHIFT + 3)); }
__device__ __forceinline__ uint16_t t_addr_part(int tc) { return (uint16_t)(((tc >> 2) << TAB_ROW_SHIFT) | (tc & 3)); }

// sub < 0: the full band (64 lanes x 2 diagonals).  sub >= 0: the sub-band of lanes [sub, sub + 32) only - lanes 32..63 then mirror lanes
// 0..31 (same cells, same values), so that the code path stays one; only lanes < 32 store.  Returns the band's best score (wave-uniform).
template <bool LDS_RES, bool TRACE>
__device__ __forceinline__ int sw_one(const SwArgs &a, uint64_t c, const unsigned char *lds_tab, uint16_t *lds_res, int lane, int sub = -1)
{
    const uint64_t key = a.cands[c];
    const uint32_t q = (uint32_t)(key >> 43), t = (uint32_t)((key >> 18) & ((1u << 25) - 1));
    const int bin = (int)(key & ((1u << 18) - 1));
    const int W = sub >= 0 ? SUB_LANES : 64;                   // lanes that own diagonals
    const int dlo = bin * 64 - (1 << 23) - 32 + (sub >= 0 ? 2 * sub : 0);
    const int ln = lane & (W - 1);
    const int Lq = (int)a.q_len[q], Lt = (int)a.t_len[t];
    const uint8_t *qg = a.q_res + a.q_off[q], *tg = a.t_res + a.t_off[t];
    int a0, nblk;
    band_geom(Lq, Lt, dlo, 2 * W, a0, nblk);
    if (sub < 0) nblk = (int)a.nblk[c];
    uint2 *dir = TRACE ? reinterpret_cast<uint2 *>(a.dirs) + a.dir_off[c] * 64 : nullptr;

    // residues used by step pair m:  A: (i, j) = (a0 + m - ln, a0 + dlo + m + ln),  B: (i, j + 1)
    int i = a0 - ln, j = a0 + dlo + ln;
    uint16_t *lq = nullptr, *lt = nullptr;
    int qlo = 0, tlo = 0;
    if (LDS_RES) {
        // stage the residue windows the sweep can touch: i in [a0-W+1, a0+8*nblk], j in [a0+dlo, a0+dlo+8*nblk+W]
        qlo = a0 - W; tlo = a0 + dlo - 1;
        const int qn = 8 * nblk + W + 8, tn = 8 * nblk + W + 8;
        lq = lds_res; lt = lds_res + ((qn + 7) & ~7);
        for (int x = lane; x < qn; x += 64) { const int g = qlo + x; lq[x] = q_addr_part(((unsigned)g < (unsigned)Lq) ? qg[g] : PEP_PAD_CODE); }
        for (int x = lane; x < tn; x += 64) { const int g = tlo + x; lt[x] = t_addr_part(((unsigned)g < (unsigned)Lt) ? tg[g] : PEP_PAD_CODE); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // volatile: keeps hipcc from fusing neighbouring 2-byte reads into ds_read_b64 at 2-byte alignment, which the LDS
    // replays at 64 cycles per instruction (SQ_LDS_UNALIGNED_STALL; cdna guide G17)
    typedef const volatile __attribute__((address_space(3))) uint16_t lds_cu16;
    lds_cu16 *vq = (lds_cu16 *)lq, *vt = (lds_cu16 *)lt;
    auto Qat = [&](int ii) -> int { return LDS_RES ? (int)vq[ii - qlo] : (int)q_addr_part(((unsigned)ii < (unsigned)Lq) ? (int)qg[ii] : PEP_PAD_CODE); };
    auto Tat = [&](int jj) -> int { return LDS_RES ? (int)vt[jj - tlo] : (int)t_addr_part(((unsigned)jj < (unsigned)Lt) ? (int)tg[jj] : PEP_PAD_CODE); };
    // gather: byte ((q*8 + t/4)*PEP_TAB_REP + copy)*4 + (t&3) of the replicated table, read as a signed byte (copy = lane mod PEP_TAB_REP)
    const signed char *tab = reinterpret_cast<const signed char *>(lds_tab) + (lane & (PEP_TAB_REP - 1)) * 4;
    // sub-band: the lanes at its two edges must read the band boundary (0), not the neighbouring mirror lane
    const int keep_l = (sub >= 0 && ln == 0) ? 0 : -1, keep_r = (sub >= 0 && ln == W - 1) ? 0 : -1;

    int HA = 0, EA = 0, FA = 0, HB = 0, EB = 0, FB = 0;
    int best = 0, best_k = -1;
    const int oe = a.oe, ext = a.ext;
    int tv = Tat(j), qv = 0;
    int k = 0;
    for (int b = 0; b < nblk; ++b) {
        uint32_t accA = 0, accB = 0;       // 8 consecutive cells of diagonal A / of diagonal B, 4 bits each
#pragma unroll 1
        for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // ---- A step: cell (i, j) on diagonal dlo + 2*ln
            qv = Qat(i);
            {
                const int sub_ = tab[qv + tv];
                const int hl = shr1z(HB) & keep_l, el = shr1z(EB) & keep_l;
                const int e_ext = el - ext, e_open = hl - oe;
                const int f_ext = FB - ext, f_open = HB - oe;
                const int E = max(e_ext, e_open), F = max(f_ext, f_open);
                const int h = HA + sub_;
                const int H = max(max(max(h, E), F), 0);
                if (TRACE) {
                    const uint32_t src = (H == 0) ? 0u : (H == h) ? 1u : (H == E) ? 2u : 3u;
                    const uint32_t nib = src | (e_ext > e_open ? 4u : 0u) | (f_ext > f_open ? 8u : 0u);
                    accA = (accA >> 4) | (nib << 28);
                    if (H > best) { best = H; best_k = k; }
                } else best = max(best, H);
                HA = H; EA = E; FA = F;
            }
            ++k; ++j;
            // ---- B step: cell (i, j) on diagonal dlo + 2*ln + 1   (j already advanced)
            tv = Tat(j);
            {
                const int sub_ = tab[qv + tv];
                const int hu = shl1z(HA) & keep_r, fu = shl1z(FA) & keep_r;
                const int e_ext = EA - ext, e_open = HA - oe;
                const int f_ext = fu - ext, f_open = hu - oe;
                const int E = max(e_ext, e_open), F = max(f_ext, f_open);
                const int h = HB + sub_;
                const int H = max(max(max(h, E), F), 0);
                if (TRACE) {
                    const uint32_t src = (H == 0) ? 0u : (H == h) ? 1u : (H == E) ? 2u : 3u;
                    const uint32_t nib = src | (e_ext > e_open ? 4u : 0u) | (f_ext > f_open ? 8u : 0u);
                    accB = (accB >> 4) | (nib << 28);
                    if (H > best) { best = H; best_k = k; }
                } else best = max(best, H);
                HB = H; EB = E; FB = F;
            }
            ++k; ++i;
        }
        if (TRACE && lane < W) dir[(size_t)b * W + ln] = make_uint2(accA, accB);
    }
    if (!TRACE) {
        // score pass: the maximum and the lowest lane that holds it (the end cell comes from the traceback pass of the selected pairs)
        const int mine = best;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
        const unsigned long long at = __ballot(mine == best);
        if (lane == 0) a.out[c] = make_int4(best, (int)__builtin_ctzll(at), -1, a0);
        return best;
    }
    // the lane's best cell: earliest step with the lane maximum == (min i, then min j) among its two diagonals
    int bi = 0x7fffffff, bj = 0x7fffffff;
    if (best_k >= 0) {
        const int m = best_k >> 1;
        bi = a0 + m - ln;
        bj = a0 + dlo + m + ln + (best_k & 1);
    }
    // wave reduction: max score, then min i, then min j
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int os = __shfl_xor(best, d, 64), oi = __shfl_xor(bi, d, 64), oj = __shfl_xor(bj, d, 64);
        const bool take = (os > best) || (os == best && (oi < bi || (oi == bi && oj < bj)));
        if (take) { best = os; bi = oi; bj = oj; }
    }
    if (lane == 0) a.out[c] = make_int4(best, best > 0 ? bi : -1, best > 0 ? bj : -1, a0);
    return best;
}

// traceback of one candidate with the 32-bit sweep: sub-band first, the full band when the sub-band cannot reach the known score
template <bool LDS_RES>
__device__ __forceinline__ void trace_one(const SwArgs &a, uint64_t c, const unsigned char *lds_tab, uint16_t *lds_res, int lane)
{
    int mode = -1;
    if (a.known && a.end_lane && a.known[c] > 0) {
        const int L0 = sub_band_first_lane(a.end_lane[c]);
        if (sw_one<LDS_RES, true>(a, c, lds_tab, lds_res, lane, L0) == a.known[c]) mode = L0;
    }
    if (mode < 0) sw_one<LDS_RES, true>(a, c, lds_tab, lds_res, lane, -1);
    if (lane == 0 && a.mode) a.mode[c] = mode;
}

// ---- packed 16-bit score pass: ONE wavefront sweeps TWO candidates, candidate 0 in the low and candidate 1 in the
// high half of every register (v_pk_add/sub/max_i16), so the recurrences cost half the VALU issue slots per cell.
// Valid while every score fits a signed 16-bit value (min(Lq, Lt) * largest table entry + 64 < 32767); other pairs use sw_one.
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 pk_shr1z(s16x2 v) { return __builtin_bit_cast(s16x2, shr1z(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ s16x2 pk_shl1z(s16x2 v) { return __builtin_bit_cast(s16x2, shl1z(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// table address = replica base + q part + t part in ONE instruction: left to itself hipcc keeps (base + q) for the B step and spends
// three additions per candidate and step pair instead of two
__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef const __attribute__((address_space(3))) signed char lds_ci8;
__device__ __forceinline__ short tab_at(uint32_t base, int qv, int tv) { return (short)*(lds_ci8 *)(uintptr_t)add3(base, (uint32_t)qv, (uint32_t)tv); }

// H + score for BOTH candidates of a packed wavefront in one 32-bit addition.  Candidate 0's score is read sign-extended (ds_read_i8),
// candidate 1's with ds_read_i8_d16_hi, which on gfx950 delivers score << 16 with a zero low half (tools/micro/d16_probe.hip); with
// H carried as H + PK_BIAS (>= -score for every table entry) the low half never borrows from the high one.  The two loads are inline
// assembly (the compiler does not select the d16 forms on this target) and bring their own s_waitcnt: the compiler's waits stay correct
// with extra loads in flight (the counter is in order), they only become stricter.
constexpr int PK_BIAS = 128;
__device__ __forceinline__ void diag_issue(uint32_t addr0, uint32_t addr1, uint32_t &s0, uint32_t &s1)
{
    asm volatile("ds_read_i8 %0, %2\n\tds_read_i8_d16_hi %1, %3" : "=&v"(s0), "=&v"(s1) : "v"(addr0), "v"(addr1) : "memory");
}
__device__ __forceinline__ uint32_t diag_finish(uint32_t Hd, uint32_t s0, uint32_t s1)
{
    uint32_t r;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(Hd), "v"(s0), "v"(s1));
    return r;
}

struct CandGeom {
    const uint8_t *qg, *tg;
    int Lq, Lt, dlo, a0, nblk;
};

__device__ __forceinline__ CandGeom cand_geom(const SwArgs &a, uint64_t c)
{
    CandGeom g;
    const uint64_t key = a.cands[c];
    const uint32_t q = (uint32_t)(key >> 43), t = (uint32_t)((key >> 18) & ((1u << 25) - 1));
    const int bin = (int)(key & ((1u << 18) - 1));
    g.dlo = bin * 64 - (1 << 23) - 32;
    g.Lq = (int)a.q_len[q]; g.Lt = (int)a.t_len[t];
    g.qg = a.q_res + a.q_off[q]; g.tg = a.t_res + a.t_off[t];
    const int dl = max(g.dlo, -(g.Lq - 1)), dh = min(g.dlo + 127, g.Lt - 1);
    const int s_lo = (dl <= 0 && dh >= 0) ? 0 : (dl > 0 ? dl : -dh);
    const int s0 = s_lo - ((s_lo - g.dlo) & 1);
    g.a0 = (s0 - g.dlo) / 2;
    g.nblk = (int)a.nblk[c];
    return g;
}

__device__ __forceinline__ bool fits16(const CandGeom &g, int max_sub) { return min(g.Lq, g.Lt) * max_sub + 64 + PK_BIAS < 32767; }

// the window entries that blocks [b0, b1) of a sweep read (8 per block + 72 of overlap / slack), stored from index 0: a pair whose windows do not
// fit the staging area is swept in chunks of blocks, each staged on its own (the DP state stays in registers across chunks)
__device__ __forceinline__ void stage_windows(const CandGeom &g, int b0, int b1, uint16_t *lq, uint16_t *lt, int lane)
{
    const int qlo = g.a0 - 64 + 8 * b0, tlo = g.a0 + g.dlo - 1 + 8 * b0, n = 8 * (b1 - b0) + 72;
    for (int x = lane; x < n; x += 64) { const int p = qlo + x; lq[x] = q_addr_part(((unsigned)p < (unsigned)g.Lq) ? g.qg[p] : PEP_PAD_CODE); }
    for (int x = lane; x < n; x += 64) { const int p = tlo + x; lt[x] = t_addr_part(((unsigned)p < (unsigned)g.Lt) ? g.tg[p] : PEP_PAD_CODE); }
}

template <bool CHUNKED>          // false: the windows of the whole pair fit the staging area (one staging, the loop nest of the short pairs untouched)
__device__ __forceinline__ void sw_two_pk16(const SwArgs &a, uint64_t c0, uint64_t c1, const CandGeom &g0, const CandGeom &g1,
                                            const unsigned char *lds_tab, uint16_t *lds_res, int lane)
{
    const int nb = max(g0.nblk, g1.nblk);
    // blocks per staging chunk: four windows of 8 * blocks + 80 entries in the wavefront's staging area (nucleotide pairs of 1 000 bases have
    // 125 blocks and used to miss the packed sweep by seven: they went through the 32-bit one-candidate sweep at half the speed)
    const int chunk = CHUNKED ? max(1, min(nb, (a.lds_res_bytes / 8 - 80) / 8)) : nb;
    const int win = (8 * chunk + 80 + 7) & ~7;
    uint16_t *q0 = lds_res, *t0 = q0 + win, *q1 = t0 + win, *t1 = q1 + win;
    typedef const volatile __attribute__((address_space(3))) uint16_t lds_cu16;
    const uint32_t tab = (uint32_t)(uintptr_t)(lds_ci8 *)(reinterpret_cast<const signed char *>(lds_tab) + (lane & (PEP_TAB_REP - 1)) * 4);
    // every H / E / F below is the true value + PK_BIAS (the recurrences are shift-invariant; the floor 0 becomes PK_BIAS; band-edge
    // lanes read 0 = -PK_BIAS, still "never selected")
    const s16x2 zero = {PK_BIAS, PK_BIAS};
    const s16x2 oe2 = {(short)a.oe, (short)a.oe}, ext2 = {(short)a.ext, (short)a.ext};
    // The gap states are carried as E + (open + extend) and F + (open + extend): "open a gap from H" then needs no subtraction, and the
    // one subtraction moves to the single place where max(E, F) meets H - one instruction less per step.  Every POSITIVE E / F (the
    // only ones that can reach H >= 0) is the same as in the plain recurrence.
    s16x2 HA = zero, EA = zero, FA = zero, HB = zero, EB = zero, FB = zero, best = zero;
    int cb = 0;
    do {
    const int ce = CHUNKED ? min(nb, cb + chunk) : nb;
    if (CHUNKED && cb) {                            // every lane is done with the windows of the chunk before
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    stage_windows(g0, cb, ce, q0, t0, lane);
    stage_windows(g1, cb, ce, q1, t1, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // per-lane read cursors: the A cell of step pair m is (a0 + m - lane, a0 + dlo + m + lane); window origins a0-64 / a0+dlo-1
    // explicit LDS address space: through a generic volatile pointer hipcc emits flat_load_ushort instead of ds_read_u16
    lds_cu16 *vq0 = (lds_cu16 *)(q0 + (64 - lane)), *vt0 = (lds_cu16 *)(t0 + (1 + lane));
    lds_cu16 *vq1 = (lds_cu16 *)(q1 + (64 - lane)), *vt1 = (lds_cu16 *)(t1 + (1 + lane));
    // the residue cursors run one step ahead of the table look-ups: the reads for the next step are issued right behind the two table
    // reads of this one, and the single s_waitcnt of diag_finish covers all four (the windows are staged with 8 entries of slack)
    int tv0 = vt0[0], tv1 = vt1[0], qv0 = vq0[0], qv1 = vq1[0];
    for (int b = cb; b < ce; ++b) {
#pragma unroll 1
        for (int half = 0; half < 2; ++half, vq0 += 4, vq1 += 4, vt0 += 4, vt1 += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // ---- A step: cell (q[u], t[u])
            int tn0, tn1, qn0, qn1;
            {
                uint32_t s0, s1;
                diag_issue(add3(tab, (uint32_t)qv0, (uint32_t)tv0), add3(tab, (uint32_t)qv1, (uint32_t)tv1), s0, s1);
                tn0 = vt0[u + 1]; tn1 = vt1[u + 1];
                // E of this cell = max(E_left - ext, H_left) with both operands in lane l-1: the maximum is taken THERE and one DPP move
                // brings it over (moving H_left and E_left separately costs a second 4-cycle move per step); band-edge lanes read 0 either way
                const s16x2 E = pk_shr1z(pk_max(EB - ext2, HB)), F = pk_max(FB - ext2, HB);
                const s16x2 m = pk_max(E, F) - oe2;
                const s16x2 h = __builtin_bit_cast(s16x2, diag_finish(__builtin_bit_cast(uint32_t, HA), s0, s1));
                const s16x2 H = pk_max(pk_max(h, m), zero);
                best = pk_max(best, H);
                HA = H; EA = E; FA = F;
            }
            // ---- B step (target cursor advances): cell (q[u], t[u + 1])
            {
                uint32_t s0, s1;
                diag_issue(add3(tab, (uint32_t)qv0, (uint32_t)tn0), add3(tab, (uint32_t)qv1, (uint32_t)tn1), s0, s1);
                qn0 = vq0[u + 1]; qn1 = vq1[u + 1];
                const s16x2 E = pk_max(EA - ext2, HA), F = pk_shl1z(pk_max(FA - ext2, HA));
                const s16x2 m = pk_max(E, F) - oe2;
                const s16x2 h = __builtin_bit_cast(s16x2, diag_finish(__builtin_bit_cast(uint32_t, HB), s0, s1));
                const s16x2 H = pk_max(pk_max(h, m), zero);
                best = pk_max(best, H);
                HB = H; EB = E; FB = F;
            }
            tv0 = tn0; tv1 = tn1; qv0 = qn0; qv1 = qn1;
        }
    }
    cb = ce;
    } while (CHUNKED && cb < nb);
    int b0 = best.x - PK_BIAS, b1 = best.y - PK_BIAS;
    const int m0 = b0, m1 = b1;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { b0 = max(b0, __shfl_xor(b0, d, 64)); b1 = max(b1, __shfl_xor(b1, d, 64)); }
    // the lowest lane that met the maximum: the traceback pass centres its sub-band there
    const unsigned long long at0 = __ballot(m0 == b0), at1 = __ballot(m1 == b1);
    if (lane == 0) {
        a.out[c0] = make_int4(b0, (int)__builtin_ctzll(at0), -1, g0.a0);
        a.out[c1] = make_int4(b1, (int)__builtin_ctzll(at1), -1, g1.a0);
    }
}

// ---- packed 16-bit TRACEBACK pass: the same two-candidates-per-wavefront sweep, plus the 4-bit codes of both candidates.
// The codes come out of packed arithmetic instead of compare/select pairs (which would have to run once per candidate):
//   gt(x, y) = (y - x) >> 15 per 16-bit half (the sign bit of the difference): 0/1 flags without compare instructions
//   nz = min(H, 1), neh = gt(H, h), gfe = gt(F, E)                        (H >= 0, H >= h: "greater" is "not equal")
//   src = nz * (1 + neh * (1 + gfe))                                      -> 0 none, 1 diagonal, 2 E, 3 F   (same priority as sw_one)
//   eflag = gt(e_ext, e_open), fflag likewise;  nibble = src + 4 * eflag + 8 * fflag
// Four nibbles per candidate accumulate in the halves of one register (one v_pk_mad_u16 with the place value 16^u each); two such
// registers make the 8-cell word of one diagonal, rearranged per candidate with v_perm_b32 once per 16 steps.
// The end cell is the first step at which H equals the candidate's score T, which the score pass already delivered.  The running
// maximum of a lane only grows, so that step number is the count of steps at which the maximum was still below T:
//   best = max(best, H);  first += gt(T, best)                             (four instructions per step; < 0x8000 steps)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// The flag arithmetic goes through inline assembly: written as C, LLVM recognises the 0/1 values, turns the multiplications into
// selects and ends up with one 16-bit compare + select per HALF and v_perm to repack - more instructions than the 32-bit kernel.
// x > y per half as 0/1: the sign bit of y - x (no overflow: all DP values stay far inside +-2^15)
__device__ __forceinline__ u16x2 pk_gt(s16x2 x, s16x2 y)
{
    u16x2 r;
    asm("v_pk_sub_i16 %0, %1, %2\n\tv_pk_lshrrev_b16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(r) : "v"(y), "v"(x));
    return r;
}
__device__ __forceinline__ u16x2 pk_mad(u16x2 a, u16x2 b, u16x2 c)
{
    u16x2 r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct PkConst { u16x2 four, eight, place[4]; };      // place[u] = 16^u: the nibble of the u-th cell of a half block goes to bits 4u .. 4u+3

// min(x, 1) per half = "x != 0" for x >= 0
__device__ __forceinline__ u16x2 pk_nonzero(s16x2 x)
{
    u16x2 r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
    return r;
}

// E and F arrive in the pre-opened form (true value + open + extend, as in the score pass): H is neither h nor 0 exactly when it came
// out of max(E, F), and then the source is F only if F is strictly greater (the same priority as sw_one: diagonal, E, F)
__device__ __forceinline__ u16x2 pk_codes(const PkConst &K, s16x2 H, s16x2 h, s16x2 E, s16x2 F, s16x2 e_ext, s16x2 e_open, s16x2 f_ext, s16x2 f_open)
{
    const u16x2 nz = pk_nonzero(H), neh = pk_gt(H, h), gfe = pk_gt(F, E);       // H >= 0, H >= h: "greater" == "not equal"
    const u16x2 t = pk_mad(neh, gfe, neh);
    const u16x2 src = pk_mad(nz, t, nz);
    const u16x2 ef = pk_gt(e_ext, e_open), ff = pk_gt(f_ext, f_open);
    return pk_mad(ff, K.eight, pk_mad(ef, K.four, src));
}

// Geometry of a candidate's SUB-band (lanes [L0, L0 + 32) of its band)
struct SubGeom {
    const uint8_t *qg, *tg;
    int Lq, Lt, dlo, a0, nblk;
};
__device__ __forceinline__ SubGeom sub_geom(const CandGeom &g, int L0)
{
    SubGeom s;
    s.qg = g.qg; s.tg = g.tg; s.Lq = g.Lq; s.Lt = g.Lt;
    s.dlo = g.dlo + 2 * L0;
    band_geom(g.Lq, g.Lt, s.dlo, 2 * SUB_LANES, s.a0, s.nblk);
    return s;
}
// (the entries that blocks [b0, b1) read, stored from index 0: long pairs are swept in chunks of blocks, see stage_windows)
__device__ __forceinline__ void stage_sub_windows(const SubGeom &g, int b0, int b1, uint16_t *lq, uint16_t *lt, int lane)
{
    const int qlo = g.a0 - SUB_LANES + 8 * b0, tlo = g.a0 + g.dlo - 1 + 8 * b0, n = 8 * (b1 - b0) + SUB_LANES + 8;
    for (int x = lane; x < n; x += 64) { const int p = qlo + x; lq[x] = q_addr_part(((unsigned)p < (unsigned)g.Lq) ? g.qg[p] : PEP_PAD_CODE); }
    for (int x = lane; x < n; x += 64) { const int p = tlo + x; lt[x] = t_addr_part(((unsigned)p < (unsigned)g.Lt) ? g.tg[p] : PEP_PAD_CODE); }
}

// FOUR candidates per wavefront: lanes 0..31 sweep the sub-bands of candidates 0 / 1 (low / high half of every packed register), lanes
// 32..63 those of candidates 2 / 3.  The recurrences, the 4-bit codes and the end-cell search are those of the two-candidate packed
// sweep (see above); the only additions are the AND masks that make lanes 0 / 31 of each half read the band boundary instead of the
// other half's edge lane (v_and_b32 issues at twice the rate of the packed operations).  ok[x] = candidate x reached its known score
// inside its sub-band (wave-uniform); a candidate that did not is traced again in its full band by the caller.
template <bool CHUNKED>
__device__ __forceinline__ void sw_four_pk16_trace(const SwArgs &a, const uint64_t cc[4], const SubGeom gg[4], const int L0[4], bool ok[4],
                                                   const unsigned char *lds_tab, uint16_t *lds_res, int lane)
{
    const int nb = max(max(gg[0].nblk, gg[1].nblk), max(gg[2].nblk, gg[3].nblk));
    // blocks per staging chunk: eight windows of 8 * blocks + 48 entries in the wavefront's staging area; pairs with more blocks (nucleotide
    // alignments, long proteins) are swept chunk by chunk with the DP state kept in registers - they used to go one per wavefront through the 32-bit sweep
    const int chunk = CHUNKED ? max(1, min(nb, (a.lds_res_bytes / 16 - (SUB_LANES + 16)) / 8)) : nb;
    const int win = (8 * chunk + SUB_LANES + 16 + 7) & ~7;
    uint16_t *wq[4], *wt[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { wq[x] = lds_res + (2 * x) * win; wt[x] = lds_res + (2 * x + 1) * win; }
    const int hf = lane >> 5, ln = lane & (SUB_LANES - 1);      // which pair of candidates, which lane of the sub-band
    typedef const volatile __attribute__((address_space(3))) uint16_t lds_cu16;
    const uint32_t tab = (uint32_t)(uintptr_t)(lds_ci8 *)(reinterpret_cast<const signed char *>(lds_tab) + (lane & (PEP_TAB_REP - 1)) * 4);
    const uint64_t c0 = hf ? cc[2] : cc[0], c1 = hf ? cc[3] : cc[1];
    const int nb0 = hf ? gg[2].nblk : gg[0].nblk, nb1 = hf ? gg[3].nblk : gg[1].nblk;
    uint2 *dir0 = reinterpret_cast<uint2 *>(a.dirs) + a.dir_off[c0] * 64, *dir1 = reinterpret_cast<uint2 *>(a.dirs) + a.dir_off[c1] * 64;
    const s16x2 zero = {0, 0};
    const s16x2 oe2 = {(short)a.oe, (short)a.oe}, ext2 = {(short)a.ext, (short)a.ext};
    const s16x2 T2 = {(short)a.known[c0], (short)a.known[c1]};
    const int keep_l = ln == 0 ? 0 : -1, keep_r = ln == SUB_LANES - 1 ? 0 : -1;
    auto edge_l = [&](s16x2 v) { return __builtin_bit_cast(s16x2, __builtin_bit_cast(int, v) & keep_l); };
    auto edge_r = [&](s16x2 v) { return __builtin_bit_cast(s16x2, __builtin_bit_cast(int, v) & keep_r); };
    PkConst K;
    K.four = u16x2{4, 4}; K.eight = u16x2{8, 8};
    K.place[0] = u16x2{1, 1}; K.place[1] = u16x2{16, 16}; K.place[2] = u16x2{256, 256}; K.place[3] = u16x2{4096, 4096};
    s16x2 HA = zero, EA = zero, FA = zero, HB = zero, EB = zero, FB = zero;
    u16x2 first = {0, 0};
    s16x2 best = zero;
    int cb = 0;
    do {
    const int ce = CHUNKED ? min(nb, cb + chunk) : nb;
    if (CHUNKED && cb) {                            // every lane is done with the windows of the chunk before
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) stage_sub_windows(gg[x], cb, ce, wq[x], wt[x], lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    lds_cu16 *vq0 = (lds_cu16 *)((hf ? wq[2] : wq[0]) + (SUB_LANES - ln)), *vt0 = (lds_cu16 *)((hf ? wt[2] : wt[0]) + (1 + ln));
    lds_cu16 *vq1 = (lds_cu16 *)((hf ? wq[3] : wq[1]) + (SUB_LANES - ln)), *vt1 = (lds_cu16 *)((hf ? wt[3] : wt[1]) + (1 + ln));
    int tv0 = vt0[0], tv1 = vt1[0], qv0 = 0, qv1 = 0;
    for (int b = cb; b < ce; ++b) {
        u16x2 accA = {0, 0}, accB = {0, 0}, loA = {0, 0}, loB = {0, 0};
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // ---- A step
                qv0 = vq0[u]; qv1 = vq1[u];
                {
                    const s16x2 sub = {tab_at(tab, qv0, tv0), tab_at(tab, qv1, tv1)};
                    const s16x2 hl = edge_l(pk_shr1z(HB)), el = edge_l(pk_shr1z(EB));
                    const s16x2 e_ext = el - ext2, f_ext = FB - ext2;                 // "open from H" needs no subtraction in this form
                    const s16x2 E = pk_max(e_ext, hl), F = pk_max(f_ext, HB);
                    const s16x2 h = HA + sub;
                    const s16x2 H = pk_max(pk_max(h, pk_max(E, F) - oe2), zero);
                    {   // u is a compile-time constant of the unrolled loop: the first nibble of a half block simply starts the word
                        const u16x2 code = pk_codes(K, H, h, E, F, e_ext, hl, f_ext, HB);
                        accA = u == 0 ? code : pk_mad(code, K.place[u], accA);
                    }
                    best = pk_max(best, H);
                    first += pk_gt(T2, best);
                    HA = H; EA = E; FA = F;
                }
                // ---- B step (target cursor advances)
                tv0 = vt0[u + 1]; tv1 = vt1[u + 1];
                {
                    const s16x2 sub = {tab_at(tab, qv0, tv0), tab_at(tab, qv1, tv1)};
                    const s16x2 hu = edge_r(pk_shl1z(HA)), fu = edge_r(pk_shl1z(FA));
                    const s16x2 e_ext = EA - ext2, f_ext = fu - ext2;
                    const s16x2 E = pk_max(e_ext, HA), F = pk_max(f_ext, hu);
                    const s16x2 h = HB + sub;
                    const s16x2 H = pk_max(pk_max(h, pk_max(E, F) - oe2), zero);
                    {
                        const u16x2 code = pk_codes(K, H, h, E, F, e_ext, HA, f_ext, hu);
                        accB = u == 0 ? code : pk_mad(code, K.place[u], accB);
                    }
                    best = pk_max(best, H);
                    first += pk_gt(T2, best);
                    HB = H; EB = E; FB = F;
                }
            }
            vq0 += 4; vq1 += 4; vt0 += 4; vt1 += 4;
            if (half == 0) { loA = accA; loB = accB; }
        }
        // word of the low-half candidate = low halves (first four cells | next four cells), of the high-half candidate = high halves
        const uint32_t la = __builtin_bit_cast(uint32_t, loA), ha = __builtin_bit_cast(uint32_t, accA);
        const uint32_t lb = __builtin_bit_cast(uint32_t, loB), hb = __builtin_bit_cast(uint32_t, accB);
        if (b < nb0) dir0[(size_t)b * SUB_LANES + ln] = make_uint2(__builtin_amdgcn_perm(ha, la, 0x05040100u), __builtin_amdgcn_perm(hb, lb, 0x05040100u));
        if (b < nb1 && c1 != c0) dir1[(size_t)b * SUB_LANES + ln] = make_uint2(__builtin_amdgcn_perm(ha, la, 0x07060302u), __builtin_amdgcn_perm(hb, lb, 0x07060302u));
    }
    cb = ce;
    } while (CHUNKED && cb < nb);
    // end cell per candidate: earliest step with H == T in the lane, then min i, then min j over the 32 lanes of the half
    const int fk[2] = {(int)first.x, (int)first.y};
    const int TT[2] = {(int)T2.x, (int)T2.y};
    bool okh[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int ga0 = hf ? gg[2 + x].a0 : gg[x].a0, gdlo = hf ? gg[2 + x].dlo : gg[x].dlo;
        const uint64_t c = x ? c1 : c0;
        int bi = 0x7fffffff, bj = 0x7fffffff;
        if (fk[x] < 16 * nb) {                    // (a lane that never reaches T counted every step)
            const int mm = fk[x] >> 1;
            bi = ga0 + mm - ln;
            bj = ga0 + gdlo + mm + ln + (fk[x] & 1);
        }
#pragma unroll
        for (int d = SUB_LANES / 2; d > 0; d >>= 1) {
            const int oi = __shfl_xor(bi, d, 64), oj = __shfl_xor(bj, d, 64);
            if (oi < bi || (oi == bi && oj < bj)) { bi = oi; bj = oj; }
        }
        okh[x] = bi != 0x7fffffff;
        if (ln == 0 && okh[x] && (x == 0 || c1 != c0)) {
            a.out[c] = make_int4(TT[x], bi, bj, ga0);
            a.mode[c] = hf ? L0[2 + x] : L0[x];
        }
    }
    // wave-uniform outcome of all four
    ok[0] = __shfl((int)okh[0], 0, 64) != 0;  ok[1] = __shfl((int)okh[1], 0, 64) != 0;
    ok[2] = __shfl((int)okh[0], 32, 64) != 0; ok[3] = __shfl((int)okh[1], 32, 64) != 0;
}

// score pass: wave w of the grid-stride loop takes the candidate pair (order[2w], order[2w+1]): neighbours in the length order, so the
// two halves of the packed registers finish together, and the longest pairs start first
// LONG = false: the pairs whose windows fit the staging area (and the ones the packed sweep cannot take at all); LONG = true: the pairs that the
// packed sweep takes in chunks of blocks - a launch of its own, made only when the sequence lengths allow such pairs, because the chunked
// sweep next to the plain one made both kernels a few per cent slower on short pairs (code size, register allocation)
template <bool LONG>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void sw_score_kernel(SwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint64_t n_act = a.counts[5];                          // candidates in the order (sw_prep): all of them but the ones settled without a sweep
    const uint64_t n_pairs = (n_act + 1) / 2;
    // the candidates with more blocks than the staging area takes at once are a prefix of the length order (sw_prep counted them): the pairs
    // [0, w_long) belong to the LONG launch, the rest to the other
    const uint64_t w_long = a.split_long ? min(n_pairs, ((uint64_t)a.counts[3] + 1) / 2) : 0;
    if ((LONG ? 0 : w_long) + (uint64_t)blockIdx.x * WAVES_PER_BLOCK >= (LONG ? w_long : n_pairs)) return;      // nothing for this block (before it loads the table)
    uint32_t *lds_tab = reinterpret_cast<uint32_t *>(smem);
    for (int x = threadIdx.x; x < LDS_TABLE_BYTES / 4; x += blockDim.x) lds_tab[x] = a.sub_image[x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *lds_res = reinterpret_cast<uint16_t *>(smem + LDS_TABLE_BYTES + (size_t)wave * a.lds_res_bytes);
    for (uint64_t w = (LONG ? 0 : w_long) + (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave; w < (LONG ? w_long : n_pairs); w += (uint64_t)gridDim.x * WAVES_PER_BLOCK) {
        const uint64_t c0 = a.order[2 * w], c1 = a.order[min(2 * w + 1, n_act - 1)];
        const CandGeom g0 = cand_geom(a, c0), g1 = cand_geom(a, c1);
        const int nbm = max(g0.nblk, g1.nblk);
        const bool can_pack = a.pk16 && fits16(g0, a.max_sub) && fits16(g1, a.max_sub) && a.lds_res_bytes >= 4 * 2 * 88;
        const bool fits = 4 * 2 * ((8 * nbm + 80 + 7) & ~7) <= a.lds_res_bytes;
        if (LONG && can_pack) sw_two_pk16<LONG>(a, c0, c1, g0, g1, smem, lds_res, lane);
        else if (!LONG && can_pack && fits) {
            sw_two_pk16<false>(a, c0, c1, g0, g1, smem, lds_res, lane);
        } else {
            for (int x = 0; x < (c1 != c0 ? 2 : 1); ++x) {
                const uint64_t c = x ? c1 : c0;
                const int need1 = 2 * 2 * ((8 * (int)a.nblk[c] + 72 + 7) & ~7);
                if (need1 <= a.lds_res_bytes) sw_one<true, false>(a, c, smem, lds_res, lane);
                else sw_one<false, false>(a, c, smem, lds_res, lane);
            }
        }
    }
}

// The traceback kernel's rare paths - the 32-bit sweep for a candidate the packed sweep cannot take, the full band for an alignment that
// left its sub-band - as ONE function that is NOT inlined: inlined, their live ranges sat in the kernel's register allocation next to the
// packed sweep's (128 VGPRs at the 4 waves per SIMD the LDS budget allows, 55 vector + 47 scalar spills, 96 B of scratch).  The arguments
// are read from the kernel-argument segment (a reference to the by-value kernel parameter would put a copy of it into private memory).
__device__ __attribute__((noinline)) void trace_slow_path(const SwArgs *ka, uint64_t c, const unsigned char *smem, uint16_t *lds_res, int lane, int full_band)
{
    const SwArgs &a = *ka;
    const int need1 = 2 * 2 * ((8 * (int)a.nblk[c] + 72 + 7) & ~7);
    if (full_band) {
        if (need1 <= a.lds_res_bytes) sw_one<true, true>(a, c, smem, lds_res, lane, -1);
        else sw_one<false, true>(a, c, smem, lds_res, lane, -1);
        if (lane == 0) a.mode[c] = -1;
    } else {
        if (need1 <= a.lds_res_bytes) trace_one<true>(a, c, smem, lds_res, lane);
        else trace_one<false>(a, c, smem, lds_res, lane);
    }
}

// traceback pass over the pairs that survived best-per-(q,t) and the e-value cut: four candidates per wavefront in their sub-bands
// (packed 16-bit), the 32-bit sweep for anything that does not fit (scores beyond 16 bits, windows beyond the LDS staging area), and the
// full band for the few alignments that leave their sub-band
template <bool LONG>            // true: the long candidates only (part 0 of the order), four per wavefront in chunks of blocks - see sw_score_kernel
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, 4) void sw_trace_kernel(SwArgs a)      // 4 waves per SIMD is what the LDS budget allows: keep the VGPRs within that
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds_tab = reinterpret_cast<uint32_t *>(smem);
    for (int x = threadIdx.x; x < LDS_TABLE_BYTES / 4; x += blockDim.x) lds_tab[x] = a.sub_image[x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *lds_res = reinterpret_cast<uint16_t *>(smem + LDS_TABLE_BYTES + (size_t)wave * a.lds_res_bytes);
    const SwArgs *ka = (const SwArgs *)__builtin_amdgcn_kernarg_segment_ptr();      // (the kernel's only argument)
    // The grid is only as large as the chip holds at once and every wavefront PULLS its next item (four candidates) from a counter.
    // With one item per wavefront and a block per eight items the 80 KB of LDS a block holds came free only when its slowest
    // wavefront was done: 2.8 of 4 wavefronts per SIMD on average and 64 % VALU utilisation (SQ_WAVE_CYCLES / SQ_INSTS_VALU,
    // profiles/r02_pmc_counters.txt).  The items are sorted by decreasing length, so the tail of the queue is its shortest work.
    // The counts live on the device (sw_prep wrote them): the host queues this launch without having seen them.  Part 0 = the prefix of the
    // order that is too long for the staging area, one candidate per item; part 1 = the rest, four candidates per item.
    const uint64_t n_active = a.counts[5];
    const uint64_t n_long = a.pk16 ? min((uint64_t)a.counts[3], n_active) : n_active;
#pragma unroll 1
    for (int part = LONG ? 0 : (a.split_long ? 1 : 0); part < (LONG ? 1 : 2); ++part) {
        const uint64_t item_first = part ? n_long : 0, item_count = part ? n_active - n_long : n_long;
        const int item_span = (part || LONG) ? 4 : 1;
        unsigned int *queue = reinterpret_cast<unsigned int *>(a.counts + (part ? 4 : 2));
        if (item_count == 0) continue;
        const uint64_t n_items = (item_count + item_span - 1) / item_span;
        for (;;) {
            unsigned int wq = 0;
            if (lane == 0) wq = atomicAdd(queue, 1u);
            const uint64_t w = (uint64_t)__builtin_amdgcn_readfirstlane((int)wq);
            if (w >= n_items) break;
            uint64_t cc[4];
            int n_own = 0;                                       // candidates of this item (the last item may hold fewer than four)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const uint64_t k = (uint64_t)item_span * w + x;
                const bool own = x < item_span && k < item_count;
                cc[x] = a.order[item_first + min(k, item_count - 1)];
                n_own += own ? 1 : 0;
            }
            bool packed = a.pk16 && a.known && a.end_lane && n_own == 4;
            SubGeom gg[4];
            int L0[4];
            if (packed) {
                int nb = 0;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const CandGeom g = cand_geom(a, cc[x]);
                    L0[x] = sub_band_first_lane(a.end_lane[cc[x]]);
                    gg[x] = sub_geom(g, L0[x]);
                    nb = max(nb, gg[x].nblk);
                    packed = packed && fits16(g, a.max_sub) && a.known[cc[x]] > 0;
                }
                packed = packed && (LONG ? a.lds_res_bytes >= 8 * 2 * (SUB_LANES + 24) : 8 * 2 * ((8 * nb + SUB_LANES + 16 + 7) & ~7) <= a.lds_res_bytes) && nb < 2040;
            }
            if (packed) {
                bool ok[4];
                sw_four_pk16_trace<LONG>(a, cc, gg, L0, ok, smem, lds_res, lane);
#ifdef PEP_WALK_PROXY
                // MEASUREMENT ONLY (make EXTRA=-DPEP_WALK_PROXY=2500; profiles/r06_walk_fusion_proxy.txt): what the four walks of this item would cost the pass if they ran
                // here, one lane per pair - a dependent chain of 3 x PEP_WALK_PROXY vector instructions on four lanes (the walk kernel issues 4.5 M wave instructions for 590
                // wavefronts: 7 600 per wavefront, whatever the number of its lanes that walk)
                {
                    uint32_t x = (uint32_t)cc[0] + (uint32_t)lane;
                    if (lane < 4) for (int i = 0; i < PEP_WALK_PROXY; ++i) x = (x ^ (uint32_t)i) + (x >> 3);
                    if (x == 0x12345u) a.mode[cc[0]] = -7;
                }
#endif
                // a pair whose alignment left its sub-band is swept once more in the full band, one pair per wavefront in 32 bits: the longest single
                // piece of work of the pass (a millisecond for a pair of 1 000 bases).  It is not done here - a wavefront that met three of them held the
                // whole launch up (nucleotide tool, 10 000 genes: 1 % of the pairs, 1.98 ms with 1.9 of 4 wavefronts per SIMD resident on average) - but
                // put on a list that a launch of its own works off, one pair per wavefront over the whole chip (sw_trace_retry_kernel)
#pragma unroll 1
                for (int x = 0; x < 4; ++x)
                    if (!ok[x] && lane == 0) a.defer[atomicAdd(a.counts + 6, 1ull)] = (uint32_t)cc[x];
            } else {
#pragma unroll 1
                for (int x = 0; x < n_own; ++x) trace_slow_path(ka, cc[x], smem, lds_res, lane, 0);
            }
        }
    }
}

// The pairs whose alignment left its sub-band (sw_trace_kernel put them on a list: counts[6] of them), swept once more in the full band: one pair per
// wavefront, every wavefront of the chip pulling from the list (counts[7] = pairs taken).
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, 4) void sw_trace_retry_kernel(SwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long n_list = a.counts[6];
    if ((unsigned long long)blockIdx.x * WAVES_PER_BLOCK >= n_list) return;          // nothing for this block (before it loads the table)
    uint32_t *lds_tab = reinterpret_cast<uint32_t *>(smem);
    for (int x = threadIdx.x; x < LDS_TABLE_BYTES / 4; x += blockDim.x) lds_tab[x] = a.sub_image[x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *lds_res = reinterpret_cast<uint16_t *>(smem + LDS_TABLE_BYTES + (size_t)wave * a.lds_res_bytes);
    const SwArgs *ka = (const SwArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    for (;;) {
        unsigned long long w = 0;
        if (lane == 0) w = atomicAdd(a.counts + 7, 1ull);
        w = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= n_list) break;
        trace_slow_path(ka, (uint64_t)a.defer[w], smem, lds_res, lane, 1);
    }
}

// Score pass without a sweep for IDENTICAL pairs (every gene of an all-vs-all against itself: a fifth of the candidates of the benchmark, more
// on real exemplar sets, where most genes have few homologues).  One wavefront per candidate.  Residue classes, derived from the score table
// of the search: r is DOMINANT if sub[r][r] > 0 and sub[r][x] < sub[r][r] for every other x, HARMLESS if no entry of its row is positive
// (BLOSUM62: the twenty amino acids are dominant, X - which is what a stop codon becomes - is harmless, B / Z are neither).  If the two
// sequences are the same string P + H - one or more dominant residues, then any number of harmless ones (the stop at the end of a gene) - and
// the band contains diagonal 0, the banded optimum is diagonal 0 over P:
//   * a path collects at most max_x sub[r_i][x] for every row i it touches and pays for its gaps (every gap costs something: checked on the host), i.e. at most the sum S of sub[r][r] over
//     the rows of P it touches (rows of H add nothing positive);
//   * S over ALL of P needs every row of P, no gap, and equality in every row: a gap-free path through all rows of P lies on one diagonal k,
//     k < 0 misses row 0, and k > 0 puts the last k rows of P against columns of H, where strict dominance makes it lose.
// So the score is S, reached on diagonal 0 and on no other, i.e. in lane (-dlo) >> 1: exactly what the sweep writes (score, lowest lane that
// met it, a0).  The traceback stage then settles the pair by rule 5a.  settled[c] = -2 (the sweep skips the candidate) or 0.
struct IdentArgs { int8_t diag[32]; uint32_t dominant, harmless; };
__global__ __launch_bounds__(256) void identical_check(const uint64_t *__restrict__ cands, uint64_t n, const uint8_t *__restrict__ q_res, const uint32_t *__restrict__ q_off,
                                                       const uint32_t *__restrict__ q_len, const uint8_t *__restrict__ t_res, const uint32_t *__restrict__ t_off,
                                                       const uint32_t *__restrict__ t_len, IdentArgs ia, int4 *__restrict__ out, int32_t *__restrict__ settled)
{
    const int lane = threadIdx.x & 63;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n) return;
    const uint64_t key = cands[c];
    const uint32_t q = (uint32_t)(key >> 43), t = (uint32_t)((key >> 18) & ((1u << 25) - 1));
    const int bin = (int)(key & ((1u << 18) - 1));
    const int dlo = bin * 64 - (1 << 23) - 32;
    const int L = (int)q_len[q];
    const int my_diag = ia.diag[lane & 31];                       // lane r holds sub[r][r]
    bool same = L > 0 && L == (int)t_len[t] && dlo <= 0 && dlo + 127 >= 0;
    bool tail = false;                                            // a harmless residue has been seen: only harmless ones may follow
    int sum = 0;
    if (same) {
        const uint8_t *qg = q_res + q_off[q], *tg = t_res + t_off[t];
        for (int x0 = 0; x0 < L && same; x0 += 256) {               // (wave-uniform trip count: the look-up below is a shuffle)
            // four strips of 64 residues per trip, their eight loads in flight together: the kernel's time is the chain of these round trips
            uint32_t qa[4], ta[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int x = x0 + 64 * k + lane; qa[k] = x < L ? qg[x] : 0u; ta[k] = x < L ? tg[x] : 0u; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = x0 + 64 * k + lane < L;
                const int d = __shfl(my_diag, (int)(qa[k] & 31u), 64);
                const bool dom = in && qa[k] < 32u && ((ia.dominant >> qa[k]) & 1u), harm = in && qa[k] < 32u && ((ia.harmless >> qa[k]) & 1u);
                const unsigned long long m_dom = __ballot(dom), m_harm = __ballot(harm);
                const bool bad = __ballot(in && (qa[k] != ta[k] || !(dom || harm))) != 0ull                               // differs, or a residue of neither class
                                 || (m_dom && (tail || (m_harm && (63 - __builtin_clzll(m_dom)) > (__builtin_ffsll((long long)m_harm) - 1))));      // dominant behind harmless
                if (bad) same = false;
                if (dom) sum += d;
                tail = tail || m_harm != 0ull;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    same = same && sum > 0;
    if (lane == 0) {
        settled[c] = same ? -2 : 0;
        if (same) out[c] = make_int4(sum, (-dlo) >> 1, -1, (-dlo) >> 1);       // (a0 of a band that holds diagonal 0 is the lane of diagonal 0: band_geom)
    }
}

// per candidate: number of 8-step blocks and the exact count of in-band in-matrix cells
__global__ __launch_bounds__(256) void sw_prep(const uint64_t *__restrict__ cands, uint64_t n_host, const uint32_t *__restrict__ d_n, const uint32_t *__restrict__ q_len,
                                               const uint32_t *__restrict__ t_len, uint32_t *__restrict__ nblk, uint64_t *__restrict__ nblk64,
                                               unsigned long long *__restrict__ cells_total,           // [0] cells, [1] 16-step blocks, [3] candidates above nb_limit
                                               uint32_t *__restrict__ len_hist,                        // [LEN_BUCKETS] candidates per length bucket
                                               uint32_t nb_limit,
                                               const int32_t *__restrict__ skip_mode,                  // candidates settled without a sweep (mode -2: gapless_check, identical_check) take no part
                                               int count_skipped,                                      // score pass: their cells still count in [0] (and, with their number, in [6] [7])
                                               uint64_t *__restrict__ dir_off,                         // traceback pass: start of every candidate's traceback codes (in 16-step blocks) = exclusive
                                               uint64_t *__restrict__ lb_state, uint32_t lb_ticket_base, uint64_t lb_epoch)      //   scan of the block counts, taken in here (lookback.h)
{
    __shared__ uint32_t lh[LEN_BUCKETS];
    __shared__ uint32_t s_tile;
    for (int x = threadIdx.x; x < LEN_BUCKETS; x += 256) lh[x] = 0;
    __syncthreads();
    const uint64_t n = d_n ? (uint64_t)*d_n : n_host;          // (the grid is sized from n_host, an upper bound, when the count lives on the device)
    const uint32_t tile = lb_state ? lb_take_tile(lb_state, lb_ticket_base, &s_tile) : blockIdx.x;
    const uint64_t c = (uint64_t)tile * 256 + threadIdx.x;
    unsigned long long cells = 0, blocks = 0, cells_skipped = 0, n_skipped = 0;
    bool is_long = false;
    const bool skipped = c < n && skip_mode && skip_mode[c] == -2;
    if (skipped && !count_skipped) { nblk[c] = 0; nblk64[c] = 0; }
    else if (c < n) {
        const uint64_t key = cands[c];
        const uint32_t q = (uint32_t)(key >> 43), t = (uint32_t)((key >> 18) & ((1u << 25) - 1));
        const int bin = (int)(key & ((1u << 18) - 1));
        const int dlo = bin * 64 - (1 << 23) - 32;
        const int Lq = (int)q_len[q], Lt = (int)t_len[t];
        const int dl = max(dlo, -(Lq - 1)), dh = min(dlo + 127, Lt - 1);
        int steps = 0;
        if (dl <= dh) {
            const int s_lo = (dl <= 0 && dh >= 0) ? 0 : (dl > 0 ? dl : -dh);
            const int s0 = s_lo - ((s_lo - dlo) & 1);
            const int dstar = min(max(Lt - Lq, dl), dh);
            const int s_hi = 2 * min(Lq - 1, Lt - 1 - dstar) + dstar;
            steps = s_hi - s0 + 1;
            // in-matrix cells of the diagonals dl .. dh: sum of min(Lq, Lt - d) + min(0, d), in closed form (a loop over the 128 diagonals of
            // every candidate was most of this kernel's time)
            auto tri = [](long long a, long long b) { return b < a ? 0ll : (a + b) * (b - a + 1) / 2; };      // a + (a + 1) + ... + b
            const long long k = (long long)Lt - Lq;                       // min(Lq, Lt - d) = Lq for d <= k, Lt - d beyond
            const long long flat_hi = min((long long)dh, k), slope_lo = max((long long)dl, k + 1);
            long long total = (flat_hi >= dl ? (flat_hi - dl + 1) * (long long)Lq : 0ll);
            if (slope_lo <= dh) total += (long long)(dh - slope_lo + 1) * Lt - tri(slope_lo, dh);
            total += tri(dl, min((long long)dh, -1ll));                     // + d for the negative diagonals
            cells += (unsigned long long)total;
        }
        if (skipped) { nblk[c] = 0; nblk64[c] = 0; cells_skipped = cells; n_skipped = 1; }
        else {
            const uint32_t nb = (uint32_t)((steps + 15) / 16);
            nblk[c] = nb;
            nblk64[c] = nb;
            blocks = nb;
            is_long = nb > nb_limit;
            atomicAdd(&lh[len_bucket(nb)], 1u);
        }
    }
    if (lb_state) {
        __shared__ uint64_t lds64[4], s_pre;
        uint64_t tot;
        const uint64_t ex = block_excl_scan_256<uint64_t>(blocks, &tot, lds64);
        if (threadIdx.x < 64) {
            const uint64_t p = lb_tile_prefix<48>(lb_state + 1, tile, tot, lb_epoch, (int)threadIdx.x);
            if (threadIdx.x == 0) s_pre = p;
        }
        __syncthreads();
        if (c <= n) dir_off[c] = s_pre + ex;                    // (entry n: the total)
    }
    const bool active = c < n && !(skip_mode && skip_mode[c] == -2);
    const int longs = __syncthreads_count(is_long);
    const int actives = __syncthreads_count(active);
    if (threadIdx.x == 0 && longs) atomicAdd(&cells_total[3], (unsigned long long)longs);
    if (threadIdx.x == 0 && actives) atomicAdd(&cells_total[5], (unsigned long long)actives);      // candidates that enter the order (and the sweep)
    // totals: one pair of atomics per block (every wavefront adding to the same two words serialises in the L2)
    __shared__ unsigned long long tot[4][4];
    for (int d = 32; d > 0; d >>= 1) {
        cells += __shfl_down(cells, d, 64); blocks += __shfl_down(blocks, d, 64);
        cells_skipped += __shfl_down(cells_skipped, d, 64); n_skipped += __shfl_down(n_skipped, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { tot[0][threadIdx.x >> 6] = cells; tot[1][threadIdx.x >> 6] = blocks; tot[2][threadIdx.x >> 6] = cells_skipped; tot[3][threadIdx.x >> 6] = n_skipped; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long v = tot[threadIdx.x][0] + tot[threadIdx.x][1] + tot[threadIdx.x][2] + tot[threadIdx.x][3];
        if (v) atomicAdd(&cells_total[threadIdx.x < 2 ? threadIdx.x : threadIdx.x + 4], v);          // [0] cells, [1] blocks, [6] cells settled without a sweep, [7] their number
    }
    for (int x = threadIdx.x; x < LEN_BUCKETS; x += 256) if (lh[x]) atomicAdd(&len_hist[x], lh[x]);
}

// candidates in the order of decreasing length bucket (counting sort over sw_prep's histogram; the order inside a bucket is whatever the
// atomics hand out - results are per candidate, so it does not matter).  Equal lengths side by side keep both halves of a packed
// wavefront busy to the end; longest first keeps the tail of the launch short.
__global__ __launch_bounds__(256) void sw_order(const uint32_t *__restrict__ nblk, uint64_t n_host, const uint32_t *__restrict__ d_n, const uint32_t *__restrict__ len_hist,
                                                uint32_t *__restrict__ cursor, uint32_t *__restrict__ order, const int32_t *__restrict__ skip_mode)
{
    __shared__ uint32_t start[LEN_BUCKETS], lh[LEN_BUCKETS], base[LEN_BUCKETS], part[256];
    // start[b] = candidates in longer buckets: every block rebuilds the (tiny) scan
    uint32_t mine[LEN_BUCKETS / 256], sum = 0;
#pragma unroll
    for (int k = 0; k < LEN_BUCKETS / 256; ++k) { mine[k] = len_hist[LEN_BUCKETS - 1 - (threadIdx.x * (LEN_BUCKETS / 256) + k)]; sum += mine[k]; }
    part[threadIdx.x] = sum;
    for (int x = threadIdx.x; x < LEN_BUCKETS; x += 256) lh[x] = 0;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < LEN_BUCKETS / 256; ++k) { start[LEN_BUCKETS - 1 - (threadIdx.x * (LEN_BUCKETS / 256) + k)] = run; run += mine[k]; }
    const uint64_t n = d_n ? (uint64_t)*d_n : n_host;
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t b = 0, rank = 0;
    const bool active = c < n && !(skip_mode && skip_mode[c] == -2);
    if (active) { b = len_bucket(nblk[c]); rank = atomicAdd(&lh[b], 1u); }
    __syncthreads();
    for (int x = threadIdx.x; x < LEN_BUCKETS; x += 256) if (lh[x]) base[x] = atomicAdd(&cursor[x], lh[x]);
    __syncthreads();
    if (active) order[start[b] + base[b] + rank] = (uint32_t)c;
}

// one wavefront of the context's start-up self-test (pep_selftest_dpp): the lane semantics of the two DPP shifts every sweep relies on
__global__ void dpp_selftest(int *out)
{
    const int lane = threadIdx.x;
    out[lane] = shr1(-1, lane);
    out[64 + lane] = shl1(-2, lane);
}

}  // namespace

int pep_selftest_dpp(pep_ctx *ctx)
{
    PEP_TRY(dev_reserve(ctx, ctx->ws[9], 128 * sizeof(int)));
    hipLaunchKernelGGL(dpp_selftest, dim3(1), dim3(64), 0, ctx->stream, ctx->ws[9].as<int>());
    int h[128];
    PEP_HIP(ctx, hipMemcpyAsync(h, ctx->ws[9].p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int l = 0; l < 64; ++l) {
        const int e_shr = l == 0 ? -1 : l - 1, e_shl = l == 63 ? -2 : l + 1;
        if (h[l] != e_shr || h[64 + l] != e_shl) return pep_fail(ctx, PEP_ERR_INTERNAL, "DPP wave shift self-test failed (unexpected lane semantics)");
    }
    return PEP_OK;
}

// Runs K5 over `n` candidate keys.  trace = false: score pass (ws[12] <- score / end cell / a0 per candidate).
// trace = true: same DP plus traceback codes (ws[11] dir_off u64[n+1], ws[13] dirs).  ws[10] nblk, ws[14] scan input.
// d_n (traceback pass, optional): the number of candidates lives on the device and `n` is an upper bound of it (grids and buffers are
// sized from the bound).  dir_blocks_bound (traceback pass, optional): an upper bound of the pass's 16-step blocks - with it the traceback
// area is sized without the host ever seeing the totals (no synchronisation in here; the caller reads the counter block *d_hdr -
// [0] cells, [1] 16-step blocks, [3] long candidates, [5] candidates swept - whenever it synchronises next); 0 = read them here.
int pep_sw_run(pep_ctx *ctx, const uint64_t *d_cands, uint64_t n, bool trace, const int32_t *d_known, const int32_t *d_end_lane, const int32_t *d_skip_mode,
               const uint32_t *d_n, uint64_t dir_blocks_bound, unsigned long long **d_hdr)
{
    const pep_search_params &P = ctx->params;
    if (d_hdr) *d_hdr = nullptr;
    if (n == 0) return PEP_OK;
    PEP_TRY(dev_reserve(ctx, ctx->ws[10], (n + 1) * sizeof(uint32_t)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[11], (n + 2) * sizeof(uint64_t)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[14], (n + 2) * sizeof(uint64_t)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[12], (n + 1) * sizeof(int4)));
    if (n >= (1ull << 32)) return pep_fail(ctx, PEP_ERR_LIMIT, "pep_sw_run: more than 2^32 candidates in one launch");
    // per-wave staging area (u16 per residue; score pass: query + target windows of two candidates, traceback pass: of four candidates'
    // sub-bands), sized for the longest possible pair, capped at 8 KiB: two blocks of eight wavefronts per CU next to their table images
    const uint64_t max_blk = ((uint64_t)ctx->q.max_len + ctx->t.max_len) / 16 + 2;
    const uint64_t want = 2 * 2 * ((8 * max_blk + 80 + 7) & ~7ull);
    const uint64_t want4 = 8 * 2 * ((8 * max_blk + SUB_LANES + 16 + 7) & ~7ull);
    const int pk16 = (P.use_lds && P.reserved[1] == 0 && (!trace || d_known)) ? 1 : 0;       // reserved[1] != 0 forces the 32-bit passes (tests)
    const int lds_res_bytes = P.use_lds ? (int)std::min<uint64_t>(8192, ((pk16 ? (trace ? std::max(want4, 2 * want) : 2 * want) : want) + 255) & ~255ull) : 0;
    // traceback pass: the four-candidate sweep stages 8 windows of 8 * blocks + 48 entries; candidates with more 16-step blocks than fit
    // are swept one per wavefront.  The order below is by decreasing length, so they are a prefix of it.
    uint32_t nb_limit = 0;
    const bool split_long = pk16 && (trace ? want4 : 2 * want) > (uint64_t)lds_res_bytes;
    if (!trace && split_long) {
        // score pass: the longest pair the packed sweep stages at once (four windows of 8 * blocks + 80 entries)
        const int entries = lds_res_bytes / (4 * 2);
        nb_limit = std::min<uint32_t>(entries > 88 ? (uint32_t)((entries - 87) / 8) : 0, LEN_BUCKETS - 2);
    }
    if (trace && pk16) {
        const int entries = lds_res_bytes / (8 * 2);
        nb_limit = entries > SUB_LANES + 16 + 8 ? (uint32_t)((entries - SUB_LANES - 16) / 8 - 1) : 0;     // (the sub-band may need one block more than the band)
        nb_limit = std::min<uint32_t>(nb_limit, LEN_BUCKETS - 2);
    }
    // counter block of this pass (d_zero): [0..63] totals (cells, 16-step blocks, work-queue counters, candidates above nb_limit, candidates in
    // the order), [64..) length histogram, then the scatter cursors; ws[15]: the order itself
    static_assert(PEP_ZERO_SW_BYTES == 64 + 2 * LEN_BUCKETS * sizeof(uint32_t), "counter block of a Smith-Waterman pass");
    void *zb = nullptr;
    PEP_TRY(pep_zero_block(ctx, trace ? PEP_ZC_SW_TRACE : PEP_ZC_SW_SCORE, trace ? PEP_ZERO_SW_TRACE : PEP_ZERO_SW_SCORE, PEP_ZERO_SW_BYTES, &zb));
    PEP_TRY(dev_reserve(ctx, ctx->ws[15], (n + 1) * sizeof(uint32_t)));
    unsigned long long *cells = reinterpret_cast<unsigned long long *>(zb);
    if (d_hdr) *d_hdr = cells;
    uint32_t *len_hist = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(zb) + 64), *cursor = len_hist + LEN_BUCKETS, *order = ctx->ws[15].as<uint32_t>();
    uint64_t *lb_state = nullptr, lb_epoch = 0;
    uint32_t lb_ticket = 0;
    const uint64_t prep_tiles = ceil_div(n + (trace ? 1 : 0), 256);        // (the traceback pass writes one entry more: the total)
    if (trace) PEP_TRY(pep_lookback_begin(ctx, ctx->scan_state[1], prep_tiles, (1u << 14) - 1, &lb_state, &lb_ticket, &lb_epoch));
    // score pass: identical pairs are settled without a sweep (identical_check; params.reserved2 bit 1 switches it off, bit 2 on whatever the
    // size - tests).  Not below 16 k candidates: a pass that small is one or two rounds of wavefronts, a fifth fewer of them does not shorten it,
    // and the check is a launch and a chain of loads of its own (8x1 cell of the benchmark, 6 163 candidates: 0.965 -> 1.0 ms with it)
    const int32_t *skip = trace ? d_skip_mode : nullptr;
    if (!trace && (P.reserved2 & 2) == 0 && (n >= 16384 || (P.reserved2 & 4)) && P.gap_ext >= 0 && P.gap_open + P.gap_ext > 0) {       // (the argument needs gaps that cost something)
        PEP_TRY(dev_reserve(ctx, ctx->ws[13], (n + 1) * sizeof(int32_t)));          // (the traceback codes' buffer: not in use before the traceback pass)
        IdentArgs ia;
        ia.dominant = ia.harmless = 0;
        for (int r = 0; r < 32; ++r) {
            ia.diag[r] = P.sub[r * 32 + r];
            bool dom = P.sub[r * 32 + r] > 0, harm = true;
            for (int x = 0; x < 32; ++x) {
                if (x != r) dom = dom && P.sub[r * 32 + x] < P.sub[r * 32 + r];
                harm = harm && P.sub[r * 32 + x] <= 0;
            }
            if (dom) ia.dominant |= 1u << r;
            if (harm) ia.harmless |= 1u << r;
        }
        hipLaunchKernelGGL(identical_check, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, ctx->stream, d_cands, n, ctx->q.res.as<const uint8_t>(), ctx->q.off.as<const uint32_t>(),
                           ctx->q.len.as<const uint32_t>(), ctx->t.res.as<const uint8_t>(), ctx->t.off.as<const uint32_t>(), ctx->t.len.as<const uint32_t>(), ia,
                           ctx->ws[12].as<int4>(), ctx->ws[13].as<int32_t>());
        skip = ctx->ws[13].as<const int32_t>();
    }
    hipLaunchKernelGGL(sw_prep, dim3((unsigned)prep_tiles), dim3(256), 0, ctx->stream, d_cands, n, trace ? d_n : nullptr, ctx->q.len.as<const uint32_t>(),
                       ctx->t.len.as<const uint32_t>(), ctx->ws[10].as<uint32_t>(), ctx->ws[14].as<uint64_t>(), cells, len_hist, nb_limit, skip, trace ? 0 : 1,
                       ctx->ws[11].as<uint64_t>(), lb_state, lb_ticket, lb_epoch);
    hipLaunchKernelGGL(sw_order, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, ctx->ws[10].as<const uint32_t>(), n, trace ? d_n : nullptr,
                       (const uint32_t *)len_hist, cursor, order, skip);
    if (trace) {
        uint64_t total_blk = dir_blocks_bound;
        if (!dir_blocks_bound) {
            // the traceback area is sized from the block total: the host has to see it before the launch
            unsigned long long h_tot[6] = {0, 0, 0, 0, 0, 0};      // cells, 16-step blocks, (work-queue counter), candidates above nb_limit, (queue), candidates in the order: one copy
            PEP_TRY(pep_read_back(ctx, h_tot, cells, sizeof(h_tot)));
            PEP_TRY(pep_sync_reads(ctx));
            total_blk = h_tot[1];
            ctx->trace_swept = h_tot[5];
            ctx->stats.cells_trace += h_tot[0];
            ctx->stats.cells_swept_trace += total_blk * 16 * 64;
            ctx->stats.dir_bytes += total_blk * 512;
        }
        PEP_TRY(dev_reserve(ctx, ctx->ws[13], total_blk * 512 + 512));
        PEP_TRY(dev_reserve(ctx, ctx->d_trace_mode, (n + 1) * sizeof(int32_t)));
        PEP_TRY(dev_reserve(ctx, ctx->d_trace_defer, (n + 1) * sizeof(uint32_t)));
    }

    SwArgs a;
    a.cands = d_cands; a.n = n;
    a.q_res = ctx->q.res.as<const uint8_t>(); a.t_res = ctx->t.res.as<const uint8_t>();
    a.q_off = ctx->q.off.as<const uint32_t>(); a.q_len = ctx->q.len.as<const uint32_t>();
    a.t_off = ctx->t.off.as<const uint32_t>(); a.t_len = ctx->t.len.as<const uint32_t>();
    a.sub_image = ctx->sub_lds.as<const uint32_t>();
    a.dir_off = ctx->ws[11].as<const uint64_t>(); a.nblk = ctx->ws[10].as<const uint32_t>();
    a.dirs = trace ? ctx->ws[13].as<uint32_t>() : nullptr; a.out = ctx->ws[12].as<int4>();
    a.oe = P.gap_open + P.gap_ext; a.ext = P.gap_ext;
    a.pk16 = pk16;
    a.known = trace ? d_known : nullptr;
    a.end_lane = trace ? d_end_lane : nullptr;
    a.mode = trace ? ctx->d_trace_mode.as<int32_t>() : nullptr;
    a.defer = trace ? ctx->d_trace_defer.as<uint32_t>() : nullptr;
    a.counts = cells;
    // (windows of the longest possible pair against the staging area: score pass 4 windows of 8 * blocks + 80 entries, traceback pass 8 of 8 * blocks + 48)
    a.split_long = split_long ? 1 : 0;
    a.order = order;
    a.max_sub = 1;
    for (int x = 0; x < 32 * 32; ++x) a.max_sub = std::max(a.max_sub, (int)P.sub[x]);
    a.lds_res_bytes = lds_res_bytes;
    const size_t smem = LDS_TABLE_BYTES + (size_t)WAVES_PER_BLOCK * a.lds_res_bytes;
    // The score launch of the LONG candidates (the packed sweep in chunks of blocks: nucleotide alignments, long proteins) takes a SMALL staging area: a
    // chunk is restaged every few hundred steps whatever its size, and 3 KiB per wavefront put four blocks - eight wavefronts per SIMD, which its
    // 36 VGPRs allow - on a CU where the 6.5 KiB of the short pairs' launch put two: 1.09 -> 1.01 ms on the nucleotide tool's 10 000-gene search.  (The same
    // for the traceback launch made it slower, 1.98 -> 2.37 ms: what held that kernel back was not occupancy - see the deferred full-band pairs there.)
    SwArgs al = a;
    al.lds_res_bytes = std::min(lds_res_bytes, 3072);
    const size_t smem_long = LDS_TABLE_BYTES + (size_t)WAVES_PER_BLOCK * al.lds_res_bytes;
    pep_timer_begin(ctx, trace ? TM_SW_TRACE : TM_SW);
    if (trace) {
        // resident blocks only: every wavefront pulls its next item from a counter, first the candidates that are too long for the
        // four-candidate sweep (one per wavefront), then the rest; the counts are read from the pass's counter block on the device.
        // 160 KB of LDS and four wavefronts per SIMD (the kernel's launch bound) per CU
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device);
        const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(16 / WAVES_PER_BLOCK, (160 * 1024) / std::max<size_t>(smem, 1)));
        const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(n, WAVES_PER_BLOCK), (uint64_t)n_cu * per_cu);        // (one candidate per item at worst)
        // candidates with more blocks than the staging area takes at once exist only when the sequences are long enough: then they get a launch
        // of their own (the packed sweep in chunks), in front - they are the longest work
        if (a.split_long) hipLaunchKernelGGL(sw_trace_kernel<true>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), smem, ctx->stream, a);
        hipLaunchKernelGGL(sw_trace_kernel<false>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), smem, ctx->stream, a);
        // (the pairs that left their sub-band, in the full band: blocks beyond the list's length leave at once)
        hipLaunchKernelGGL(sw_trace_retry_kernel, dim3(grid), dim3(64 * WAVES_PER_BLOCK), smem, ctx->stream, a);
    } else {
        // one item (a packed candidate pair, or one candidate) per wavefront: the hardware's block dispatcher balances the load better than a
        // grid-stride loop inside fewer blocks (score pass 0.82 -> 0.80 ms on the benchmark), and the 16 KiB table load per block comes out
        // of the L2
        const uint64_t items = ceil_div(a.pk16 ? (n + 1) / 2 : n, WAVES_PER_BLOCK);
        const unsigned grid = (unsigned)std::min<uint64_t>(items, 256ull * 256);
        if (a.split_long) hipLaunchKernelGGL(sw_score_kernel<true>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), smem_long, ctx->stream, al);
        hipLaunchKernelGGL(sw_score_kernel<false>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), smem, ctx->stream, a);
    }
    pep_timer_end(ctx, trace ? TM_SW_TRACE : TM_SW);
    PEP_HIP(ctx, hipGetLastError());
    ctx->stats.sw_launches += 1;
    return PEP_OK;
}
