// K12: aligned-allele strings of the genes->genomes mapping (iter_map_bsn, PEPPAN.py:812-835) and their base-5 packing
// (PEPPAN.py:846-848).  The reference does this per hit in Python: slice the contig, reverse-complement, walk the CIGAR
// string with a regex, join pieces, regex the codons, look every base up.  Here:
//   k12_codes  one wavefront per hit row walks the nt CIGAR; the 64 lanes stride over the columns of each run and write
//              the base codes (A1 C2 G3 T4, anything else / query-only column 0) of the aligned allele; the per-frame
//              M-column counts are wave-uniform scalars.
//   k12_orf    one wavefront per row reads its codes back as codons, 64 at a time; a ballot marks the stop codons and
//              the (rare) set bits are folded into the longest stop-free stretch.
//   k12_pack   one wavefront per gene group overlays the rows of the group on the gene's coordinate system (later rows
//              win, as the reference's slice assignment does) and emits codes[j]*25 + codes[s+j]*5 + codes[2s+j].
// HBM-bound byte work: per row ~1 B read + 1 B written per aligned column, then 1 B read per column and 1/3 B written.
#include "common.h"
#include <algorithm>

namespace {



__device__ __forceinline__ uint32_t base_code(uint8_t ch)
{
    switch (ch) {
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 3;
        case 'T': return 4;
        default: return 0;
    }
}

__global__ __launch_bounds__(256) void k12_codes(uint64_t n, const pep_locus *__restrict__ rows, const uint32_t *__restrict__ cigar,
                                                 const uint8_t *__restrict__ nt, const uint64_t *__restrict__ nt_off,
                                                 const uint64_t *__restrict__ row_off, uint8_t *__restrict__ codes, long long *__restrict__ in_frame)
{
    const int lane = threadIdx.x & 63;
    const uint64_t h = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= n) return;
    const pep_locus L = rows[h];
    const uint8_t *c = nt + nt_off[L.contig];
    const bool rev = L.rs > L.re;
    uint8_t *out = codes + row_off[h];
    long long at = 0, fr0 = 0, fr1 = 0, fr2 = 0;
    int f = 0;
    const uint32_t *cg = cigar + L.cigar_off;
    for (uint32_t k = 0; k < L.cigar_runs; ++k) {
        const uint32_t run = cg[k];
        const long long len = run >> 2;
        const uint32_t op = run & 3u;
        if (op == 0) {
            for (long long x = lane; x < len; x += 64) {
                uint32_t b = rev ? base_code(c[(long long)L.rs - 1 - (at + x)]) : base_code(c[(long long)L.rs - 1 + at + x]);
                if (rev && b) b = 5 - b;
                out[x] = (uint8_t)b;
            }
            out += len; at += len;
            if (f == 0) fr0 += len; else if (f == 1) fr1 += len; else fr2 += len;
        } else if (op == 2) {
            at += len;
            f = (int)(((f - len) % 3 + 3) % 3);
        } else {
            for (long long x = lane; x < len; x += 64) out[x] = 0;
            out += len;
            f = (int)((f + len) % 3);
        }
    }
    if (lane == 0) in_frame[h] = max(fr0, max(fr1, fr2));
}

__global__ __launch_bounds__(256) void k12_orf(uint64_t n, const uint64_t *__restrict__ row_off, const uint8_t *__restrict__ codes, int stop_tga,
                                               long long *__restrict__ orf)
{
    const int lane = threadIdx.x & 63;
    const uint64_t h = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= n) return;
    const uint8_t *ms = codes + row_off[h];
    const long long span = (long long)(row_off[h + 1] - row_off[h]);
    const long long n_codon = span / 3;
    long long prev = 0, longest = 0;
    for (long long base = 0; base < n_codon; base += 64) {
        const long long cd = base + lane;
        bool stop = false;
        if (cd < n_codon) {
            const uint32_t a = ms[3 * cd], b = ms[3 * cd + 1], d = ms[3 * cd + 2];
            stop = a == 4 && ((b == 1 && (d == 1 || d == 3)) || (stop_tga && b == 3 && d == 1));
        }
        unsigned long long mask = __ballot(stop);
        while (mask) {
            const long long pos = (base + __builtin_ctzll(mask)) * 3;
            longest = max(longest, pos - prev);
            prev = pos;
            mask &= mask - 1;
        }
    }
    longest = max(longest, span - prev);
    if (lane == 0) orf[h] = longest;
}

__global__ __launch_bounds__(256) void k12_pack(uint32_t n_groups, const uint64_t *__restrict__ grp_off, const uint32_t *__restrict__ grp_qlen,
                                                const uint64_t *__restrict__ pack_off, const pep_locus *__restrict__ rows,
                                                const uint64_t *__restrict__ row_off, const uint8_t *__restrict__ codes, uint8_t *__restrict__ packed)
{
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const long long ql = grp_qlen[g], s = (ql + 2) / 3;
    const uint64_t r0 = grp_off[g], r1 = grp_off[g + 1];
    uint8_t *out = packed + pack_off[g];
    for (long long j = lane; j < s; j += 64) {
        uint32_t v[3] = {0, 0, 0};
        for (int part = 0; part < 3; ++part) {
            const long long p = part * s + j;
            if (p >= ql) continue;
            for (uint64_t r = r1; r-- > r0;) {                    // the last row that covers p wins
                const long long lo = (long long)rows[r].q_start - 1, len = (long long)(row_off[r + 1] - row_off[r]);
                if (p >= lo && p < lo + len) { v[part] = codes[row_off[r] + (uint64_t)(p - lo)]; break; }
            }
        }
        out[j] = (uint8_t)(v[0] * 25 + v[1] * 5 + v[2]);
    }
}

}  // namespace

int pep_k12_alleles(pep_ctx *ctx, const uint8_t *h_nt, const uint64_t *h_nt_off, uint32_t n_contigs, uint64_t n, const pep_locus *h_rows,
                    const uint32_t *h_cigar, uint64_t n_cigar, uint32_t n_groups, const uint64_t *h_grp_off, const uint32_t *h_grp_qlen,
                    int gtable, int64_t *h_in_frame, int64_t *h_orf, uint8_t *h_packed, uint64_t packed_cap)
{
    if (n_groups == 0) {
        if (n != 0) return pep_fail(ctx, PEP_ERR_ARG, "pep_alleles: rows without groups");
        return PEP_OK;
    }
    if (h_grp_off[0] != 0 || h_grp_off[n_groups] != n) return pep_fail(ctx, PEP_ERR_ARG, "pep_alleles: grp_off must run from 0 to n_rows");
    // validate on the host so that a bad table is an error, not an out-of-bounds access; also lays out the buffers
    std::vector<uint64_t> row_off(n + 1), pack_off((size_t)n_groups + 1);
    row_off[0] = 0; pack_off[0] = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint64_t ql = h_grp_qlen[g];
        if (ql < 3) return pep_fail(ctx, PEP_ERR_ARG, "pep_alleles: gene shorter than one codon");
        if (h_grp_off[g + 1] < h_grp_off[g] || h_grp_off[g + 1] > n) return pep_fail(ctx, PEP_ERR_ARG, "pep_alleles: grp_off not monotonic");
        pack_off[g + 1] = pack_off[g] + (ql + 2) / 3;
        for (uint64_t r = h_grp_off[g]; r < h_grp_off[g + 1]; ++r) {
            const pep_locus &L = h_rows[r];
            if (L.contig >= n_contigs || L.cigar_off + L.cigar_runs > n_cigar) return pep_fail(ctx, PEP_ERR_ARG, "pep_alleles: row index out of range");
            const uint64_t cl = h_nt_off[L.contig + 1] - h_nt_off[L.contig];
            uint64_t span = 0, rcons = 0;
            for (uint32_t k = 0; k < L.cigar_runs; ++k) {
                const uint32_t run = h_cigar[L.cigar_off + k];
                if ((run & 3u) == 3u) return pep_fail(ctx, PEP_ERR_ARG, "pep_alleles: unknown CIGAR op");
                if ((run & 3u) != 2) span += run >> 2;
                if ((run & 3u) != 1) rcons += run >> 2;
            }
            const uint64_t lo = std::min(L.rs, L.re), hi = std::max(L.rs, L.re);
            if (lo < 1 || hi > cl || rcons != hi - lo + 1 || L.q_start < 1 || (uint64_t)L.q_start - 1 + span > ql)
                return pep_fail(ctx, PEP_ERR_ARG, "pep_alleles: CIGAR inconsistent with the row coordinates");
            row_off[r + 1] = row_off[r] + span;
        }
    }
    if (pack_off[n_groups] > packed_cap) return pep_fail(ctx, PEP_ERR_ARG, "pep_alleles: packed buffer too small");
    const uint64_t nt_total = h_nt_off[n_contigs], code_total = row_off[n];
    DevBuf *W = ctx->ws;
    hipStream_t st = ctx->stream;
    PEP_TRY(dev_reserve(ctx, W[0], nt_total + 1));
    PEP_TRY(dev_reserve(ctx, W[1], ((size_t)n_contigs + 1) * 8));
    PEP_TRY(dev_reserve(ctx, W[2], (n + 1) * sizeof(pep_locus)));
    PEP_TRY(dev_reserve(ctx, W[3], (n_cigar + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[4], (n + 1) * 8));
    PEP_TRY(dev_reserve(ctx, W[5], code_total + 1));
    PEP_TRY(dev_reserve(ctx, W[6], (n + 1) * 16));
    PEP_TRY(dev_reserve(ctx, W[7], ((size_t)n_groups + 1) * 8));
    PEP_TRY(dev_reserve(ctx, W[8], ((size_t)n_groups + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[9], ((size_t)n_groups + 1) * 8));
    PEP_TRY(dev_reserve(ctx, W[10], pack_off[n_groups] + 1));
    PEP_TRY(pep_h2d(ctx, W[0].p, h_nt, nt_total));
    PEP_TRY(pep_h2d(ctx, W[1].p, h_nt_off, ((size_t)n_contigs + 1) * 8));
    PEP_TRY(pep_h2d(ctx, W[2].p, h_rows, n * sizeof(pep_locus)));
    PEP_TRY(pep_h2d(ctx, W[3].p, h_cigar, n_cigar * 4));
    PEP_TRY(pep_h2d(ctx, W[4].p, row_off.data(), (n + 1) * 8));        // (22 000 groups of a 50 000-exemplar genome: these tables pass the runtime's staging limit too)
    PEP_TRY(pep_h2d(ctx, W[7].p, h_grp_off, ((size_t)n_groups + 1) * 8));
    PEP_TRY(pep_h2d(ctx, W[8].p, h_grp_qlen, (size_t)n_groups * 4));
    PEP_TRY(pep_h2d(ctx, W[9].p, pack_off.data(), ((size_t)n_groups + 1) * 8));
    long long *d_frame = W[6].as<long long>(), *d_orf = W[6].as<long long>() + n;
    if (n) {
        hipLaunchKernelGGL(k12_codes, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, n, W[2].as<const pep_locus>(), W[3].as<const uint32_t>(),
                           W[0].as<const uint8_t>(), W[1].as<const uint64_t>(), W[4].as<const uint64_t>(), W[5].as<uint8_t>(), d_frame);
        hipLaunchKernelGGL(k12_orf, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, n, W[4].as<const uint64_t>(), W[5].as<const uint8_t>(),
                           gtable != 4 ? 1 : 0, d_orf);
    }
    hipLaunchKernelGGL(k12_pack, dim3((unsigned)ceil_div(n_groups, 4)), dim3(256), 0, st, n_groups, W[7].as<const uint64_t>(), W[8].as<const uint32_t>(),
                       W[9].as<const uint64_t>(), W[2].as<const pep_locus>(), W[4].as<const uint64_t>(), W[5].as<const uint8_t>(), W[10].as<uint8_t>());
    PEP_HIP(ctx, hipGetLastError());
    if (n) {
        PEP_TRY(pep_d2h_queue(ctx, h_in_frame, d_frame, n * 8));
        PEP_TRY(pep_d2h_queue(ctx, h_orf, d_orf, n * 8));
    }
    PEP_TRY(pep_d2h_queue(ctx, h_packed, W[10].p, pack_off[n_groups]));
    PEP_HIP(ctx, pep_stream_wait(ctx));
    pep_d2h_finish(ctx);
    return PEP_OK;
}
