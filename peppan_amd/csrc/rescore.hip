// K7: nucleotide rescoring counts (mode 1 of the reference's cigar2score, uberBlast.py:226-249, called from
// RunBlast.reScore uberBlast.py:397-415).  One wavefront per hit walks the nt CIGAR; the 64 lanes stride over the
// columns of every M run (16 per lane and trip) comparing encoded bases (A0 C1 G3 T4 other 2, uberBlast.py:270-271; a reverse-strand hit
// reads the reference backwards as 4 - code, uberBlast.py:412).  Integer outputs only: the float identity / score
// and numpy's round-half-even are applied on the host in float64 exactly as the reference does.
// HBM-bound scan: 2 x aligned length bytes read per hit.
#include "common.h"

namespace {

__device__ __forceinline__ int enc(uint8_t ch)
{
    switch (ch & 0xDF) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 3;
        case 'T': return 4;
        default: return 2;
    }
}

// One wavefront per hit.  Inside an M run every lane takes 16 columns per trip (one unaligned 16-byte load from each sequence: a
// wavefront covers 1 024 columns per trip - most M runs in one), reverse-strand hits read their 16 bytes from the other end;
// byte loads serve the last columns of a run and anything within 16 bytes of the ends of the concatenations.
__global__ __launch_bounds__(256) void k7_rescore(uint64_t n, const pep_nt_hit *__restrict__ hits, const uint32_t *__restrict__ cigar,
                                                  const uint8_t *__restrict__ q_nt, const uint64_t *__restrict__ q_off, uint64_t q_total,
                                                  const uint8_t *__restrict__ r_nt, const uint64_t *__restrict__ r_off, uint64_t r_total,
                                                  long long *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t h = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= n) return;
    const pep_nt_hit hit = hits[h];
    const long long qb = (long long)q_off[hit.q], rb = (long long)r_off[hit.r];        // positions inside the concatenations
    const bool rev = hit.rs > hit.re;
    long long qi = qb + (long long)hit.qs - 1, ri = rb + (long long)hit.rs - 1;
    long long nmatch = 0, ncol = 0, ngap = 0, bgap = 0, mgap = 0;
    const uint32_t *cg = cigar + hit.cigar_off;
    for (uint32_t k = 0; k < hit.cigar_runs; ++k) {
        const uint32_t run = cg[k];
        const long long len = run >> 2;
        const uint32_t op = run & 3u;
        if (op == 0) {
            for (long long x = (long long)lane * 16; x < len; x += 64 * 16) {
                const long long todo = len - x < 16 ? len - x : 16;
                const long long qa = qi + x, ra = rev ? ri - x - 15 : ri + x;              // first byte of the two 16-byte windows
                if (todo == 16 && qa + 16 <= (long long)q_total && ra >= 0 && ra + 16 <= (long long)r_total) {
                    uint32_t qw[4], rl[4], rw[4];
                    __builtin_memcpy(qw, q_nt + qa, 16);
                    __builtin_memcpy(rl, r_nt + ra, 16);
                    // a reverse-strand window is turned round as a whole (byte swap of every dword, dwords in reverse order) so that the
                    // compare loop indexes both windows with compile-time constants - a run-time byte index would push them out of registers
#pragma unroll
                    for (int w = 0; w < 4; ++w) rw[w] = rev ? __builtin_bswap32(rl[3 - w]) : rl[w];
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const int a = enc((uint8_t)(qw[c >> 2] >> ((c & 3) * 8)));
                        const int e = enc((uint8_t)(rw[c >> 2] >> ((c & 3) * 8)));
                        nmatch += (a == (rev ? 4 - e : e)) ? 1 : 0;
                    }
                } else {
                    for (long long c = 0; c < todo; ++c) {
                        const int a = enc(q_nt[qa + c]);
                        const int e = rev ? 4 - enc(r_nt[ri - x - c]) : enc(r_nt[ri + x + c]);
                        nmatch += (a == e) ? 1 : 0;
                    }
                }
            }
            ncol += len;
            qi += len; ri += rev ? -len : len;
        } else {
            ++ngap; bgap += len; if (len > 3) mgap += len;
            if (op == 1) qi += len; else ri += rev ? -len : len;
        }
    }
    for (int d = 32; d > 0; d >>= 1) nmatch += __shfl_xor(nmatch, d, 64);
    if (lane == 0) {
        long long *o = out + h * 5;
        o[0] = nmatch; o[1] = ncol - nmatch; o[2] = ngap; o[3] = bgap; o[4] = mgap;
    }
}

}  // namespace

int pep_k7_rescore(pep_ctx *ctx, uint64_t n, const pep_nt_hit *h_hits, const uint32_t *h_cigar, uint64_t n_cigar, int64_t *h_out)
{
    if (n == 0) return PEP_OK;
    if (!ctx->q_nt.nt.p || !ctx->r_nt.nt.p) return pep_fail(ctx, PEP_ERR_STATE, "pep_rescore_nt needs pep_set_query_nt and pep_set_ref_nt first");
    // validate coordinates on the host so that a bad table is an error, not an out-of-bounds read
    for (uint64_t i = 0; i < n; ++i) {
        const pep_nt_hit &h = h_hits[i];
        if (h.q >= ctx->q_nt.n || h.r >= ctx->r_nt.n || h.cigar_off + h.cigar_runs > n_cigar) return pep_fail(ctx, PEP_ERR_ARG, "pep_rescore_nt: hit index out of range");
        const uint64_t ql = ctx->q_nt.h_off[h.q + 1] - ctx->q_nt.h_off[h.q], rl = ctx->r_nt.h_off[h.r + 1] - ctx->r_nt.h_off[h.r];
        uint64_t qa = 0, ra = 0;
        for (uint32_t k = 0; k < h.cigar_runs; ++k) {
            const uint32_t run = h_cigar[h.cigar_off + k];
            if ((run & 3u) != 2) qa += run >> 2;
            if ((run & 3u) != 1) ra += run >> 2;
        }
        const bool rev = h.rs > h.re;
        const uint64_t rlo = rev ? h.re : h.rs, rhi = rev ? h.rs : h.re;
        if (h.qs < 1 || h.qs - 1 + qa > ql || rlo < 1 || rhi > rl || ra != rhi - rlo + 1)
            return pep_fail(ctx, PEP_ERR_ARG, "pep_rescore_nt: CIGAR inconsistent with the hit coordinates");
    }
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], n * sizeof(pep_nt_hit)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[1], (n_cigar + 1) * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[2], n * 5 * 8));
    PEP_HIP(ctx, hipMemcpyAsync(ctx->ws[0].p, h_hits, n * sizeof(pep_nt_hit), hipMemcpyHostToDevice, ctx->stream));
    PEP_HIP(ctx, hipMemcpyAsync(ctx->ws[1].p, h_cigar, n_cigar * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k7_rescore, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, ctx->stream, n, ctx->ws[0].as<const pep_nt_hit>(), ctx->ws[1].as<const uint32_t>(),
                       ctx->q_nt.nt.as<const uint8_t>(), ctx->q_nt.off.as<const uint64_t>(), ctx->q_nt.total, ctx->r_nt.nt.as<const uint8_t>(), ctx->r_nt.off.as<const uint64_t>(), ctx->r_nt.total,
                       ctx->ws[2].as<long long>());
    PEP_HIP(ctx, hipGetLastError());
    PEP_HIP(ctx, hipMemcpyAsync(h_out, ctx->ws[2].p, n * 5 * 8, hipMemcpyDeviceToHost, ctx->stream));
    PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PEP_OK;
}
