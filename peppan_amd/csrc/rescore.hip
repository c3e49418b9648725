// K7: nucleotide rescoring counts (mode 1 of the reference's cigar2score, uberBlast.py:226-249, called from
// RunBlast.reScore uberBlast.py:397-415).  One wavefront per hit walks the nt CIGAR; the 64 lanes stride over the
// columns of every M run comparing encoded bases (A0 C1 G3 T4 other 2, uberBlast.py:270-271; a reverse-strand hit
// reads the reference backwards as 4 - code, uberBlast.py:412).  Integer outputs only: the float identity / score
// and numpy's round-half-even are applied on the host in float64 exactly as the reference does.
// Scan of 2 x aligned length bytes per hit; the sequences of a search (tens of MB) stay in the L2 / Infinity Cache, so what bounds it is
// the latency of the byte loads, not HBM bandwidth.  Tried in round 2 and dropped: 16 columns per lane and trip through unaligned 16-byte
// loads (+ a byte-swapped window for reverse-strand hits) - 2 to 2.7x SLOWER (216 - 290 us instead of 107 us per call on the mapping
// workload of tools/other_kernels.py): the unaligned wide loads are split by the memory pipeline and the per-byte decoding then costs
// more than the 16 short trips of the byte version.
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

__global__ __launch_bounds__(256) void k7_rescore(uint64_t n, const pep_nt_hit *__restrict__ hits, const uint32_t *__restrict__ cigar,
                                                  const uint8_t *__restrict__ q_nt, const uint64_t *__restrict__ q_off,
                                                  const uint8_t *__restrict__ r_nt, const uint64_t *__restrict__ r_off, long long *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t h = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= n) return;
    const pep_nt_hit hit = hits[h];
    const uint8_t *q = q_nt + q_off[hit.q], *r = r_nt + r_off[hit.r];
    const bool rev = hit.rs > hit.re;
    long long qi = (long long)hit.qs - 1, ri = (long long)hit.rs - 1;
    long long nmatch = 0, ncol = 0, ngap = 0, bgap = 0, mgap = 0;
    const uint32_t *cg = cigar + hit.cigar_off;
    for (uint32_t k = 0; k < hit.cigar_runs; ++k) {
        const uint32_t run = cg[k];
        const long long len = run >> 2;
        const uint32_t op = run & 3u;
        if (op == 0) {
            for (long long x = lane; x < len; x += 64) {
                const int a = enc(q[qi + x]);
                const int b = rev ? 4 - enc(r[ri - x]) : enc(r[ri + x]);
                nmatch += (a == b) ? 1 : 0;
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

// K7 over the hits of a search where they lie - the device copy of the table the search has just emitted (pep_set_nt_match).  The table row a hit becomes is a
// function of the hit and of K1's descriptors of its two packed sequences (pep_table_from_hits: parseDiamond's / parseBlast's coordinate algebra, uberBlast.py:25-58,
// 275-290), so the walk can start from the hit itself: no table is uploaded again, no second round trip, and of K7's five counts only this one needs the sequences -
// the gap counts are functions of the CIGAR alone and are taken by the host while it builds the table.
//   TOOL 0, translated search: CIGAR runs count residues (x 3), the query's frame and the target's (sequence, frame, chunk offset) give the nucleotide coordinates
//   TOOL 1, nucleotide search: runs count bases, a target is a strand of its sequence
template <int TOOL, typename DESC>
__global__ __launch_bounds__(256) void k7_hits(uint64_t n_bound, const uint32_t *__restrict__ d_n_hits, const pep_hit *__restrict__ hits, const uint32_t *__restrict__ cigar,
                                               const DESC *__restrict__ q_desc, const DESC *__restrict__ t_desc,
                                               const uint8_t *__restrict__ q_nt, const uint64_t *__restrict__ q_off,
                                               const uint8_t *__restrict__ r_nt, const uint64_t *__restrict__ r_off, uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t h = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t n = d_n_hits ? min((uint64_t)*d_n_hits, n_bound) : n_bound;
    if (h >= n) return;
    const pep_hit hit = hits[h];
    long long qi, ri;
    bool rev;
    uint32_t r_seq;
    if (TOOL == 0) {
        const PackDesc dq = reinterpret_cast<const PackDesc *>(q_desc)[hit.q], dt = reinterpret_cast<const PackDesc *>(t_desc)[hit.t];
        r_seq = dt.seq;
        const long long rl = (long long)(r_off[r_seq + 1] - r_off[r_seq]), rf = dt.frame, rs_aa = (long long)hit.t_start + dt.aa_off;
        rev = rf > 3;
        qi = (long long)hit.q_start * 3 + dq.frame - 3 - 1;
        ri = (rev ? rl - (rs_aa * 3 + rf - 6) + 1 : rs_aa * 3 + rf - 3) - 1;
    } else {
        const NuclDesc dt = reinterpret_cast<const NuclDesc *>(t_desc)[hit.t];
        r_seq = dt.seq;
        const long long sl = (long long)(r_off[r_seq + 1] - r_off[r_seq]);
        rev = dt.rev != 0;
        qi = (long long)hit.q_start - 1;
        ri = (rev ? sl - (long long)hit.t_start + 1 : (long long)hit.t_start) - 1;
    }
    const uint8_t *q = q_nt + q_off[hit.q], *r = r_nt + r_off[r_seq];
    uint32_t nmatch = 0;
    const uint32_t *cg = cigar + hit.cigar_off;
    for (uint32_t k = 0; k < hit.cigar_runs; ++k) {
        const uint32_t run = cg[k];
        const long long len = (long long)(run >> 2) * (TOOL == 0 ? 3 : 1);
        const uint32_t op = run & 3u;
        if (op == 0) {
            for (long long x = lane; x < len; x += 64) {
                const int a = enc(q[qi + x]);
                const int b = rev ? 4 - enc(r[ri - x]) : enc(r[ri + x]);
                nmatch += (a == b) ? 1u : 0u;
            }
            qi += len; ri += rev ? -len : len;
        } else if (op == 1) qi += len;
        else ri += rev ? -len : len;
    }
    for (int d = 32; d > 0; d >>= 1) nmatch += __shfl_xor(nmatch, d, 64);
    if (lane == 0) out[h] = nmatch;
}

}  // namespace

// queued behind a search on its stream: counts for hits [0, n) - or [0, *d_n_hits) with n as the bound when the count is still on the device - into
// ctx->pin_nt_match; the caller waits for the stream.  The packed sets must come from the context's nucleotide sets (K1 or pep_use_nt_as_residues).
int pep_k7_hits_queue(pep_ctx *ctx, uint64_t n, const pep_hit *d_hits, const uint32_t *d_cigar, const uint32_t *d_n_hits)
{
    if (n == 0) return PEP_OK;
    if (!ctx->q_nt.nt.p || !ctx->r_nt.nt.p) return pep_fail(ctx, PEP_ERR_STATE, "pep_set_nt_match needs pep_set_query_nt and pep_set_ref_nt first");
    const bool nucl = ctx->resid_from_nucl;
    if (!nucl && !(ctx->q_from_nt && ctx->t_from_nt))
        return pep_fail(ctx, PEP_ERR_STATE, "pep_set_nt_match: the packed sets of this search were not made from the context's nucleotide sets");
    if (nucl ? (!ctx->nucl_q.d_desc.p || !ctx->nucl_t.d_desc.p) : (!ctx->d_k1_desc_q.p || !ctx->d_k1_desc_t.p))
        return pep_fail(ctx, PEP_ERR_STATE, "pep_set_nt_match: no descriptors of the packed sets on the device");
    PEP_TRY(dev_reserve(ctx, ctx->d_nt_match, n * 4));
    PEP_TRY(pin_reserve(ctx, ctx->pin_nt_match, n * 4));
    const dim3 grid((unsigned)ceil_div(n, 4)), block(256);
    uint32_t *out = ctx->d_nt_match.as<uint32_t>();
    if (nucl)
        hipLaunchKernelGGL((k7_hits<1, NuclDesc>), grid, block, 0, ctx->stream, n, d_n_hits, d_hits, d_cigar, ctx->nucl_q.d_desc.as<const NuclDesc>(), ctx->nucl_t.d_desc.as<const NuclDesc>(),
                           ctx->q_nt.nt.as<const uint8_t>(), ctx->q_nt.off.as<const uint64_t>(), ctx->r_nt.nt.as<const uint8_t>(), ctx->r_nt.off.as<const uint64_t>(), out);
    else
        hipLaunchKernelGGL((k7_hits<0, PackDesc>), grid, block, 0, ctx->stream, n, d_n_hits, d_hits, d_cigar, ctx->d_k1_desc_q.as<const PackDesc>(), ctx->d_k1_desc_t.as<const PackDesc>(),
                           ctx->q_nt.nt.as<const uint8_t>(), ctx->q_nt.off.as<const uint64_t>(), ctx->r_nt.nt.as<const uint8_t>(), ctx->r_nt.off.as<const uint64_t>(), out);
    PEP_HIP(ctx, hipGetLastError());
    PEP_HIP(ctx, hipMemcpyAsync(ctx->pin_nt_match.p, out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    return PEP_OK;
}

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
    PEP_TRY(pep_h2d(ctx, ctx->ws[0].p, h_hits, n * sizeof(pep_nt_hit)));
    PEP_TRY(pep_h2d(ctx, ctx->ws[1].p, h_cigar, n_cigar * 4));
    hipLaunchKernelGGL(k7_rescore, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, ctx->stream, n, ctx->ws[0].as<const pep_nt_hit>(), ctx->ws[1].as<const uint32_t>(),
                       ctx->q_nt.nt.as<const uint8_t>(), ctx->q_nt.off.as<const uint64_t>(), ctx->r_nt.nt.as<const uint8_t>(), ctx->r_nt.off.as<const uint64_t>(),
                       ctx->ws[2].as<long long>());
    PEP_HIP(ctx, hipGetLastError());
    PEP_TRY(pep_d2h_queue(ctx, h_out, ctx->ws[2].p, n * 5 * 8));
    PEP_HIP(ctx, pep_stream_wait(ctx));
    pep_d2h_finish(ctx);
    return PEP_OK;
}
