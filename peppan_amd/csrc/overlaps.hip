// K11: pairs of hits whose reference intervals overlap (flag -O) - replaces the numba-compiled sweep tab2overlaps
// (uberBlast.py:73-97) driven by RunBlast.returnOverlap (uberBlast.py:378-395).
// Input rows are sorted by (contig, start, end).  Thread i walks forward from i+1 while the contig is the same and
// start_j <= end_i; a pair is reported when ovl >= min(ovl_l, ovl_p * len_i) or ovl >= ovl_p * len_j (double arithmetic,
// as the reference's float parameters imply).  Count pass -> exclusive scan -> write pass keeps the reference's
// (i ascending, j ascending) output order.  HBM-bound: 16 B per interval read per visited neighbour, 12 B per pair written.
#include "common.h"

namespace {

struct OvlArgs {
    const int32_t *contig;
    const int64_t *start, *end, *rid;
    uint64_t n;
    double ovl_l, ovl_p;
};

template <bool WRITE>
__global__ __launch_bounds__(256) void ovl_sweep(OvlArgs a, uint64_t *__restrict__ cnt, const uint64_t *__restrict__ off, int64_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int32_t c = a.contig[i];
    const int64_t s1 = a.start[i], e1 = a.end[i];
    const double need = fmin(a.ovl_l, a.ovl_p * (double)(e1 - s1 + 1));
    uint64_t k = 0, w = WRITE ? off[i] : 0;
    for (uint64_t j = i + 1; j < a.n; ++j) {
        if (a.contig[j] != c) break;
        const int64_t s2 = a.start[j];
        if (s2 > e1) break;
        const int64_t e2 = a.end[j];
        const int64_t ovl = (e1 < e2 ? e1 : e2) - s2 + 1;
        if ((double)ovl >= need || (double)ovl >= a.ovl_p * (double)(e2 - s2 + 1)) {
            if (WRITE) { out[3 * w] = a.rid[i]; out[3 * w + 1] = a.rid[j]; out[3 * w + 2] = ovl; ++w; }
            ++k;
        }
    }
    if (!WRITE) cnt[i] = k;
}

}  // namespace

// h_out receives up to cap triples (id1, id2, overlap); *n_pairs is always the full count
int pep_k11_overlaps(pep_ctx *ctx, uint64_t n, const int32_t *h_contig, const int64_t *h_start, const int64_t *h_end, const int64_t *h_rid,
                     double ovl_l, double ovl_p, int64_t *h_out, uint64_t cap, uint64_t *n_pairs)
{
    *n_pairs = 0;
    if (n == 0) return PEP_OK;
    for (uint64_t i = 1; i < n; ++i) {
        const bool ok = h_contig[i - 1] < h_contig[i] || (h_contig[i - 1] == h_contig[i] && (h_start[i - 1] < h_start[i] || (h_start[i - 1] == h_start[i] && h_end[i - 1] <= h_end[i])));
        if (!ok) return pep_fail(ctx, PEP_ERR_ARG, "pep_overlaps: rows must be sorted by (contig, start, end)");
    }
    hipStream_t st = ctx->stream;
    DevBuf *W = ctx->ws;
    PEP_TRY(dev_reserve(ctx, W[0], n * 4));
    PEP_TRY(dev_reserve(ctx, W[1], n * 8));
    PEP_TRY(dev_reserve(ctx, W[2], n * 8));
    PEP_TRY(dev_reserve(ctx, W[3], n * 8));
    PEP_TRY(dev_reserve(ctx, W[4], (n + 2) * 8));
    PEP_TRY(dev_reserve(ctx, W[5], (n + 2) * 8));
    PEP_TRY(pep_h2d(ctx, W[0].p, h_contig, n * 4));
    PEP_TRY(pep_h2d(ctx, W[1].p, h_start, n * 8));
    PEP_TRY(pep_h2d(ctx, W[2].p, h_end, n * 8));
    PEP_TRY(pep_h2d(ctx, W[3].p, h_rid, n * 8));
    OvlArgs a;
    a.contig = W[0].as<const int32_t>(); a.start = W[1].as<const int64_t>(); a.end = W[2].as<const int64_t>(); a.rid = W[3].as<const int64_t>();
    a.n = n; a.ovl_l = ovl_l; a.ovl_p = ovl_p;
    const unsigned g = (unsigned)ceil_div(n, 256);
    hipLaunchKernelGGL(ovl_sweep<false>, dim3(g), dim3(256), 0, st, a, W[4].as<uint64_t>(), (const uint64_t *)nullptr, (int64_t *)nullptr);
    PEP_TRY(pep_scan_u64(ctx, W[4].as<uint64_t>(), W[5].as<uint64_t>(), n, W[7]));
    uint64_t total = 0;
    PEP_HIP(ctx, hipMemcpyAsync(&total, W[5].as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, pep_stream_wait(ctx));
    *n_pairs = total;
    if (total == 0 || total > cap) return PEP_OK;          // caller re-calls with a buffer of *n_pairs triples
    PEP_TRY(dev_reserve(ctx, W[6], total * 24));
    hipLaunchKernelGGL(ovl_sweep<true>, dim3(g), dim3(256), 0, st, a, (uint64_t *)nullptr, (const uint64_t *)W[5].as<uint64_t>(), W[6].as<int64_t>());
    PEP_HIP(ctx, hipGetLastError());
    PEP_TRY(pep_d2h_queue(ctx, h_out, W[6].p, total * 24));
    PEP_HIP(ctx, pep_stream_wait(ctx));
    pep_d2h_finish(ctx);
    return PEP_OK;
}
