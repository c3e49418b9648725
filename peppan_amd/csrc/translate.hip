// K1: nucleotide -> protein on the GPU.
//   query side : three forward frames per gene, keep the frame with the fewest stop/ambiguous-separated
//                segments (ties -> lowest frame)                      [replaces uberBlast.py:525-529]
//   reference  : 6 (or 3) frames per sequence, each frame string cut after the first stop at or beyond
//                1000 residues from the chunk start                    [replaces uberBlast.py:535-544]
//   both       : residues written as codes (letter - 'A') into the padded packed layout that the seed
//                and Smith-Waterman kernels read (16-byte aligned sequence starts, >= 16 pad bytes of
//                code 31 between sequences, 64 at both ends).          [translation: configure.py:160-194]
// Byte-granular, HBM-bound: 3 nt bytes read per residue byte written.
#include "common.h"
#include <algorithm>

namespace {

// codon index a<<4 | b<<2 | c with A0 C1 G2 T3 -> residue code; stops are X (23), as in the reference's table
__constant__ uint8_t c_codon[2][64];

__device__ __forceinline__ int base2(uint8_t ch)
{
    switch (ch & 0xDF) {              // fold case
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return ch == '-' ? -2 : -1;
    }
}

// residue `aa` of frame `frame` (1..6) of the sequence nt[0..L): returns code 0..25, X for ambiguous/partial codons;
// *is_gap set when the codon contains '-' (the reference emits '-', which is not an 'X' for frame choice)
__device__ __forceinline__ int translate_at(const uint8_t *__restrict__ nt, int64_t L, int frame, int64_t aa, int tab, bool *is_gap)
{
    int b[3];
    const int64_t p0 = (frame <= 3 ? frame - 1 : frame - 4) + 3 * aa;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t p = p0 + k;
        int v = -1;
        if (p < L) {
            if (frame <= 3) v = base2(nt[p]);
            else { v = base2(nt[L - 1 - p]); if (v >= 0) v = 3 - v; }
        }
        b[k] = v;
    }
    const bool gap = (b[0] == -2) | (b[1] == -2) | (b[2] == -2);
    if (is_gap) *is_gap = gap;
    if (gap || (b[0] | b[1] | b[2]) < 0) return 23;
    return c_codon[tab][(b[0] << 4) | (b[1] << 2) | b[2]];
}

__device__ __forceinline__ int64_t frame_len(int64_t L, int frame)
{
    const int64_t rem = L - (frame <= 3 ? frame - 1 : frame - 4);
    return rem > 0 ? (rem + 2) / 3 : 0;
}

// one wavefront per query gene
__global__ __launch_bounds__(256) void k1_query_frames(const uint8_t *__restrict__ nt, const uint64_t *__restrict__ off, uint32_t n, int tab,
                                                       uint32_t *__restrict__ frame_out, uint32_t *__restrict__ len_out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n) return;
    const uint8_t *s = nt + off[g];
    const int64_t L = (int64_t)(off[g + 1] - off[g]);
    uint32_t best_cnt = 0xFFFFFFFFu, best_f = 1, best_len = 0;
    for (int f = 1; f <= 3; ++f) {
        const int64_t na = frame_len(L, f);
        uint32_t x = 0;
        for (int64_t a = lane; a + 1 < na; a += 64) {       // s[:-1]
            bool gap;
            const int c = translate_at(s, L, f, a, tab, &gap);
            x += (c == 23 && !gap) ? 1u : 0u;
        }
        for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
        if (x < best_cnt) { best_cnt = x; best_f = (uint32_t)f; best_len = (uint32_t)na; }
    }
    if (lane == 0) { frame_out[g] = best_f; len_out[g] = best_len; }
}

// one wavefront per (reference sequence, frame): chunk boundaries.  chunk_base[w] = first slot of this frame's chunk list.
__global__ __launch_bounds__(256) void k1_ref_chunks(const uint8_t *__restrict__ nt, const uint64_t *__restrict__ off, uint32_t n, int n_frames, int tab,
                                                     const uint64_t *__restrict__ chunk_base, uint32_t *__restrict__ chunk_cnt,
                                                     uint32_t *__restrict__ chunk_off, uint32_t *__restrict__ chunk_len)
{
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (uint64_t)n * n_frames) return;
    const uint32_t g = (uint32_t)(w / n_frames);
    const int f = (int)(w % n_frames) + 1;
    const uint8_t *s = nt + off[g];
    const int64_t L = (int64_t)(off[g + 1] - off[g]);
    const int64_t na = frame_len(L, f);         // the frame string; the reference appends one more 'X' at index na
    const uint64_t base = chunk_base[w];
    uint32_t cnt = 0;
    int64_t c0 = 0;
    while (c0 < na + 1) {
        int64_t end;                             // index of the chunk's last character in s + 'X'
        if (na + 1 - c0 >= 1001) {
            int64_t x0 = c0 + 1000;
            end = -1;
            while (end < 0) {
                const int64_t a = x0 + lane;
                bool isx = false;
                if (a < na) { bool gap; isx = (translate_at(s, L, f, a, tab, &gap) == 23) && !gap; }
                else if (a == na) isx = true;
                const unsigned long long m = __ballot(isx);
                if (m) end = x0 + (int64_t)__ffsll((long long)m) - 1; else x0 += 64;
            }
        } else end = na;
        int64_t len = end - c0 + 1;
        if (end == na) --len;                    // drop the appended 'X'
        if (len > 0) {
            if (lane == 0) { chunk_off[base + cnt] = (uint32_t)c0; chunk_len[base + cnt] = (uint32_t)len; }
            ++cnt;
        }
        c0 = end + 1;
    }
    if (lane == 0) chunk_cnt[w] = cnt;
}

struct PackDesc {          // one per packed sequence: where its residues come from
    uint32_t seq;
    uint32_t frame;
    uint32_t aa_off;
    uint32_t len;
};

__device__ __forceinline__ uint32_t find_seq(const uint32_t *__restrict__ off, uint32_t n, uint32_t p)
{
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// one thread per 16-byte block of the packed layout: sequence starts are 16-aligned and separated by >= 16 padding
// bytes, so a block belongs to at most one sequence -> one owner search and one 16-byte store per thread
__global__ __launch_bounds__(256) void k1_pack(const uint8_t *__restrict__ nt, const uint64_t *__restrict__ nt_off, int tab,
                                               const PackDesc *__restrict__ desc, const uint32_t *__restrict__ pk_off, uint32_t n_packed,
                                               uint8_t *__restrict__ res, uint32_t *__restrict__ blk2seq, uint64_t total)
{
    const uint64_t blk = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t p0 = blk * 16;
    if (p0 >= total) return;
    uint32_t w[4] = {0x1F1F1F1Fu, 0x1F1F1F1Fu, 0x1F1F1F1Fu, 0x1F1F1F1Fu};      // PEP_PAD_CODE x 16
    if (n_packed && p0 >= pk_off[0]) {
        const uint32_t s = find_seq(pk_off, n_packed, (uint32_t)p0);
        blk2seq[blk] = s;
        const PackDesc d = desc[s];
        const uint32_t x0 = (uint32_t)p0 - pk_off[s];
        if (x0 < d.len) {
            const uint8_t *src = nt + nt_off[d.seq];
            const int64_t L = (int64_t)(nt_off[d.seq + 1] - nt_off[d.seq]);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (x0 + k < d.len) {
                    const uint32_t c = (uint32_t)translate_at(src, L, (int)d.frame, (int64_t)d.aa_off + x0 + k, tab, nullptr);
                    w[k >> 2] = (w[k >> 2] & ~(0xFFu << ((k & 3) * 8))) | (c << ((k & 3) * 8));
                }
            }
        }
    }
    if (p0 + 16 <= total) *reinterpret_cast<uint4 *>(res + p0) = make_uint4(w[0], w[1], w[2], w[3]);
    else for (uint64_t k = 0; p0 + k < total; ++k) res[p0 + k] = (uint8_t)(w[k >> 2] >> ((k & 3) * 8));
}

void fill_codon_table(uint8_t tab[2][64])
{
    static const char *aa = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVXYXYSSSSXCWCLFLF";
    for (int i = 0; i < 64; ++i) tab[0][i] = tab[1][i] = (uint8_t)(aa[i] - 'A');
    tab[1][56] = (uint8_t)('W' - 'A');          // translation table 4: TGA -> W
}

int upload_codon_table(pep_ctx *ctx)
{
    uint8_t tab[2][64];
    fill_codon_table(tab);
    PEP_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(c_codon), tab, sizeof(tab), 0, hipMemcpyHostToDevice, ctx->stream));
    return PEP_OK;
}

// builds the padded layout (host prefix sums) and launches k1_pack
int pack_from_desc(pep_ctx *ctx, const NtSet &nt, int tab, const std::vector<PackDesc> &desc, SeqSet &out, DevBuf &d_desc)
{
    const uint32_t n = (uint32_t)desc.size();
    out.n = n;
    out.h_off.assign(n + 1, 0);
    out.h_len.assign(n, 0);
    uint64_t pos = PEP_END_PAD, residues = 0;
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        out.h_off[i] = (uint32_t)pos;
        out.h_len[i] = desc[i].len;
        residues += desc[i].len;
        max_len = std::max(max_len, desc[i].len);
        pos += ((uint64_t)desc[i].len + 15) / 16 * 16 + PEP_SEQ_GAP;
        if (pos > PEP_MAX_RESIDUES) return pep_fail(ctx, PEP_ERR_LIMIT, "packed protein set exceeds 2^29 bytes");
    }
    pos += PEP_END_PAD;
    out.h_off[n] = (uint32_t)pos;
    out.total = pos; out.residues = residues; out.max_len = max_len;
    if (max_len > PEP_MAX_SEQ_LEN) return pep_fail(ctx, PEP_ERR_LIMIT, "protein longer than PEP_MAX_SEQ_LEN");
    PEP_TRY(dev_reserve(ctx, out.res, pos + 64));
    PEP_TRY(dev_reserve(ctx, out.off, (size_t)(n + 1) * 4));
    PEP_TRY(dev_reserve(ctx, out.len, (size_t)(n + 1) * 4));
    PEP_TRY(dev_reserve(ctx, d_desc, (size_t)(n + 1) * sizeof(PackDesc)));
    PEP_TRY(dev_reserve(ctx, out.blk2seq, (pos / 16 + 2) * 4));
    PEP_HIP(ctx, hipMemcpyAsync(out.off.p, out.h_off.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n) {
        PEP_HIP(ctx, hipMemcpyAsync(out.len.p, out.h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        PEP_HIP(ctx, hipMemcpyAsync(d_desc.p, desc.data(), (size_t)n * sizeof(PackDesc), hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(k1_pack, dim3((unsigned)ceil_div(ceil_div(pos, 16), 256)), dim3(256), 0, ctx->stream, nt.nt.as<const uint8_t>(), nt.off.as<const uint64_t>(), tab,
                       d_desc.as<const PackDesc>(), out.off.as<const uint32_t>(), n, out.res.as<uint8_t>(), out.blk2seq.as<uint32_t>(), pos);
    PEP_HIP(ctx, hipGetLastError());
    // the host vectors must outlive the async copies
    PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PEP_OK;
}

}  // namespace

int pep_k1_query(pep_ctx *ctx, int gtable)
{
    const NtSet &nt = ctx->q_nt;
    const int tab = gtable == 4 ? 1 : 0;
    PEP_TRY(upload_codon_table(ctx));
    const uint32_t n = nt.n;
    if (n > PEP_MAX_QUERIES) return pep_fail(ctx, PEP_ERR_LIMIT, "too many queries");
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], (size_t)(n + 1) * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[1], (size_t)(n + 1) * 4));
    std::vector<uint32_t> frame(n), len(n);
    if (n) {
        hipLaunchKernelGGL(k1_query_frames, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, ctx->stream, nt.nt.as<const uint8_t>(), nt.off.as<const uint64_t>(), n, tab,
                           ctx->ws[0].as<uint32_t>(), ctx->ws[1].as<uint32_t>());
        PEP_HIP(ctx, hipGetLastError());
        PEP_HIP(ctx, hipMemcpyAsync(frame.data(), ctx->ws[0].p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, hipMemcpyAsync(len.data(), ctx->ws[1].p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    std::vector<PackDesc> desc(n);
    ctx->q_meta.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        desc[i] = PackDesc{i, frame[i], 0u, len[i]};
        ctx->q_meta[i] = pep_query_meta{i, frame[i], len[i], (uint32_t)(nt.h_off[i + 1] - nt.h_off[i])};
    }
    return pack_from_desc(ctx, nt, tab, desc, ctx->q, ctx->ws[2]);
}

int pep_k1_ref(pep_ctx *ctx, int frames, int gtable)
{
    const NtSet &nt = ctx->r_nt;
    const int tab = gtable == 4 ? 1 : 0;
    PEP_TRY(upload_codon_table(ctx));
    const uint32_t n = nt.n;
    const int nf = frames == 3 ? 3 : 6;
    const uint64_t nw = (uint64_t)n * nf;
    // upper bound of chunks per frame: one per started 1001 residues of (frame string + 'X')
    std::vector<uint64_t> base(nw + 1, 0);
    for (uint32_t g = 0; g < n; ++g) {
        const uint64_t L = nt.h_off[g + 1] - nt.h_off[g];
        for (int f = 1; f <= nf; ++f) {
            const uint64_t shift = (uint64_t)(f <= 3 ? f - 1 : f - 4);
            const uint64_t na = L > shift ? (L - shift + 2) / 3 : 0;
            base[(uint64_t)g * nf + f] = base[(uint64_t)g * nf + f - 1] + (na + 1) / 1001 + 1;
        }
    }
    const uint64_t slots = base[nw];
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], (nw + 1) * 8));
    PEP_TRY(dev_reserve(ctx, ctx->ws[1], (nw + 1) * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[2], (slots + 1) * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[3], (slots + 1) * 4));
    std::vector<uint32_t> cnt(nw), coff(slots), clen(slots);
    if (nw) {
        PEP_HIP(ctx, hipMemcpyAsync(ctx->ws[0].p, base.data(), (nw + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k1_ref_chunks, dim3((unsigned)ceil_div(nw, 4)), dim3(256), 0, ctx->stream, nt.nt.as<const uint8_t>(), nt.off.as<const uint64_t>(), n, nf, tab,
                           ctx->ws[0].as<const uint64_t>(), ctx->ws[1].as<uint32_t>(), ctx->ws[2].as<uint32_t>(), ctx->ws[3].as<uint32_t>());
        PEP_HIP(ctx, hipGetLastError());
        PEP_HIP(ctx, hipMemcpyAsync(cnt.data(), ctx->ws[1].p, nw * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, hipMemcpyAsync(coff.data(), ctx->ws[2].p, slots * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, hipMemcpyAsync(clen.data(), ctx->ws[3].p, slots * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    std::vector<PackDesc> desc;
    desc.reserve(nw);
    ctx->t_meta.clear();
    for (uint64_t w = 0; w < nw; ++w) {
        const uint32_t g = (uint32_t)(w / nf), f = (uint32_t)(w % nf) + 1;
        for (uint32_t c = 0; c < cnt[w]; ++c) {
            const uint64_t k = base[w] + c;
            desc.push_back(PackDesc{g, f, coff[k], clen[k]});
            ctx->t_meta.push_back(pep_target_meta{g, f, coff[k], clen[k]});
        }
    }
    if (desc.size() > PEP_MAX_TARGETS) return pep_fail(ctx, PEP_ERR_LIMIT, "too many targets");
    return pack_from_desc(ctx, nt, tab, desc, ctx->t, ctx->ws[4]);
}
