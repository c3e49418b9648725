// K13: exact-duplicate collapse of gene instances (the front end of the clustering path).
//   k13_sha1     SHA-1 of every sequence, the digest PEPPAN keys duplicates and breaks priority ties with
//                (int(hashlib.sha1(seq).hexdigest(), 16), PEPPAN.py:62, 1019).  One thread per sequence: 64-byte blocks arrive as
//                four unaligned 16-byte loads, 80 rounds of 32-bit integer VALU per block.  ~1100 VALU ops per 64 B: compute-bound.
//   k13_insert / k13_lookup   writeGenes (PEPPAN.py:1023-1039) over genes in priority order: a gene is a duplicate of the FIRST gene
//                of the same (length run, digest), where a "length run" is a maximal stretch of equal lengths in priority order -
//                the reference rebuilds its seen-table whenever a new length shows up (PEPPAN.py:1032-1033).  An open-addressing
//                table of gene indices keyed by (run, digest), smallest index wins (atomicMin), makes the result independent of
//                execution order.  Random 4-byte probes + 24-byte key compares: latency-bound, 28 B per gene of algorithmic traffic.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t rol(uint32_t x, int n) { return __builtin_rotateleft32(x, n); }

__device__ void sha1_block(uint32_t h[5], const uint32_t *wi)
{
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = wi[i];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
#pragma unroll
    for (int t = 0; t < 80; ++t) {
        uint32_t x;
        if (t < 16) x = w[t];
        else {
            x = rol(w[(t - 3) & 15] ^ w[(t - 8) & 15] ^ w[(t - 14) & 15] ^ w[t & 15], 1);
            w[t & 15] = x;
        }
        uint32_t f, k;
        if (t < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
        else if (t < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
        else if (t < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
        else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
        const uint32_t tmp = rol(a, 5) + f + e + k + x;
        e = d; d = c; c = rol(b, 30); b = a; a = tmp;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
}

__global__ __launch_bounds__(256) void k13_sha1(uint32_t n, const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint32_t *__restrict__ digest)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *s = bytes + off[i];
    const uint64_t len = off[i + 1] - off[i];
    uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    uint32_t w[16];
    uint64_t at = 0;
    for (; at + 64 <= len; at += 64) {
        __builtin_memcpy(w, s + at, 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = __builtin_bswap32(w[k]);
        sha1_block(h, w);
    }
    // tail: remaining bytes, 0x80, zeros, 64-bit big-endian bit length (one or two blocks)
    const uint32_t rem = (uint32_t)(len - at);
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = 0;
    for (uint32_t k = 0; k < rem; ++k) w[k >> 2] |= (uint32_t)s[at + k] << (24 - 8 * (k & 3));
    w[rem >> 2] |= 0x80u << (24 - 8 * (rem & 3));
    if (rem >= 56) {
        sha1_block(h, w);
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = 0;
    }
    const uint64_t bits = len * 8;
    w[14] = (uint32_t)(bits >> 32);
    w[15] = (uint32_t)bits;
    sha1_block(h, w);
#pragma unroll
    for (int k = 0; k < 5; ++k) digest[(size_t)i * 5 + k] = h[k];
}

__global__ void k13_run_flags(uint32_t n, const uint32_t *__restrict__ len, uint32_t *__restrict__ flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i > 0 && len[i] != len[i - 1]) ? 1u : 0u;
}

struct Key { uint32_t run, d[5]; };

__device__ __forceinline__ Key load_key(const uint32_t *run_excl, const uint32_t *flag, const uint32_t *digest, uint32_t i)
{
    Key k;
    k.run = run_excl[i] + flag[i];          // inclusive scan of the "new length" flags = run id
#pragma unroll
    for (int x = 0; x < 5; ++x) k.d[x] = digest[(size_t)i * 5 + x];
    return k;
}
__device__ __forceinline__ bool same(const Key &a, const Key &b)
{
    return a.run == b.run && a.d[0] == b.d[0] && a.d[1] == b.d[1] && a.d[2] == b.d[2] && a.d[3] == b.d[3] && a.d[4] == b.d[4];
}
__device__ __forceinline__ uint32_t slot_of(const Key &k, uint32_t mask)
{
    uint64_t x = ((uint64_t)k.d[0] << 32 | k.d[1]) ^ ((uint64_t)k.run * 0x9E3779B97F4A7C15ull);
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
    return (uint32_t)x & mask;
}

#define K13_EMPTY 0xFFFFFFFFu

template <bool INSERT>
__global__ void k13_table(uint32_t n, const uint32_t *__restrict__ run_excl, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ digest,
                          uint32_t *__restrict__ slots, uint32_t mask, uint32_t *__restrict__ rep, uint32_t *__restrict__ fail)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Key me = load_key(run_excl, flag, digest, i);
    uint32_t pos = slot_of(me, mask);
    for (uint32_t probe = 0; probe <= mask; ++probe, pos = (pos + 1) & mask) {
        uint32_t cur = INSERT ? atomicCAS(&slots[pos], K13_EMPTY, i) : slots[pos];
        if (cur == K13_EMPTY) {
            if (INSERT) return;                     // claimed
            break;                                   // lookup of a key that was never inserted: cannot happen
        }
        if (cur == i || same(me, load_key(run_excl, flag, digest, cur))) {
            if (INSERT) atomicMin(&slots[pos], i);
            else rep[i] = cur;
            return;
        }
    }
    atomicAdd(fail, 1u);
}

}  // namespace

int pep_k13_sha1(pep_ctx *ctx, const uint8_t *h_bytes, const uint64_t *h_off, uint32_t n, uint8_t *h_digest)
{
    if (n == 0) return PEP_OK;
    DevBuf *W = ctx->ws;
    hipStream_t st = ctx->stream;
    const uint64_t total = h_off[n];
    PEP_TRY(dev_reserve(ctx, W[0], total + 64));
    PEP_TRY(dev_reserve(ctx, W[1], ((size_t)n + 1) * 8));
    PEP_TRY(dev_reserve(ctx, W[2], (size_t)n * 20));
    PEP_HIP(ctx, hipMemcpyAsync(W[0].p, h_bytes, total, hipMemcpyHostToDevice, st));
    PEP_HIP(ctx, hipMemcpyAsync(W[1].p, h_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k13_sha1, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, n, W[0].as<const uint8_t>(), W[1].as<const uint64_t>(), W[2].as<uint32_t>());
    PEP_HIP(ctx, hipGetLastError());
    std::vector<uint32_t> words((size_t)n * 5);
    PEP_HIP(ctx, hipMemcpyAsync(words.data(), W[2].p, (size_t)n * 20, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipStreamSynchronize(st));
    for (size_t k = 0; k < words.size(); ++k) {          // big-endian bytes, as hashlib's digest()
        const uint32_t v = words[k];
        h_digest[4 * k] = (uint8_t)(v >> 24); h_digest[4 * k + 1] = (uint8_t)(v >> 16); h_digest[4 * k + 2] = (uint8_t)(v >> 8); h_digest[4 * k + 3] = (uint8_t)v;
    }
    return PEP_OK;
}

int pep_k13_dedup(pep_ctx *ctx, uint32_t n, const uint32_t *h_len, const uint8_t *h_digest, uint32_t *h_rep)
{
    if (n == 0) return PEP_OK;
    DevBuf *W = ctx->ws;
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> words((size_t)n * 5);
    for (size_t k = 0; k < words.size(); ++k)
        words[k] = (uint32_t)h_digest[4 * k] << 24 | (uint32_t)h_digest[4 * k + 1] << 16 | (uint32_t)h_digest[4 * k + 2] << 8 | h_digest[4 * k + 3];
    uint32_t bits = 4;
    while ((1ull << bits) < 2ull * n) ++bits;
    const uint32_t mask = (uint32_t)((1ull << bits) - 1);
    PEP_TRY(dev_reserve(ctx, W[0], (size_t)n * 4));
    PEP_TRY(dev_reserve(ctx, W[1], (size_t)n * 20));
    PEP_TRY(dev_reserve(ctx, W[2], ((size_t)n + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[3], ((size_t)n + 2) * 4));
    PEP_TRY(dev_reserve(ctx, W[4], ((size_t)mask + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[5], ((size_t)n + 1) * 4));
    PEP_HIP(ctx, hipMemcpyAsync(W[0].p, h_len, (size_t)n * 4, hipMemcpyHostToDevice, st));
    PEP_HIP(ctx, hipMemcpyAsync(W[1].p, words.data(), (size_t)n * 20, hipMemcpyHostToDevice, st));
    PEP_HIP(ctx, hipMemsetAsync(W[4].p, 0xFF, ((size_t)mask + 1) * 4, st));
    PEP_HIP(ctx, hipMemsetAsync(W[5].as<uint32_t>() + n, 0, 4, st));
    const unsigned g = (unsigned)ceil_div(n, 256);
    hipLaunchKernelGGL(k13_run_flags, dim3(g), dim3(256), 0, st, n, W[0].as<const uint32_t>(), W[2].as<uint32_t>());
    PEP_TRY(pep_scan_u32(ctx, W[2].as<const uint32_t>(), W[3].as<uint32_t>(), n, W[6]));
    uint32_t *fail = W[5].as<uint32_t>() + n;
    hipLaunchKernelGGL(k13_table<true>, dim3(g), dim3(256), 0, st, n, W[3].as<const uint32_t>(), W[2].as<const uint32_t>(), W[1].as<const uint32_t>(),
                       W[4].as<uint32_t>(), mask, (uint32_t *)nullptr, fail);
    hipLaunchKernelGGL(k13_table<false>, dim3(g), dim3(256), 0, st, n, W[3].as<const uint32_t>(), W[2].as<const uint32_t>(), W[1].as<const uint32_t>(),
                       W[4].as<uint32_t>(), mask, W[5].as<uint32_t>(), fail);
    PEP_HIP(ctx, hipGetLastError());
    uint32_t n_fail = 0;
    PEP_HIP(ctx, hipMemcpyAsync(h_rep, W[5].p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipMemcpyAsync(&n_fail, fail, 4, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipStreamSynchronize(st));
    if (n_fail) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_dedup: hash table probe failed");
    return PEP_OK;
}
