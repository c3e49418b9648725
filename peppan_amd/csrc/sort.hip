// LSD radix sort of u64 keys (stable), used for the candidate list (K4) and k-mer tables.
// Per pass: per-tile digit histogram -> exclusive scan of the [digit][tile] table -> stable scatter.
// The digit width is chosen per call: the fewest passes (an even number, so that the result lands in the input buffer without a
// copy) with digits of at most 11 bits - the candidate keys of a search are 35-40 bits wide: four passes of 9-10 bits instead of
// five of 8 plus a copy.  Every pass is three launches, and the sort of ~10^5 keys is bound by launches, not by bytes.
#include "common.h"

namespace {

constexpr int ST = 256;                 // threads per block
constexpr int SI = 8;                   // keys per thread
constexpr int STILE = ST * SI;
constexpr int MAX_DIGIT_BITS = 11;

template <int DB>
__global__ __launch_bounds__(ST) void sort_hist(const uint64_t *__restrict__ keys, uint32_t *__restrict__ hist, uint64_t n, int shift, uint32_t nb)
{
    constexpr uint32_t R = 1u << DB;
    __shared__ uint32_t h[R];
    for (uint32_t d = threadIdx.x; d < R; d += ST) h[d] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * STILE;
#pragma unroll
    for (int r = 0; r < SI; ++r) {
        uint64_t i = base + (uint64_t)r * ST + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (R - 1)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < R; d += ST) hist[(uint64_t)d * nb + blockIdx.x] = h[d];
}

template <int DB>
__global__ __launch_bounds__(ST) void sort_scatter(const uint64_t *__restrict__ keys, uint64_t *__restrict__ out, const uint32_t *__restrict__ hist_scan,
                                                   uint64_t n, int shift, uint32_t nb)
{
    constexpr uint32_t R = 1u << DB;
    __shared__ uint32_t wave_cnt[ST / 64][R];
    __shared__ uint32_t digit_base[R];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t d = threadIdx.x; d < R; d += ST) digit_base[d] = hist_scan[(uint64_t)d * nb + blockIdx.x];
    const uint64_t base = (uint64_t)blockIdx.x * STILE;
    for (int r = 0; r < SI; ++r) {
        for (uint32_t d = threadIdx.x; d < R; d += ST) {
#pragma unroll
            for (int w = 0; w < ST / 64; ++w) wave_cnt[w][d] = 0;
        }
        __syncthreads();
        const uint64_t i = base + (uint64_t)r * ST + threadIdx.x;
        const bool valid = i < n;
        const uint64_t key = valid ? keys[i] : 0;
        const uint32_t d = (uint32_t)(key >> shift) & (R - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        if (valid && rank == 0) wave_cnt[wave][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t o = digit_base[d] + rank;
            for (int w = 0; w < wave; ++w) o += wave_cnt[w][d];
            out[o] = key;
        }
        __syncthreads();
        for (uint32_t dd = threadIdx.x; dd < R; dd += ST) {
            uint32_t add = 0;
#pragma unroll
            for (int w = 0; w < ST / 64; ++w) add += wave_cnt[w][dd];
            digit_base[dd] += add;
        }
        __syncthreads();
    }
}

template <int DB>
int sort_passes(pep_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint64_t n, int bits, int passes, DevBuf &hist, uint32_t nb)
{
    const uint64_t hn = (uint64_t)(1u << DB) * nb;
    PEP_TRY(dev_reserve(ctx, hist, (hn + 2) * sizeof(uint32_t)));
    uint64_t *src = d_keys, *dst = d_tmp;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * DB;
        if (shift >= bits || shift >= 64) {  // padding pass beyond the key width (keeps the pass count even): plain copy
            PEP_HIP(ctx, hipMemcpyAsync(dst, src, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            hipLaunchKernelGGL(sort_hist<DB>, dim3(nb), dim3(ST), 0, ctx->stream, (const uint64_t *)src, hist.as<uint32_t>(), n, shift, nb);
            PEP_TRY(pep_scan_u32(ctx, hist.as<uint32_t>(), hist.as<uint32_t>(), hn, ctx->ws[7]));
            hipLaunchKernelGGL(sort_scatter<DB>, dim3(nb), dim3(ST), 0, ctx->stream, (const uint64_t *)src, dst, (const uint32_t *)hist.as<uint32_t>(), n, shift, nb);
        }
        uint64_t *t = src; src = dst; dst = t;
    }
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

}  // namespace

// sorts d_keys ascending on their low `bits` bits; d_tmp has room for n keys; the result is in d_keys
int pep_sort_u64(pep_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint64_t n, int bits, DevBuf &hist)
{
    if (n < 2) return PEP_OK;
    if (n >= (1ull << 32)) return pep_fail(ctx, PEP_ERR_LIMIT, "pep_sort_u64: more than 2^32 keys");
    bits = std::max(1, std::min(64, bits));
    const uint32_t nb = (uint32_t)ceil_div(n, STILE);
    int passes = 2;
    while ((bits + passes - 1) / passes > MAX_DIGIT_BITS) passes += 2;
    const int db = std::max(8, (bits + passes - 1) / passes);
    switch (db) {
        case 8: return sort_passes<8>(ctx, d_keys, d_tmp, n, bits, passes, hist, nb);
        case 9: return sort_passes<9>(ctx, d_keys, d_tmp, n, bits, passes, hist, nb);
        case 10: return sort_passes<10>(ctx, d_keys, d_tmp, n, bits, passes, hist, nb);
        default: return sort_passes<11>(ctx, d_keys, d_tmp, n, bits, passes, hist, nb);
    }
}
