// LSD radix sort of u64 keys (8-bit digits, stable), used for the candidate list (K4) and k-mer tables.
// Per pass: per-tile digit histogram -> exclusive scan of the [digit][tile] table -> stable scatter.
#include "common.h"

namespace {

constexpr int ST = 256;                 // threads per block
constexpr int SI = 8;                   // keys per thread
constexpr int STILE = ST * SI;

__global__ __launch_bounds__(ST) void sort_hist(const uint64_t *__restrict__ keys, uint32_t *__restrict__ hist, uint64_t n, int shift, uint32_t nb)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * STILE;
#pragma unroll
    for (int r = 0; r < SI; ++r) {
        uint64_t i = base + (uint64_t)r * ST + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(ST) void sort_scatter(const uint64_t *__restrict__ keys, uint64_t *__restrict__ out, const uint32_t *__restrict__ hist_scan,
                                                   uint64_t n, int shift, uint32_t nb)
{
    __shared__ uint32_t wave_cnt[ST / 64][256];
    __shared__ uint32_t digit_base[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    digit_base[threadIdx.x] = hist_scan[(uint64_t)threadIdx.x * nb + blockIdx.x];
    const uint64_t base = (uint64_t)blockIdx.x * STILE;
    for (int r = 0; r < SI; ++r) {
#pragma unroll
        for (int w = 0; w < ST / 64; ++w) wave_cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const uint64_t i = base + (uint64_t)r * ST + threadIdx.x;
        const bool valid = i < n;
        const uint64_t key = valid ? keys[i] : 0;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
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
        uint32_t add = 0;
#pragma unroll
        for (int w = 0; w < ST / 64; ++w) add += wave_cnt[w][threadIdx.x];
        digit_base[threadIdx.x] += add;
        __syncthreads();
    }
}

}  // namespace

// sorts d_keys ascending on their low `bits` bits; d_tmp has room for n keys; the result is in d_keys
int pep_sort_u64(pep_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint64_t n, int bits, DevBuf &hist)
{
    if (n < 2) return PEP_OK;
    if (n >= (1ull << 32)) return pep_fail(ctx, PEP_ERR_LIMIT, "pep_sort_u64: more than 2^32 keys");
    const uint32_t nb = (uint32_t)ceil_div(n, STILE);
    const uint64_t hn = 256ull * nb;
    // hist buffer: [hn + 1] counts, then scan scratch
    PEP_TRY(dev_reserve(ctx, hist, (hn + 2) * sizeof(uint32_t)));
    DevBuf &scan_tmp = ctx->ws[7];
    int passes = (bits + 7) / 8;
    if (passes & 1) ++passes;               // even number of passes so the result lands in d_keys
    uint64_t *src = d_keys, *dst = d_tmp;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * 8;
        if (shift >= bits || shift >= 64) {  // padding pass beyond the key width (keeps the pass count even): plain copy
            PEP_HIP(ctx, hipMemcpyAsync(dst, src, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            hipLaunchKernelGGL(sort_hist, dim3(nb), dim3(ST), 0, ctx->stream, (const uint64_t *)src, hist.as<uint32_t>(), n, shift, nb);
            PEP_TRY(pep_scan_u32(ctx, hist.as<uint32_t>(), hist.as<uint32_t>(), hn, scan_tmp));
            hipLaunchKernelGGL(sort_scatter, dim3(nb), dim3(ST), 0, ctx->stream, (const uint64_t *)src, dst, (const uint32_t *)hist.as<uint32_t>(), n, shift, nb);
        }
        uint64_t *t = src; src = dst; dst = t;
    }
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}
