// Exclusive prefix sums over device arrays (u32 / u64), n+1 outputs (out[n] = total).
// Three passes: per-tile reduce -> scan of the tile sums (one block) -> per-tile scan + offset.
// HBM-bound: reads the input twice and writes it once.
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <class T>
__device__ __forceinline__ T wave_incl_scan(T v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns block total in *total
template <class T>
__device__ __forceinline__ T block_excl_scan(T v, T *total, T *lds /* [4] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = wave_incl_scan(v, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        T s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce(const T *__restrict__ in, T *__restrict__ partial, uint64_t n)
{
    __shared__ T lds[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    T s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + (uint64_t)k * SCAN_THREADS + threadIdx.x;
        if (i < n) s += in[i];
    }
    T tot;
    block_excl_scan(s, &tot, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_partials(T *partial, uint64_t nb)
{
    __shared__ T lds[4];
    T carry = 0;
    for (uint64_t b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
        uint64_t i = b0 + threadIdx.x;
        T v = i < nb ? partial[i] : 0, tot;
        T ex = block_excl_scan(v, &tot, lds);
        if (i < nb) partial[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) partial[nb] = carry;
}

template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const T *in, T *out, const T *__restrict__ partial,
                                                           uint64_t n, uint64_t nb)
{
    __shared__ T lds[4];
    // thread t owns items [t*ITEMS, t*ITEMS+ITEMS) of the tile so that the in-thread order is the array order
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    T tot;
    T ex = block_excl_scan(s, &tot, lds) + partial[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = partial[nb];
}

template <class T>
int scan_impl(pep_ctx *ctx, const T *d_in, T *d_out, uint64_t n, DevBuf &tmp)
{
    if (n == 0) {
        PEP_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(T), ctx->stream));
        return PEP_OK;
    }
    const uint64_t nb = ceil_div(n, SCAN_TILE);
    PEP_TRY(dev_reserve(ctx, tmp, (nb + 1) * sizeof(T)));
    T *partial = tmp.as<T>();
    hipLaunchKernelGGL(scan_reduce<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, d_in, partial, n);
    hipLaunchKernelGGL(scan_partials<T>, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, partial, nb);
    hipLaunchKernelGGL(scan_apply<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, d_in, d_out, (const T *)partial, n, nb);
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

}  // namespace

int pep_scan_u32(pep_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n, DevBuf &tmp) { return scan_impl<uint32_t>(ctx, d_in, d_out, n, tmp); }
int pep_scan_u64(pep_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t n, DevBuf &tmp) { return scan_impl<uint64_t>(ctx, d_in, d_out, n, tmp); }
