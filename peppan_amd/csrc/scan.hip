// Exclusive prefix sums over device arrays (u32 / u64), n+1 outputs (out[n] = total).
// u32: ONE launch - every tile publishes its sum, looks back over the tiles before it (a wavefront inspects 64 predecessors at a
// time) and publishes its inclusive prefix ("decoupled look-back").  Reads the input once and writes it once; the searches run a
// dozen short scans per pass, so the two launches saved per scan matter more than the bytes.
// u64: the same with 48-bit sums in the status word.  (n = 0 or n >= 2^40: per-tile reduce -> scan of the tile sums -> per-tile
// scan + offset.)
#include "common.h"
#include "lookback.h"

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

// ---- single-launch u32 scan.  state[0] = ticket counter (tiles take their index from it, so a tile only ever waits for tiles that
// are already running); state[1 + tile] = (epoch << 34) | (flag << 32) | value with flag 1 = the tile's own sum, 2 = sum of everything
// up to and including the tile.  The epoch is a per-call number: words left by earlier scans never match, so nothing is cleared.
constexpr uint64_t FLAG_SUM = 1, FLAG_PREFIX = 2;

// status word layout: value in the low VB bits, flag in the next two, epoch above.  u32 sums: 32 value bits, 30 epoch bits.
// u64 sums (block / run / CIGAR totals, all far below 2^48): 48 value bits, 14 epoch bits - the state is cleared every 2^14 - 1 scans.
template <class T> struct ScanWord;
template <> struct ScanWord<uint32_t> { static constexpr int VB = 32; static constexpr uint32_t EPOCH_MASK = (1u << 30) - 1; };
template <> struct ScanWord<uint64_t> { static constexpr int VB = 48; static constexpr uint32_t EPOCH_MASK = (1u << 14) - 1; };

template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_lookback(const T *in, T *out, uint64_t n, uint32_t nb, uint64_t *state, uint32_t ticket_base, uint64_t epoch,
                                                              T *total_out)
{
    constexpr int VB = ScanWord<T>::VB;
    constexpr uint64_t VMASK = (1ull << VB) - 1;
    __shared__ T lds[4];
    __shared__ uint32_t s_tile;
    __shared__ T s_prefix;
    if (threadIdx.x == 0) s_tile = atomicAdd(reinterpret_cast<uint32_t *>(state), 1u) - ticket_base;
    __syncthreads();
    const uint32_t tile = s_tile;
    uint64_t *status = state + 1;
    const uint64_t base = (uint64_t)tile * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    T tot;
    T ex = block_excl_scan(s, &tot, lds);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        T prefix = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&status[tile], (epoch << (VB + 2)) | (FLAG_SUM << VB) | ((uint64_t)tot & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // window of 64 predecessors, lane 0 = the nearest; stop at the nearest tile that already knows its inclusive prefix
            for (int64_t hi = (int64_t)tile - 1; hi >= 0; hi -= 64) {
                const int64_t idx = hi - lane;
                uint64_t w = (epoch << (VB + 2)) | (FLAG_PREFIX << VB);           // before tile 0: prefix 0
                if (idx >= 0)
                    do { w = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> (VB + 2)) != epoch);
                const uint64_t is_prefix = __ballot(((w >> VB) & 3u) == FLAG_PREFIX);
                const int stop = is_prefix ? __ffsll((unsigned long long)is_prefix) - 1 : 63;      // lanes 0..stop contribute
                T part = lane <= stop ? (T)(w & VMASK) : (T)0;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
                prefix += part;
                if (is_prefix) break;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&status[tile], (epoch << (VB + 2)) | (FLAG_PREFIX << VB) | ((uint64_t)(prefix + tot) & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = prefix;
        }
    }
    __syncthreads();
    ex += s_prefix;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (tile == nb - 1 && threadIdx.x == SCAN_THREADS - 1) { out[n] = ex; if (total_out) *total_out = ex; }
}

template <class T>
int scan_onepass(pep_ctx *ctx, const T *d_in, T *d_out, uint64_t n, T *d_total)
{
    const uint64_t nb = ceil_div(n, SCAN_TILE);
    pep_ctx::ScanState &S = ctx->scan_state[sizeof(T) == 8];       // one state area per width: the two word layouts must never meet
    bool clear = false;
    if ((nb + 1) * sizeof(uint64_t) > S.buf.cap) {
        // a new (or larger) state area starts from zeros: epoch 0 is never used, so no word of it can pass for a published one
        PEP_TRY(dev_reserve(ctx, S.buf, (nb + 1) * sizeof(uint64_t) * 2));
        clear = true;
    }
    if (S.dirty) { clear = true; S.dirty = false; }
    S.epoch = (S.epoch + 1) & ScanWord<T>::EPOCH_MASK;
    if (S.epoch == 0) { clear = true; S.epoch = 1; }   // wrapped: forget every old word
    if (clear) {
        PEP_HIP(ctx, hipMemsetAsync(S.buf.p, 0, S.buf.cap, ctx->stream));
        S.ticket_base = 0;
    }
    hipLaunchKernelGGL(scan_lookback<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, d_in, d_out, n, (uint32_t)nb, S.buf.as<uint64_t>(), S.ticket_base,
                       (uint64_t)S.epoch, d_total);
    S.ticket_base += (uint32_t)nb;
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

template <class T>
int scan_impl(pep_ctx *ctx, const T *d_in, T *d_out, uint64_t n, DevBuf &tmp, T *d_total)
{
    if (n == 0) {
        PEP_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(T), ctx->stream));
        if (d_total) PEP_HIP(ctx, hipMemsetAsync(d_total, 0, sizeof(T), ctx->stream));
        return PEP_OK;
    }
    const uint64_t nb = ceil_div(n, SCAN_TILE);
    PEP_TRY(dev_reserve(ctx, tmp, (nb + 1) * sizeof(T)));
    T *partial = tmp.as<T>();
    hipLaunchKernelGGL(scan_reduce<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, d_in, partial, n);
    hipLaunchKernelGGL(scan_partials<T>, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, partial, nb);
    hipLaunchKernelGGL(scan_apply<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, d_in, d_out, (const T *)partial, n, nb);
    if (d_total) PEP_HIP(ctx, hipMemcpyAsync(d_total, d_out + n, sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

}  // namespace

int pep_scan_u32(pep_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n, DevBuf &tmp, uint32_t *d_total)
{
    if (n == 0 || n >= (1ull << 40)) return scan_impl<uint32_t>(ctx, d_in, d_out, n, tmp, d_total);
    return scan_onepass<uint32_t>(ctx, d_in, d_out, n, d_total);
}

// the single launch carries 48-bit sums; callers whose totals could pass 2^48 say so (none does: the u64 scans add up block, run and
// CIGAR counts of one search)
int pep_scan_u64(pep_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t n, DevBuf &tmp, uint64_t *d_total)
{
    if (n == 0 || n >= (1ull << 40)) return scan_impl<uint64_t>(ctx, d_in, d_out, n, tmp, d_total);
    return scan_onepass<uint64_t>(ctx, d_in, d_out, n, d_total);
}

// n 32-bit words from pinned host memory into device memory BY A KERNEL (the device reads the host buffer over the bus): between two
// kernels a copy command costs the GPU about 10 us of idle time in front of it, a kernel none
namespace {
__global__ __launch_bounds__(256) void copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace

// both directions in one launch: n_up words from pinned host memory into device memory, n_down words from device memory into pinned host memory
namespace {
__global__ __launch_bounds__(256) void exchange_words(uint32_t *__restrict__ up_dst, const uint32_t *__restrict__ up_src, uint64_t n_up,
                                                      uint32_t *__restrict__ down_dst, const uint32_t *__restrict__ down_src, uint64_t n_down)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = t0; i < n_up; i += stride) up_dst[i] = up_src[i];
    for (uint64_t i = t0; i < n_down; i += stride) down_dst[i] = down_src[i];
}
}  // namespace

int pep_exchange_pinned(pep_ctx *ctx, void *d_up_dst, const void *pinned_up_src, uint64_t n_up_words, void *pinned_down_dst, const void *d_down_src, uint64_t n_down_words)
{
    if (n_up_words + n_down_words == 0) return PEP_OK;
    hipLaunchKernelGGL(exchange_words, dim3((unsigned)std::min<uint64_t>(ceil_div(std::max(n_up_words, n_down_words), 256), 1024)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<uint32_t *>(d_up_dst), reinterpret_cast<const uint32_t *>(pinned_up_src), n_up_words,
                       reinterpret_cast<uint32_t *>(pinned_down_dst), reinterpret_cast<const uint32_t *>(d_down_src), n_down_words);
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

int pep_copy_from_pinned(pep_ctx *ctx, void *d_dst, const void *pinned_src, uint64_t n_words)
{
    if (n_words == 0) return PEP_OK;
    hipLaunchKernelGGL(copy_words, dim3((unsigned)std::min<uint64_t>(ceil_div(n_words, 256), 1024)), dim3(256), 0, ctx->stream, reinterpret_cast<uint32_t *>(d_dst),
                       reinterpret_cast<const uint32_t *>(pinned_src), n_words);
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

// (lookback.h) the state area, ticket base and epoch of one launch of a kernel that uses the look-back device functions
int pep_lookback_begin(pep_ctx *ctx, pep_ctx::ScanState &S, uint64_t n_tiles, uint32_t epoch_mask, uint64_t **state, uint32_t *ticket_base, uint64_t *epoch)
{
    bool clear = false;
    if ((n_tiles + 1) * sizeof(uint64_t) > S.buf.cap) {
        // a new (or larger) state area starts from zeros: epoch 0 is never used, so no word of it can pass for a published one
        PEP_TRY(dev_reserve(ctx, S.buf, (n_tiles + 1) * sizeof(uint64_t) * 2));
        clear = true;
    }
    if (S.dirty) { clear = true; S.dirty = false; }
    S.epoch = (S.epoch + 1) & epoch_mask;
    if (S.epoch == 0) { clear = true; S.epoch = 1; }   // wrapped: forget every old word
    if (clear) {
        PEP_HIP(ctx, hipMemsetAsync(S.buf.p, 0, S.buf.cap, ctx->stream));
        S.ticket_base = 0;
    }
    *state = S.buf.as<uint64_t>();
    *ticket_base = S.ticket_base;
    *epoch = S.epoch;
    S.ticket_base += (uint32_t)n_tiles;
    return PEP_OK;
}
