// Scattered-load rate probe for gfx950 (MI355X): what does a CU sustain when every lane of every wavefront reads a different
// cache line of a table - the access pattern of the seed matcher (filter word, start[] pair, index entry per target position)?
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/gather_rate tools/micro/gather_rate.hip && tools/micro/gather_rate
//
// Varied: the load flavour (plain, non-temporal, sc1 = agent-scope relaxed atomic load, which bypasses the L1), the width per lane
// (4 / 8 / 16 bytes), the table size (2 MiB: one XCD's L2 holds it; 32 MiB: Infinity Cache; 512 MiB: HBM), the number of
// independent loads a lane has in flight (1 = a dependent chain, 8), and "pairs" (two lanes share a 128-byte line).
// Printed: requests per second (G/s), per cycle and CU at 2.4 GHz, and the line bytes that implies at 64 / 128 B per request.
// Grid: 256 CUs x 8 blocks x 256 threads (the matcher's residency), every thread issues ITER x UNROLL loads.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

enum { PLAIN = 0, NT = 1, SC1 = 2 };

template <typename T, int MODE>
__device__ __forceinline__ T load(const T *p)
{
    return *p;
}

template <>
__device__ __forceinline__ uint64_t load<uint64_t, NT>(const uint64_t *p) { return __builtin_nontemporal_load(p); }
template <>
__device__ __forceinline__ uint32_t load<uint32_t, SC1>(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <>
__device__ __forceinline__ uint64_t load<uint64_t, SC1>(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <>
__device__ __forceinline__ uint4 load<uint4, NT>(const uint4 *p)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <>
__device__ __forceinline__ uint4 load<uint4, SC1>(const uint4 *p)
{
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint64_t v) { return (uint32_t)v ^ (uint32_t)(v >> 32); }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// UNROLL independent loads per trip; DEP: the next trip's addresses depend on this trip's data (a look-up chain)
template <typename T, int MODE, int UNROLL, bool DEP>
__global__ __launch_bounds__(256) void gather(const T *__restrict__ table, uint32_t mask, int iters, int stride_elems, uint32_t *__restrict__ out)
{
    uint32_t seed = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        T v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            seed = mix(seed + 0x9e3779b9u * (u + 1));
            v[u] = load<T, MODE>(table + (size_t)(seed & mask) * stride_elems);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= fold(v[u]);
        if (DEP) seed ^= acc & 1u;                  // (the table holds even words: the address stream stays the same, the dependence is real)
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <typename T, int MODE, int UNROLL, bool DEP>
double run(const void *table, size_t bytes, int line_stride_bytes, int iters, uint32_t *out)
{
    const size_t slots = bytes / line_stride_bytes;          // one addressable element per `line_stride_bytes`
    uint32_t mask = 1;
    while ((size_t)mask * 2 <= slots) mask *= 2;
    mask -= 1;
    const int stride = line_stride_bytes / (int)sizeof(T);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<T, MODE, UNROLL, DEP>), dim3(2048), dim3(256), 0, 0, (const T *)table, mask, 4, stride, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<T, MODE, UNROLL, DEP>), dim3(2048), dim3(256), 0, 0, (const T *)table, mask, iters, stride, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double reqs = 2048.0 * 256 * iters * UNROLL;
    return reqs / (ms * 1e-3);
}

int main()
{
    const size_t sizes[3] = {2u << 20, 32u << 20, 512u << 20};
    void *table;
    uint32_t *out;
    CHECK(hipMalloc(&table, sizes[2]));
    CHECK(hipMemset(table, 0, sizes[2]));
    CHECK(hipMalloc(&out, 64));
    printf("# scattered loads, 2048 blocks x 256 threads; G requests/s | requests per cycle and CU (2.4 GHz, 256 CUs)\n");
    printf("%-10s %-6s %-5s %-9s %10s %10s %10s\n", "flavour", "bytes", "dep", "stride", "2MiB", "32MiB", "512MiB");
#define ROW(T, MODE, NAME, UNROLL, DEP, STRIDE)                                                                        \
    do {                                                                                                               \
        printf("%-10s %-6d %-5s %-9d", NAME, (int)sizeof(T), DEP ? "chain" : (UNROLL == 8 ? "8" : "2"), STRIDE);       \
        for (int s = 0; s < 3; ++s) {                                                                                  \
            const double r = run<T, MODE, UNROLL, DEP>(table, sizes[s], STRIDE, DEP ? 200 : 50, out);                  \
            printf(" %6.1f|%4.2f", r / 1e9, r / 2.4e9 / 256);                                                          \
        }                                                                                                              \
        printf("\n");                                                                                                  \
    } while (0)
    ROW(uint32_t, PLAIN, "plain", 8, false, 128);
    ROW(uint64_t, PLAIN, "plain", 8, false, 128);
    ROW(uint4, PLAIN, "plain", 8, false, 128);
    ROW(uint64_t, NT, "nt", 8, false, 128);
    ROW(uint4, NT, "nt", 8, false, 128);
    ROW(uint64_t, SC1, "sc1", 8, false, 128);
    ROW(uint4, SC1, "sc1", 8, false, 128);
    ROW(uint64_t, PLAIN, "plain", 1, true, 128);
    ROW(uint64_t, PLAIN, "plain", 2, false, 128);
    ROW(uint64_t, NT, "nt", 1, true, 128);
    // any 8-byte word (16 per 128-byte line, so some lanes of a wavefront share lines now and then) and any 64-byte half line
    ROW(uint64_t, PLAIN, "plain", 8, false, 8);
    ROW(uint64_t, PLAIN, "plain", 8, false, 64);
    ROW(uint64_t, NT, "nt", 8, false, 64);
    return 0;
}
