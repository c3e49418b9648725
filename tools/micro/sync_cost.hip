// cost of one "launch tiny kernel -> read 8 bytes back -> continue" round trip: pageable copy vs pinned copy vs kernel writing host-mapped memory
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void bump(unsigned long long *p) { *p += 1; }
__global__ void bump2(unsigned long long *p, unsigned long long *host) { *p += 1; *host = *p; __threadfence_system(); }
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    unsigned long long *d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
    unsigned long long pageable = 0, *pinned, *mapped, *mapped_dev;
    hipHostMalloc(&pinned, 64, hipHostMallocDefault);
    hipHostMalloc(&mapped, 64, hipHostMallocMapped); hipHostGetDevicePointer((void **)&mapped_dev, mapped, 0);
    const int N = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 0) { hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, st, d); hipMemcpyAsync(&pageable, d, 8, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
                if (mode == 1) { hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, st, d); hipMemcpyAsync(pinned, d, 8, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
                if (mode == 2) { hipLaunchKernelGGL(bump2, dim3(1), dim3(1), 0, st, d, mapped_dev); hipStreamSynchronize(st); }
                if (mode == 3) { hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, st, d); hipStreamSynchronize(st); }
            }
            auto t1 = std::chrono::steady_clock::now();
            if (rep) printf("mode %d (%s): %.1f us per round trip (value %llu)\n", mode, mode == 0 ? "pageable copy" : mode == 1 ? "pinned copy" : mode == 2 ? "kernel writes mapped host memory" : "launch + sync only",
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / N, mode == 0 ? pageable : mode == 1 ? *pinned : *mapped);
        }
    }
    return 0;
}
