// what a CU of this box holds of a kernel shaped like the seed matcher: 256 threads and a static LDS buffer of 16.7 / 8.5 / 4.4 KB per block
// (hipOccupancyMaxActiveBlocksPerMultiprocessor), and the device properties that bound it.  hipcc --offload-arch=gfx950 -O2 -o occupancy occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORDS> __global__ __launch_bounds__(256) void shaped(unsigned long long *out)
{
    __shared__ unsigned long long buf[WORDS];
    buf[threadIdx.x % WORDS] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = buf[(blockIdx.x * 7) % WORDS];
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, LDS per block %zu, LDS per CU %zu, registers per block %d, max threads per CU %d, warp %d, clock %d kHz, memory clock %d kHz\n", p.gcnArchName, p.multiProcessorCount,
           p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.maxThreadsPerMultiProcessor, p.warpSize, p.clockRate, p.memoryClockRate);
    int a = 0, b = 0, c = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, shaped<2048 + 40>, 256, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, shaped<1024 + 40>, 256, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, shaped<512 + 40>, 256, 0);
    printf("blocks of 256 threads per CU by LDS per block: 16.7 KB -> %d, 8.5 KB -> %d, 4.4 KB -> %d\n", a, b, c);
    return 0;
}
