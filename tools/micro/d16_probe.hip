// Does ds_read_i8_d16 / _d16_hi keep the other half of the destination register on gfx950 (SRAM-ECC on)?
// hipcc --offload-arch=gfx950 -O2 -o d16_probe d16_probe.hip && ./d16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out)
{
    __shared__ signed char tab[256];
    tab[threadIdx.x] = (signed char)(threadIdx.x - 100);
    tab[threadIdx.x + 64] = (signed char)(threadIdx.x + 1);
    __syncthreads();
    unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) signed char *)tab + threadIdx.x, a1 = a0 + 64;
    unsigned r = 0x12345678u, r2 = 0x12345678u, r3;
    asm volatile("ds_read_i8_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(a0));
    asm volatile("ds_read_i8_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(r2) : "v"(a0));
    asm volatile("ds_read_i8_d16 %0, %1\n\tds_read_i8_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r3) : "v"(a0), "v"(a1));
    out[threadIdx.x * 3] = r; out[threadIdx.x * 3 + 1] = r2; out[threadIdx.x * 3 + 2] = r3;
}
int main()
{
    unsigned *d, h[192];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 63}) printf("lane %d: d16_hi over 0x12345678 -> %08x   d16 -> %08x   pair -> %08x (expect lo=%04x hi=%04x)\n", l, h[l * 3], h[l * 3 + 1], h[l * 3 + 2],
                                    (unsigned)(unsigned short)(short)(l - 100), (unsigned)(unsigned short)(short)(l + 1));
    return 0;
}
