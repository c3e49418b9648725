// VALU issue-rate probe for gfx950 (MI355X): how many shader cycles does one SIMD need per wave64 instruction, for the
// instructions the Smith-Waterman passes are made of?  Settles "2 cycles (SIMD-32) or 4 cycles per wave64 VALU op" for
// v_pk_max_i16 / v_pk_add_i16 / v_pk_sub_i16 / v_add3_u32 / v_mov_b32_dpp (wave_shr:1) / v_max_i32 / v_add_u32 / v_fma_f32
// at 1, 2, 4, 8 waves per SIMD.  bench.py derives roofline.valu_issue_frac from the numbers this prints
// (profiles/r02_valu_rate.txt), not from an assumed constant.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/valu_rate tools/micro/valu_rate.hip && tools/micro/valu_rate
//
// Method: a block = 256 threads = one wave on each SIMD of a CU; dynamic LDS sized so that exactly W blocks fit a CU;
// grid = 256 CUs x W blocks.  Every wave runs ITER x 32 instructions over 8 independent accumulators (no dependent
// issue closer than 8 instructions) between two s_memtime reads.  With W waves resident per SIMD and all of them in the
// loop, the SIMD issues W x ITER x 32 instructions in the slowest wave's elapsed cycles: cycles per instruction per SIMD
// = elapsed / (W x ITER x 32).  s_memrealtime (100 MHz) beside s_memtime gives the shader clock during the run.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define ITER 2000

#define BODY8(OP)                                                                                                     \
    OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define BODY32(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP)

#define KERNEL(NAME, ASM)                                                                                             \
    __global__ void __launch_bounds__(256) NAME(unsigned long long *out, int x, int y)                                \
    {                                                                                                                 \
        extern __shared__ int lds[];                                                                                  \
        int a0 = x + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        int b = y, c = y + 3;                                                                                         \
        if (x == 12345) lds[threadIdx.x] = x;                                                                         \
        __syncthreads();                                                                                              \
        unsigned long long t0 = __builtin_readcyclecounter();                                                         \
        unsigned long long r0 = wall_clock64();                                                                       \
        for (int it = 0; it < ITER; ++it) {                                                                           \
            BODY32(ASM)                                                                                               \
        }                                                                                                             \
        unsigned long long t1 = __builtin_readcyclecounter();                                                         \
        unsigned long long r1 = wall_clock64();                                                                       \
        int s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                                \
        if (s == 0x7fffffff) lds[0] = s;                                                                              \
        if ((threadIdx.x & 63) == 0) {                                                                                \
            size_t w = (size_t)blockIdx.x * 4 + threadIdx.x / 64;                                                     \
            out[2 * w] = t1 - t0;                                                                                     \
            out[2 * w + 1] = r1 - r0;                                                                                 \
        }                                                                                                             \
    }

#define OP_PKMAX(r) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(b));
#define OP_PKADD(r) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r) : "v"(b));
#define OP_PKSUB(r) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(r) : "v"(b));
#define OP_ADD3(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define OP_DPP(r) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(b));
#define OP_MAX32(r) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(b));
#define OP_ADD32(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));
#define OP_FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define OP_PKMAD(r) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define OP_PKLSHR(r) asm volatile("v_pk_lshrrev_b16 %0, 15, %0 op_sel_hi:[0,1]" : "+v"(r));
#define OP_MOV(r) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(b));
#define OP_AND(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(b));
#define OP_LSHLOR(r) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(r) : "v"(b));
#define OP_MULLO(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(b));

KERNEL(k_pk_max, OP_PKMAX)
KERNEL(k_pk_add, OP_PKADD)
KERNEL(k_pk_sub, OP_PKSUB)
KERNEL(k_add3, OP_ADD3)
KERNEL(k_dpp, OP_DPP)
KERNEL(k_max32, OP_MAX32)
KERNEL(k_add32, OP_ADD32)
KERNEL(k_fma, OP_FMA)
KERNEL(k_pk_mad, OP_PKMAD)
KERNEL(k_pk_lshr, OP_PKLSHR)
KERNEL(k_mov, OP_MOV)
KERNEL(k_and, OP_AND)
KERNEL(k_lshl_or, OP_LSHLOR)
KERNEL(k_mul_lo, OP_MULLO)

typedef void (*kern_t)(unsigned long long *, int, int);

int main()
{
    struct { const char *name; kern_t k; } tab[] = {
        {"v_pk_max_i16", k_pk_max}, {"v_pk_add_i16", k_pk_add}, {"v_pk_sub_i16", k_pk_sub}, {"v_pk_mad_u16", k_pk_mad},
        {"v_pk_lshrrev_b16", k_pk_lshr}, {"v_add3_u32", k_add3}, {"v_mov_b32_dpp wave_shr:1", k_dpp}, {"v_mov_b32", k_mov},
        {"v_max_i32", k_max32}, {"v_add_u32", k_add32}, {"v_and_b32", k_and}, {"v_lshl_or_b32", k_lshl_or}, {"v_mul_lo_u32", k_mul_lo}, {"v_fma_f32", k_fma},
    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, %d KiB LDS per CU assumed 160; %d x 32 instructions per wave between two s_memtime reads\n", prop.gcnArchName, cus, 160, ITER);
    printf("# cycles/instr/SIMD = slowest wave's elapsed shader cycles / (waves per SIMD x instructions per wave); clock = s_memtime / s_memrealtime x 100 MHz\n");
    printf("%-26s %6s %12s %12s %10s\n", "instruction", "w/SIMD", "cyc/instr", "(median wave)", "clock GHz");
    unsigned long long *d;
    const size_t max_waves = (size_t)cus * 8 * 4;
    hipMalloc(&d, max_waves * 16);
    std::vector<unsigned long long> h(max_waves * 2);
    for (auto &e : tab) {
        for (int W : {1, 2, 4, 8}) {
            const size_t lds = (size_t)(160 * 1024 / W) - 512;           // exactly W blocks fit the CU's LDS
            hipFuncSetAttribute((const void *)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int blocks = cus * W;
            for (int rep = 0; rep < 2; ++rep) {                          // first run warms the code up
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, d, 1, 2);
                hipDeviceSynchronize();
            }
            if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", e.name); continue; }
            hipMemcpy(h.data(), d, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost);
            std::vector<double> cyc, clk;
            for (size_t w = 0; w < (size_t)blocks * 4; ++w) {
                cyc.push_back((double)h[2 * w]);
                clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
            }
            std::sort(cyc.begin(), cyc.end());
            std::sort(clk.begin(), clk.end());
            const double n = (double)W * ITER * 32;
            printf("%-26s %6d %12.3f %12.3f %10.3f\n", e.name, W, cyc.back() / n, cyc[cyc.size() / 2] / n, clk[clk.size() / 2]);
        }
    }
    hipFree(d);
    return 0;
}
