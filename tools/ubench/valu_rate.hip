// valu_rate.hip -- development microbenchmark (not part of the product): issue cost of the VALU
// instructions the FFT is made of, on gfx950, at 1/2/4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_rate tools/ubench/valu_rate.hip && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                                        \
    __global__ void name(float *out, int iters)                                                   \
    {                                                                                             \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = 1.0001f, b1 = 0.9999f;                                                         \
        for(int i = 0; i < iters; ++i) {                                                          \
            REP16(body)                                                                           \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
    }

// 4 independent instructions per body -> 64 per loop iteration
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_add, asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_log, asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_sqrt, asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_cnd, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");)

KERNEL(k_cnd_s, asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "s20", "s21");)
KERNEL(k_cnd_d, asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");)
KERNEL(k_max, asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_cmp, asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");)
KERNEL(k_cmpcnd, asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");)
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)

typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(name, body)                                                                       \
    __global__ void name(float *out, int iters)                                                   \
    {                                                                                             \
        f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;           \
        f2 b0 = {1.0001f, 0.9999f}, b1 = {0.5f, 0.25f};                                           \
        for(int i = 0; i < iters; ++i) {                                                          \
            REP16(body)                                                                           \
        }                                                                                         \
        f2 s = a0 + a1 + a2 + a3;                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                   \
    }
KERNEL2(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL2(k_pk_add, asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL2(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL2(k_pk_mov, asm volatile("v_pk_mov_b32 %0, %4, %4\n v_pk_mov_b32 %1, %4, %4\n v_pk_mov_b32 %2, %4, %4\n v_pk_mov_b32 %3, %4, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)

template<class K> void run(const char *name, K kernel, float *d_out, double ghz)
{
    const int iters = 2000; // 64 instr each
    for(int wps : {2, 4}) {
        const int block = 64 * 4 * wps; // waves per CU = 4 SIMDs * wps  (one block per CU)
        if(block > 1024) {
            // two blocks per CU
        }
        const int grid = 256 * (block > 1024 ? 2 : 1);
        const int b = block > 1024 ? 1024 : block;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(b), 0, 0, d_out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(b), 0, 0, d_out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64.0 * wps;
        const double cycles = ms * 1e-3 * ghz * 1e9;
        printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, wps, ms,
               cycles / instr_per_simd, ghz);
    }
}

int main()
{
    float *d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * 2 * 1024);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    printf("%s, %d CUs, clock %.2f GHz\n", p.gcnArchName, p.multiProcessorCount, ghz);
    run("v_fma", k_fma, d_out, ghz);
    run("v_add", k_add, d_out, ghz);
    run("v_mov", k_mov, d_out, ghz);
    run("v_cndmask", k_cnd, d_out, ghz);
    run("cnd_sgpr", k_cnd_s, d_out, ghz);
    run("cnd_dst", k_cnd_d, d_out, ghz);
    run("v_max", k_max, d_out, ghz);
    run("v_cmp", k_cmp, d_out, ghz);
    run("cmp+cnd", k_cmpcnd, d_out, ghz);
    run("v_and", k_and, d_out, ghz);
    run("v_mul", k_mul, d_out, ghz);
    run("v_pk_fma", k_pk_fma, d_out, ghz);
    run("v_pk_add", k_pk_add, d_out, ghz);
    run("v_pk_mul", k_pk_mul, d_out, ghz);
    run("v_pk_mov", k_pk_mov, d_out, ghz);
    run("v_log", k_log, d_out, ghz);
    run("v_sqrt", k_sqrt, d_out, ghz);
    return 0;
}
