// tools/micro/pk_rate.hip -- development aid: issue rate of packed f32 VALU instructions on gfx950 next to scalar ones.
// Does v_pk_fma_f32 / v_pk_add_f32 retire two f32 operations per lane in the time v_fma_f32 retires one?  (The FFT butterflies are
// complex additions: one v_pk_add_f32 each if it does.)   hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template<int MODE> __global__ void k(float *out, int iters, float s)
{
    f2 a[8];
    for(int i = 0; i < 8; ++i)
        a[i] = f2{(float)threadIdx.x + i, (float)i * 0.5f};
    const f2 m = f2{s, 1.0f - s}, c = f2{0.25f, 0.125f};
    for(int it = 0; it < iters; ++it) {
#pragma unroll
        for(int r = 0; r < REP; ++r) {
#pragma unroll
            for(int i = 0; i < 8; ++i) {
                if(MODE == 0) { // scalar fma: two instructions per pair
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
                } else if(MODE == 1) { // packed fma: one instruction per pair
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(c));
                } else if(MODE == 2) { // scalar add
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(m.x));
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].y) : "v"(m.y));
                } else { // packed add
                    asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                }
            }
        }
    }
    float acc = 0.0f;
    for(int i = 0; i < 8; ++i)
        acc += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template<int MODE> float run(float *d, int blocks, int threads, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    const int blocks = 256 * 8, threads = 256, iters = 200; // 8 waves per SIMD
    float *d; hipMalloc(&d, (size_t)blocks * threads * sizeof(float));
    const double pairs = (double)blocks * threads * iters * REP * 8; // (x, y) pairs updated
    const char *names[4] = {"2 x v_fma_f32", "v_pk_fma_f32", "2 x v_add_f32", "v_pk_add_f32"};
    float ms[4] = {run<0>(d, blocks, threads, iters), run<1>(d, blocks, threads, iters), run<2>(d, blocks, threads, iters), run<3>(d, blocks, threads, iters)};
    for(int i = 0; i < 4; ++i)
        printf("%-14s %8.3f ms  %7.2f T pair-updates/s\n", names[i], ms[i], pairs / (ms[i] * 1e-3) / 1e12);
    return 0;
}
