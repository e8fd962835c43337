// stream_pattern.hip -- development microbenchmark (not part of the product): what the memory system delivers for the
// tick kernel's own access pattern with no FFT in the way.  One workgroup per stream, per spectrum:
//   read  N floats of a ring row at a moving offset (window), read N/2 floats of state,
//   write N/2 floats of state, write N/2 floats of dB                                     = 10*N bytes
// as 16-byte vectors per lane, all loads issued before the first use.  Prints GB/s for several workgroup shapes so the
// fused kernel's roofline.frac can be read against "the same bytes, no math".
//   hipcc --offload-arch=gfx950 -O3 -o build/stream_pattern tools/ubench/stream_pattern.hip && build/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(16) f4 { float x, y, z, w; };

#define CHECK(x)                                                              \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if(e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            exit(1);                                                          \
        }                                                                     \
    } while(0)

typedef float v4f __attribute__((ext_vector_type(4)));
template<bool NT> __device__ inline f4 ld(const f4 *p)
{
    if(NT) {
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
        return f4{v.x, v.y, v.z, v.w};
    }
    return *p;
}
template<bool NT> __device__ inline void st_(f4 *p, f4 v)
{
    if(NT)
        __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(p));
    else
        *p = v;
}

// T threads per spectrum, SPW spectra per workgroup, N samples per window; NT: state and rows with the non-temporal hint
// (what the fused kernel does since round 2)
template<int N, int T, int SPW, bool NT = false>
__global__ __launch_bounds__(T *SPW) void pattern_kernel(const float *ring, size_t ring_cap, unsigned start, float *state, float *db, float g)
{
    constexpr int M = N / 2;
    constexpr int WV = N / 4 / T; // 16-byte window vectors per thread
    constexpr int SV = M / 4 / T; // 16-byte state vectors per thread
    const int sub = threadIdx.x / T, t = threadIdx.x % T;
    const size_t spec = (size_t)blockIdx.x * SPW + sub;
    const f4 *x = reinterpret_cast<const f4 *>(ring + spec * ring_cap + start);
    f4 *st = reinterpret_cast<f4 *>(state + spec * M);
    f4 *out = reinterpret_cast<f4 *>(db + spec * M);
    f4 w[WV], s[SV];
#pragma unroll
    for(int i = 0; i < WV; ++i)
        w[i] = x[t + T * i];
#pragma unroll
    for(int i = 0; i < SV; ++i)
        s[i] = ld<NT>(st + t + T * i);
#pragma unroll
    for(int i = 0; i < SV; ++i) {
        const f4 a = w[2 * i], b = w[2 * i + 1];
        f4 m{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
        f4 o{g * s[i].x + m.x, g * s[i].y + m.y, g * s[i].z + m.z, g * s[i].w + m.w};
        st_<NT>(st + t + T * i, o);
        st_<NT>(out + t + T * i, f4{o.x * 2.0f, o.y * 2.0f, o.z * 2.0f, o.w * 2.0f});
    }
}

template<int N, int T, int SPW, bool NT = false> void run(size_t n_spec, int ticks, int hop, int lds_bytes)
{
    const size_t ring_cap = (size_t)N + (size_t)hop * (ticks + 2);
    float *ring, *state, *db;
    CHECK(hipMalloc(&ring, n_spec * ring_cap * 4));
    CHECK(hipMalloc(&state, n_spec * (N / 2) * 4));
    CHECK(hipMalloc(&db, n_spec * (N / 2) * 4));
    CHECK(hipMemset(ring, 0, n_spec * ring_cap * 4));
    CHECK(hipMemset(state, 0, n_spec * (N / 2) * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto k = pattern_kernel<N, T, SPW, NT>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const dim3 grid((unsigned)(n_spec / SPW)), block(T * SPW);
    float best = 1e30f;
    for(int rep = 0; rep < 4; ++rep) {
        if(rep == 1) {
            // the device's clocks settle after 15-20 ms of load (profiles/r02j_warmup.txt): 40 ms of the same launches, untimed,
            // before the three regions that count
            const int warm = (int)(40.0f / best) + 1;
            for(int i = 0; i < warm; ++i)
                hipLaunchKernelGGL(k, grid, block, lds_bytes, 0, ring, ring_cap, (unsigned)((i % ticks) * hop), state, db, 0.65f);
            best = 1e30f;
        }
        CHECK(hipEventRecord(e0));
        for(int i = 0; i < ticks; ++i)
            hipLaunchKernelGGL(k, grid, block, lds_bytes, 0, ring, ring_cap, (unsigned)(i * hop), state, db, 0.65f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if(ms / ticks < best) // (rep 0: the cold estimate the lead-in is sized by, discarded above)
            best = ms / ticks;
    }
    const double bytes = 10.0 * N * n_spec;
    printf("N=%5d T=%3d SPW=%d lds=%6d B nt=%d spectra=%zu  %.1f us/tick  %.0f GB/s  (%.1f %% of 8 TB/s)\n", N, T, SPW, lds_bytes, (int)NT, n_spec, best * 1e3,
           bytes / best / 1e6, bytes / best / 1e6 / 80.0);
    CHECK(hipFree(ring));
    CHECK(hipFree(state));
    CHECK(hipFree(db));
}

int main()
{
    // the tick kernel's launch shapes: dynamic LDS sized like the real kernel limits workgroups per CU the same way
    run<4096, 128, 2>(8192, 40, 800, 0);
    run<4096, 128, 2>(8192, 40, 800, 36 * 1024);  // 4 workgroups per CU, as the fused kernel
    run<4096, 128, 2>(8192, 40, 800, 52 * 1024);  // 3 per CU
    run<4096, 256, 1>(8192, 40, 800, 0);
    run<1024, 64, 2>(32768, 40, 800, 0);
    run<1024, 64, 2>(32768, 40, 800, 10 * 1024);
    run<16384, 512, 2>(2048, 40, 800, 139 * 1024); // 1 per CU
    run<16384, 512, 2>(2048, 40, 800, 0);
    run<4096, 128, 2>(32768, 40, 800, 36 * 1024);  // 4x the streams: launch ramp amortised, working set past the Infinity Cache
    // the same with state and rows non-temporal
    run<4096, 128, 2, true>(8192, 40, 800, 36 * 1024);
    run<4096, 128, 2, true>(16384, 40, 800, 36 * 1024);
    run<4096, 128, 2>(16384, 40, 800, 36 * 1024);
    run<4096, 128, 2, true>(32768, 40, 800, 36 * 1024);
    run<16384, 512, 1, true>(2048, 40, 800, 74 * 1024);
    return 0;
}
