// wf_kernels.hpp -- gfx950 kernels of libwaveform_hip.so (device code only; hipcc).
//
//   spectrum_tick_kernel<G, SPW, ALIGNED>   the fused per-tick pass: ring fetch -> window -> r2c FFT
//                                           in LDS -> |X| -> slope -> temporal smoothing -> dBFS
//                                           (-> volume normalisation -> roll-off), one HBM pass.
//                                           Replaces WAVSource*::tick_spectrum (reference
//                                           src/source_generic.cpp:26-180) for a whole batch of sources.
//   ring_push_kernel / ring_synth_kernel    CircularBuffer::push_back for every (stream, channel)
//                                           (reference src/source.cpp:1873-1886).
//   fill_kernel                             state initialisation (reference src/source.cpp:1170-1182).
//
// Work decomposition: a spectrum (one channel of one stream) is owned by T = G::T threads
// (one wavefront for N <= 4096); a workgroup holds SPW spectra, laid out so that the
// channels of a stream share a workgroup (needed by the mono mixdown, reference :150-154).
#pragma once
#include <hip/hip_runtime.h>
#include "wf_geometry.hpp"
#include "wf_tick_phases.hpp"
#include "wf_synth.h"

namespace wf {

template<class G> __device__ __forceinline__ void spectrum_sync()
{
    // T == 64: the spectrum lives in one wavefront; LDS operations of a wave execute in
    // program order, so a scheduling fence is all that is needed between exchange phases.
    if constexpr(G::T > 64)
        __syncthreads();
    else
        __builtin_amdgcn_wave_barrier();
}

template<class G, int SPW, bool ALIGNED>
__global__ __launch_bounds__(G::T *SPW) void spectrum_tick_kernel(const TickArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int T = G::T, M = G::M, P = G::P;
    const int tid = (int)threadIdx.x;
    const int sub = tid / T; // which spectrum of the workgroup
    const int t = tid % T;   // thread within the spectrum
    const uint32_t n_spec = a.n_streams * a.cap_ch;
    const uint32_t spec = blockIdx.x * SPW + (uint32_t)sub;
    const bool active = spec < n_spec;
    const uint32_t stream = active ? spec / a.cap_ch : 0u;
    const uint32_t ch = active ? spec % a.cap_ch : 0u;

    cf *lds = reinterpret_cast<cf *>(smem_raw) + (size_t)sub * G::LDS_CF;
    const float *x = a.ring + (size_t)(active ? spec : 0u) * a.ring_cap;
    const uint32_t start = (a.wpos[stream] - a.delay - (uint32_t)G::N) & a.ring_mask;
    float *ts = a.tsmooth + (size_t)(active ? spec : 0u) * M;

    cf v[P];
    float mag[P];

    if(active)
        p1_fetch_pass1<G, ALIGNED>(a, t, x, start, lds);
    spectrum_sync<G>();
    if(active)
        p2_read<G>(t, lds, v);
    spectrum_sync<G>();
    if(active)
        p2_pass2_write<G>(a, t, lds, v);
    spectrum_sync<G>();
    if(active)
        p3_read<G>(t, lds, v);
    spectrum_sync<G>();
    if(active)
        p3_pass3_write<G>(t, lds, v);
    spectrum_sync<G>();
    if(active)
        p4_split_smooth<G>(a, t, lds, ts, mag);

    if(a.mode & WF_MODE_MONO_MIX) {
        // reference :150-154: dB[0][i] = dbfs((dB[0][i] + dB[1][i]) * 0.5f); channel 1 keeps its linear
        // magnitudes in the reference (never displayed) and is not written here.
        // The two channels of a stream are adjacent spectra of this workgroup (SPW is even).
        float *xch = reinterpret_cast<float *>(lds); // reuse this spectrum's exchange buffer as float[M]
        __syncthreads();
        if(active && ch == 1) {
#pragma unroll
            for(int u = 0; u < P / 4; ++u) {
                const int k0 = 4 * (t + T * u);
                *reinterpret_cast<f4 *>(xch + k0) = f4{mag[4 * u], mag[4 * u + 1], mag[4 * u + 2], mag[4 * u + 3]};
            }
        }
        __syncthreads();
        if(active && ch == 0) {
            const float *other = reinterpret_cast<const float *>(lds + G::LDS_CF); // the ch-1 spectrum's buffer
#pragma unroll
            for(int u = 0; u < P / 4; ++u) {
                const int k0 = 4 * (t + T * u);
                const f4 o = *reinterpret_cast<const f4 *>(other + k0);
                mag[4 * u] = (mag[4 * u] + o.x) * 0.5f;
                mag[4 * u + 1] = (mag[4 * u + 1] + o.y) * 0.5f;
                mag[4 * u + 2] = (mag[4 * u + 2] + o.z) * 0.5f;
                mag[4 * u + 3] = (mag[4 * u + 3] + o.w) * 0.5f;
            }
            p4_db_store<G>(a, t, a.decibels + ((size_t)stream * a.out_ch) * M, mag);
        }
    } else if(active) {
        // stereo: both channels converted; single captured channel shown as stereo: channel 0 is
        // duplicated into channel 1 (reference :141-142)
        p4_db_store<G>(a, t, a.decibels + ((size_t)stream * a.out_ch + ch) * M, mag);
        if(a.out_ch > a.cap_ch)
            p4_db_store<G>(a, t, a.decibels + ((size_t)stream * a.out_ch + 1) * M, mag);
    }
}

// ---- ring maintenance ---------------------------------------------------------------------------
// src: [count*cap_ch][frames]; appends to the rings of streams [first, first+count)
__global__ void ring_push_kernel(float *ring, const uint32_t *wpos, uint32_t ring_cap, uint32_t cap_ch, uint32_t first,
                                 const float *src, uint32_t frames)
{
    const uint32_t row = blockIdx.y; // (stream - first) * cap_ch + ch
    const uint32_t stream = first + row / cap_ch;
    const uint32_t w = wpos[stream];
    float *dst = ring + ((size_t)stream * cap_ch + row % cap_ch) * ring_cap;
    for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += gridDim.x * blockDim.x)
        dst[(w + i) & (ring_cap - 1)] = src ? src[(size_t)row * frames + i] : 0.0f;
}

__global__ void ring_synth_kernel(float *ring, const uint32_t *wpos, uint32_t ring_cap, uint32_t cap_ch, uint32_t first,
                                  uint64_t seed, uint32_t stream_id0, uint64_t index0, uint32_t frames)
{
    const uint32_t row = blockIdx.y;
    const uint32_t s = row / cap_ch, c = row % cap_ch;
    const uint32_t stream = first + s;
    const uint32_t w = wpos[stream];
    float *dst = ring + ((size_t)stream * cap_ch + c) * ring_cap;
    const uint64_t key = wf_synth_key(seed, stream_id0 + s, c);
    for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += gridDim.x * blockDim.x)
        dst[(w + i) & (ring_cap - 1)] = wf_synth_sample(key, index0 + i);
}

__global__ void wpos_advance_kernel(uint32_t *wpos, uint32_t first, uint32_t count, uint32_t frames)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < count)
        wpos[first + i] += frames;
}

__global__ void fill_f32_kernel(float *p, size_t n, float v)
{
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v;
}
__global__ void fill_u32_kernel(uint32_t *p, size_t n, uint32_t v)
{
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v;
}

} // namespace wf
