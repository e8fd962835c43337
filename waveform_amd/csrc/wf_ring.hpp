// wf_ring.hpp -- gfx950 kernels around the device rings and the per-stream words (device code only; hipcc; included by the
// entry-point translation unit wf_hip.hip alone -- these kernels are not templates).
//
//   ring_push_kernel / ring_push_ragged_kernel / ring_synth_kernel
//                                           CircularBuffer::push_back for every (stream, channel) (reference
//                                           src/source.cpp:1873-1886, src/circular_buffer.hpp:42-63)
//   wpos_advance_kernel, set_hidden_kernel  the streams' write positions and show / hide / timeout / paused / starved bits
//   fill_f32_kernel / fill_u32_kernel       state initialisation (reference src/source.cpp:1170-1182)
#pragma once
#include <hip/hip_runtime.h>
#include "wf_tick_phases.hpp"
#include "wf_synth.h"
#include "wf_hip.h"

namespace wf {

// ---- ring maintenance ---------------------------------------------------------------------------
// src: [count*cap_ch][frames]; appends to the rings of streams [first, first+count)
__global__ void ring_push_kernel(float *ring, const uint32_t *wpos, uint32_t ring_cap, uint32_t ring_stride, uint32_t cap_ch, uint32_t first,
                                 const float *src, uint32_t frames)
{
    const uint32_t row = blockIdx.y; // (stream - first) * cap_ch + ch
    const uint32_t stream = first + row / cap_ch;
    const uint32_t w = wpos[stream];
    float *dst = ring + ((size_t)stream * cap_ch + row % cap_ch) * ring_stride;
    const uint32_t skip = frames > ring_cap ? frames - ring_cap : 0u; // a packet longer than the ring: only its tail survives
    for(uint32_t i = skip + blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += gridDim.x * blockDim.x)
        dst[(w + i) & (ring_cap - 1)] = src ? src[(size_t)row * frames + i] : 0.0f;
}

// the same with a frame count per stream (sources of a plugin batch hand over hops of different lengths): src is
// [count*cap_ch][max_frames], frames[count]; every stream's write position advances by its own count
__global__ void ring_push_ragged_kernel(float *ring, uint32_t *wpos, uint32_t *flags, uint32_t ring_cap, uint32_t ring_stride, uint32_t cap_ch,
                                        uint32_t first, const float *src, const uint32_t *frames, uint32_t max_frames)
{
    const uint32_t s = blockIdx.y; // stream - first
    const uint32_t stream = first + s;
    const uint32_t n = frames[s] < max_frames ? frames[s] : max_frames;
    const uint32_t w = wpos[stream];
    const uint32_t skip = n > ring_cap ? n - ring_cap : 0u;
    for(uint32_t c = 0; c < cap_ch; ++c) {
        float *dst = ring + ((size_t)stream * cap_ch + c) * ring_stride;
        const float *from = src + ((size_t)s * cap_ch + c) * max_frames;
        for(uint32_t i = skip + threadIdx.x; i < n; i += blockDim.x)
            dst[(w + i) & (ring_cap - 1)] = from[i];
    }
    __syncthreads(); // every thread has read the old position
    if(threadIdx.x == 0 && n > 0) {
        wpos[stream] = w + n;
        if(w + n < w)
            flags[stream] |= WF_STREAM_WRAPPED;
    }
}

__global__ void ring_synth_kernel(float *ring, const uint32_t *wpos, uint32_t ring_cap, uint32_t ring_stride, uint32_t cap_ch, uint32_t first,
                                  uint64_t seed, uint32_t stream_id0, uint64_t index0, uint32_t frames)
{
    const uint32_t row = blockIdx.y;
    const uint32_t s = row / cap_ch, c = row % cap_ch;
    const uint32_t stream = first + s;
    const uint32_t w = wpos[stream];
    float *dst = ring + ((size_t)stream * cap_ch + c) * ring_stride;
    const uint64_t key = wf_synth_key(seed, stream_id0 + s, c);
    const uint32_t skip = frames > ring_cap ? frames - ring_cap : 0u;
    for(uint32_t i = skip + blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += gridDim.x * blockDim.x)
        dst[(w + i) & (ring_cap - 1)] = wf_synth_sample(key, index0 + i);
}

__global__ void wpos_advance_kernel(uint32_t *wpos, uint32_t *flags, uint32_t first, uint32_t count, uint32_t frames)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < count) {
        const uint32_t w = wpos[first + i], n = w + frames;
        wpos[first + i] = n;
        if(n < w) // 2^32 samples (a day at 48 kHz): the position wraps, the stream has long had all the audio any delay asks for
            flags[first + i] |= WF_STREAM_WRAPPED;
    }
}

// show()/hide()/capture timeout: set or clear WF_STREAM_HIDDEN (mask 1: hidden, 2: capture timed out), keep m_last_silent
__global__ void set_hidden_kernel(uint32_t *flags, uint32_t first, uint32_t count, const uint8_t *mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < count) {
        const uint32_t f = flags[first + i] & ~(WF_STREAM_HIDDEN | WF_STREAM_TIMEOUT | WF_STREAM_PAUSED | WF_STREAM_STARVED);
        const uint32_t m = mask[i];
        flags[first + i] = m == WF_HIP_PAUSED    ? (f | WF_STREAM_PAUSED)
                           : m == WF_HIP_STARVED ? (f | WF_STREAM_STARVED)
                                                 : (f | (m ? WF_STREAM_HIDDEN : 0u) | (m == WF_HIP_HIDDEN_TIMEOUT ? WF_STREAM_TIMEOUT : 0u));
    }
}

// test aid (wf_hip_debug_age): every sample counter of the streams moves on by `frames`
__global__ void age_kernel(uint32_t *a, uint32_t *b, uint32_t *c, uint32_t *d, uint32_t first, uint32_t count, uint32_t frames)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < count) {
        if(a) a[first + i] += frames;
        if(b) b[first + i] += frames;
        if(c) c[first + i] += frames;
        if(d) d[first + i] += frames;
    }
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
