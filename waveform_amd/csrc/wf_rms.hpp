// wf_rms.hpp -- gfx950 kernels of the volume-normalisation producer (device code only; hipcc).
//
// What they replace (reference): the RMS part of WAVSource::capture_audio (src/source.cpp:1842-1871: per frame the
// largest |sample| of the captured channels, squared, appended to m_rms_sync_buf), WAVSource::sync_rms_buffer
// (src/source.cpp:810-835: everything older than the A/V-sync point moves into the circular m_input_rms_buf of
// m_input_rms_size = sample_rate & -16 values) and WAVSource*::update_input_rms (src/source_generic.cpp:392-403,
// AVX src/source_avx.cpp:325-345: m_input_rms = sqrt(sum / m_input_rms_size)), which WAVSource::tick runs before the
// spectrum when m_normalize_volume (src/source.cpp:1330-1331).
//
// Device form.  Every stream has a second ring, of squared peaks, written at the same positions as its audio rings.
// m_input_rms_buf's contents are the m_input_rms_size values that end at the consumption point `rend` (monotonic, like
// the meter's), so m_input_rms is a function of ring[rend - size, rend).  A second of audio per stream and tick would be
// 188 KB of HBM reads for a 6.4 KB hop, so the sum is kept in two levels: when a push completes a block of
// RMS_BLOCK frames (aligned in absolute frame index) its sum is stored once; the tick adds the <= 188 block sums
// inside the window and the raw values of the two ragged edges (< 2 * RMS_BLOCK): ~3 KB per stream and tick, no drift
// (every tick's sum is rebuilt from the stored terms, in a fixed order).
#pragma once
#include <hip/hip_runtime.h>
#include "wf_tick_phases.hpp"
#include "wf_synth.h"

namespace wf {

constexpr uint32_t RMS_BLOCK = 256;      // frames per stored partial sum
constexpr uint32_t RMS_BLOCK_SHIFT = 8;

// fixed-order sum over a wavefront
WF_DEV float wave_sum(float v)
{
#pragma unroll
    for(int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// src: [count*cap_ch][frames] (nullptr: silence); squared peaks of streams [first, first+count) at wpos.. (before wpos advances)
__global__ void rms_push_kernel(float *rms_ring, const uint32_t *wpos, uint32_t rms_cap, uint32_t cap_ch, uint32_t first,
                                const float *src, uint32_t frames)
{
    const uint32_t s = blockIdx.y;
    const uint32_t w = wpos[first + s];
    float *dst = rms_ring + (size_t)(first + s) * rms_cap;
    for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += gridDim.x * blockDim.x) {
        float val = 0.0f;
        if(src) {
            val = __builtin_fabsf(src[(size_t)s * cap_ch * frames + i]);
            if(cap_ch > 1)
                val = __builtin_fmaxf(__builtin_fabsf(src[((size_t)s * cap_ch + 1) * frames + i]), val);
        }
        dst[(w + i) & (rms_cap - 1)] = val * val;
    }
}

// the same for audio generated on the device by the counter hash (wf_hip_push_synth)
__global__ void rms_synth_kernel(float *rms_ring, const uint32_t *wpos, uint32_t rms_cap, uint32_t cap_ch, uint32_t first,
                                 uint64_t seed, uint32_t stream_id0, uint64_t index0, uint32_t frames)
{
    const uint32_t s = blockIdx.y;
    const uint32_t w = wpos[first + s];
    float *dst = rms_ring + (size_t)(first + s) * rms_cap;
    const uint64_t key0 = wf_synth_key(seed, stream_id0 + s, 0), key1 = wf_synth_key(seed, stream_id0 + s, 1);
    for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += gridDim.x * blockDim.x) {
        float val = __builtin_fabsf(wf_synth_sample(key0, index0 + i));
        if(cap_ch > 1)
            val = __builtin_fmaxf(__builtin_fabsf(wf_synth_sample(key1, index0 + i)), val);
        dst[(w + i) & (rms_cap - 1)] = val * val;
    }
}

// Partial sums of the blocks this push has completed: block k (frames [k*B, (k+1)*B)) is complete once (k+1)*B <= w + frames
// and was not before if (k+1)*B > w.  One wavefront per candidate block; runs after rms_push_kernel, before wpos advances.
__global__ __launch_bounds__(64) void rms_block_kernel(const float *rms_ring, float *bsum, const uint32_t *wpos, uint32_t rms_cap,
                                                       uint32_t first, uint32_t frames)
{
    const uint32_t stream = first + blockIdx.y;
    const uint32_t w = wpos[stream];
    const uint32_t k = (w >> RMS_BLOCK_SHIFT) + blockIdx.x;          // modulo 2^24 through the shift below
    const uint32_t end = (k + 1u) << RMS_BLOCK_SHIFT;                // modulo 2^32
    const uint32_t dist = end - w;                                    // 1..frames+B: frames from w to the block's end
    if(dist == 0u || dist > frames)
        return;
    const float *src = rms_ring + (size_t)stream * rms_cap;
    const uint32_t pos = ((k << RMS_BLOCK_SHIFT) + 4u * threadIdx.x) & (rms_cap - 1);
    const f4 v = ld4(src + pos);
    const float sum = wave_sum((v.x + v.y) + (v.z + v.w));
    if(threadIdx.x == 0)
        bsum[(size_t)stream * (rms_cap >> RMS_BLOCK_SHIFT) + (k & ((rms_cap >> RMS_BLOCK_SHIFT) - 1))] = sum;
}

// Feed mode (wf_hip_enable_input_rms_feed): the host hands over the squared peaks themselves -- what sync_rms_buffer moves
// from m_rms_sync_buf into m_input_rms_buf this tick (src/source.cpp:810-835) -- a different number of values per stream.
// sq: [count][max_frames], frames[count].  One workgroup per stream appends them at the stream's consumption point
// `rend`, stores the sums of the blocks that became complete, and advances rend.
__global__ __launch_bounds__(256) void rms_feed_ragged_kernel(float *rms_ring, float *bsum, uint32_t *rend, uint32_t rms_cap, uint32_t first,
                                                              const float *sq, const uint32_t *frames, uint32_t max_frames)
{
    const uint32_t s = blockIdx.x, stream = first + s;
    const uint32_t n = frames[s] < max_frames ? frames[s] : max_frames;
    if(n == 0)
        return;
    const uint32_t w = rend[stream];
    float *ring = rms_ring + (size_t)stream * rms_cap;
    const float *src = sq + (size_t)s * max_frames;
    const uint32_t skip = n > rms_cap ? n - rms_cap : 0u; // longer than the ring: only the tail survives
    for(uint32_t i = skip + threadIdx.x; i < n; i += blockDim.x)
        ring[(w + i) & (rms_cap - 1)] = src[i];
    __syncthreads(); // the workgroup's own stores are visible to it
    // blocks completed by this feed: block k ends at (k + 1) * B; complete now, not before, if 0 < end - w <= n
    float *bs = bsum + (size_t)stream * (rms_cap >> RMS_BLOCK_SHIFT);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for(uint32_t j = wave;; j += blockDim.x >> 6) {
        const uint32_t k = (w >> RMS_BLOCK_SHIFT) + j;
        const uint32_t end = (k + 1u) << RMS_BLOCK_SHIFT;
        const uint32_t dist = end - w; // 1..B for j == 0, then + B per step
        if(dist > n)
            break;
        const f4 v = ld4(ring + (((k << RMS_BLOCK_SHIFT) + 4u * lane) & (rms_cap - 1)));
        const float sum = wave_sum((v.x + v.y) + (v.z + v.w));
        if(lane == 0)
            bs[k & ((rms_cap >> RMS_BLOCK_SHIFT) - 1)] = sum;
    }
    __syncthreads();
    if(threadIdx.x == 0)
        rend[stream] = w + n;
}

struct RmsArgs {
    const float *rms_ring;     // [n_streams][rms_cap] squared peaks
    const float *bsum;         // [n_streams][rms_cap / RMS_BLOCK] sums of completed blocks
    const uint32_t *wpos;      // [n_streams]
    uint32_t *rend;            // [n_streams] consumption point of sync_rms_buffer
    uint32_t rms_cap;
    uint32_t size;             // m_input_rms_size
    uint32_t delay;
    const uint32_t *delay_stream;
    float *input_rms;          // [n_streams] m_input_rms
    float *vol_comp;           // [n_streams] min(m_volume_target - dbfs(m_input_rms), m_max_gain), read by the tick kernel
    float volume_target, max_gain, db_min;
    uint32_t n_streams;
    uint32_t feed;             // 1: rend is advanced by rms_feed_ragged_kernel, not derived from the audio position
};

// update_input_rms for every stream: one wavefront per stream
__global__ __launch_bounds__(64) void input_rms_kernel(const RmsArgs a)
{
    const uint32_t stream = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t wpos = a.wpos[stream];
    uint32_t rend = a.rend[stream];
    const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const uint32_t cand = wpos - delay;
    if(!a.feed && (int32_t)(cand - rend) > 0) // sync_rms_buffer: everything older than the sync point is consumed; never moves back
        rend = cand;
    if(!a.feed && lane == 0)
        a.rend[stream] = rend;

    const uint32_t mask = a.rms_cap - 1, nblk_mask = (a.rms_cap >> RMS_BLOCK_SHIFT) - 1;
    const float *ring = a.rms_ring + (size_t)stream * a.rms_cap;
    const float *bs = a.bsum + (size_t)stream * (a.rms_cap >> RMS_BLOCK_SHIFT);
    const uint32_t lo = rend - a.size, hi = rend;                     // the window [lo, hi), positions modulo 2^32
    const uint32_t ka = (lo + (RMS_BLOCK - 1u)) >> RMS_BLOCK_SHIFT;   // first block that starts inside the window
    const uint32_t kb = hi >> RMS_BLOCK_SHIFT;                        // block that holds hi (its start is the tail's start)
    const uint32_t head = (ka << RMS_BLOCK_SHIFT) - lo;               // 0..B-1 raw values before the first whole block
    const uint32_t tail = hi - (kb << RMS_BLOCK_SHIFT);               // 0..B-1 raw values after the last whole block
    const uint32_t n_full = (kb - ka) & 0x00FFFFFFu;                  // whole blocks (block indices are modulo 2^24)

    float acc = 0.0f;
    for(uint32_t i = lane; i < n_full; i += 64)
        acc += bs[(ka + i) & nblk_mask];
    for(uint32_t i = lane; i < head; i += 64)
        acc += ring[(lo + i) & mask];
    for(uint32_t i = lane; i < tail; i += 64)
        acc += ring[((kb << RMS_BLOCK_SHIFT) + i) & mask];
    const float sum = wave_sum(acc);
    if(lane == 0) {
        const float rms = __fsqrt_rn(__fdiv_rn(sum, (float)a.size)); // std::sqrt(sum / m_input_rms_size)
        a.input_rms[stream] = rms;
        const float rms_db = (rms > 0.0f) ? mul_unfused(20.0f, log10f(rms)) : a.db_min; // dbfs(), src/source.hpp:293-299
        const float comp = a.volume_target - rms_db;                  // src/source_generic.cpp:163
        a.vol_comp[stream] = comp < a.max_gain ? comp : a.max_gain;
    }
}

} // namespace wf
