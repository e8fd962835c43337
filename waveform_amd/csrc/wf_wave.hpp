// wf_wave.hpp -- gfx950 kernel of the waveform-display tick (device code only; hipcc).
//
//   waveform_tick_kernel   WAVSource*::tick_waveform for a whole batch of sources (reference src/source_generic.cpp:271-390,
//                          AVX variant src/source_avx.cpp:347-470): a history of `width` points per channel; each tick appends
//                          one point per step_ns = meter_ms / width of newly consumed audio -- the sample nearest to that
//                          time -- shifts the history left by as many, and converts the new points to dBFS (|x|, or the mean
//                          of |L| and |R| in mono display), plus volume normalisation.
//
// The reference pops the ring into a scratch buffer and indexes it backwards from the newest sample
// (temp[total - index]); on the device that is ring[wpos - index], so nothing is copied.  All time arithmetic is the
// reference's 64-bit integer arithmetic (libobs util_mul_div64 behind ns_to_audio_frames / audio_frames_to_ns).
// One workgroup per stream: the old rows are staged in LDS (the shift is in place in memory), the new rows assembled in a
// second LDS area because what is finally stored depends on whether *every* channel's row is all zeros (m_last_silent).
// HBM traffic: read + write of out_ch * width floats per stream and tick plus the few samples picked from the ring.
#pragma once
#include <hip/hip_runtime.h>
#include "wf_tick_phases.hpp"

namespace wf {

struct WaveArgs {
    const float *ring;         // [n_streams * cap_ch] rows of ring_stride floats
    const uint32_t *wpos;      // [n_streams] samples written so far, modulo 2^32
    uint32_t *cend;            // [n_streams] samples consumed so far (the reference's ring holds wpos - cend samples)
    unsigned long long *wts;   // [n_streams] m_waveform_ts
    uint32_t ring_mask, ring_stride;
    uint32_t delay;            // A/V-sync reserve in frames (dtaudio > 0, :291-292)
    const uint32_t *delay_stream;
    float *rows;               // [n_streams][out_ch][width] m_decibels
    uint32_t *stream_flags;
    unsigned long long audio_ts; // m_audio_ts: end-of-audio timestamp of the newest captured sample (ns)
    unsigned long long step_ns;  // (m_meter_ms * 1000000) / width, :299
    uint32_t waveform_samples; // m_waveform_samples
    uint32_t width;            // m_fft_size = m_width
    uint32_t sample_rate;
    uint32_t n_streams, cap_ch, out_ch;
    uint32_t stereo, normalize;
    float vol_comp;
    const float *vol_comp_stream;
    float db_min;
};

constexpr int WAVE_THREADS = 256;

// libobs util_mul_div64 (media-io/audio-io.h helpers are built on it): (num / div) * mul + ((num % div) * mul) / div
// audio_frames_to_ns(sr, frames) = util_mul_div64(frames, 10^9, sr) for frames, sr < 2^32: the quotient and remainder are
// 32-bit divisions; the last term, floor(r * 10^9 / sr) with r < sr, is below 10^9 -- a double-precision estimate corrected
// by one exact 64-bit comparison in each direction (no 64-bit division by a run-time divisor on the device).
WF_DEV unsigned long long frames_to_ns(uint32_t frames, uint32_t sr)
{
    const uint32_t q = frames / sr, r = frames - q * sr;
    const unsigned long long num = (unsigned long long)r * 1000000000ull; // < 2^62
    unsigned long long e = (unsigned long long)((double)r * 1.0e9 / (double)sr);
    while(e * sr > num) --e;          // at most one step each: the estimate is within 1 of the floor
    while((e + 1ull) * sr <= num) ++e;
    return (unsigned long long)q * 1000000000ull + e;
}
// ns_to_audio_frames(sr, ns) = util_mul_div64(ns, sr, 10^9): the divisor is a compile-time constant (multiply-high)
WF_DEV unsigned long long ns_to_frames(unsigned long long ns, uint32_t sr)
{
    const unsigned long long q = ns / 1000000000ull, r = ns - q * 1000000000ull;
    return q * sr + (r * sr) / 1000000000ull;
}

// exact dbfs of the reference (20 * log10f) -- a few hundred points per stream and tick, so the library log is affordable
WF_DEV float wave_dbfs(float mag, float db_min) { return (mag > 0.0f) ? mul_unfused(20.0f, log10f(mag)) : db_min; }

// V = 4: rows whose length is a multiple of 4 floats are 16-byte aligned; every row access is a 16-byte vector and a thread
// handles four consecutive points per step.  V = 1: any width, dword accesses.
template<int V> WF_DEV void wave_ld(const float *p, float (&v)[V])
{
    if constexpr(V == 4) {
        const f4 q = ld4(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else
        v[0] = *p;
}
template<int V> WF_DEV void wave_st(float *p, const float (&v)[V])
{
    if constexpr(V == 4)
        st4(p, f4{v[0], v[1], v[2], v[3]});
    else
        *p = v[0];
}

template<int V>
__global__ __launch_bounds__(WAVE_THREADS) void waveform_tick_kernel(const WaveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wave_lds[]; // old rows [cap_ch][W], then new rows [cap_ch][W]
    const uint32_t stream = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = a.width;
    const uint32_t sflags = a.stream_flags[stream];
    const bool was_silent = (sflags & WF_STREAM_LAST_SILENT) != 0;
    float *rows = a.rows + (size_t)stream * a.out_ch * W;
    const uint32_t disp = a.stereo ? 2u : 1u;
    float dbm[V];
#pragma unroll
    for(int e = 0; e < V; ++e)
        dbm[e] = a.db_min;

    if(sflags & WF_STREAM_HIDDEN) { // !m_show || capture timed out, :279-288
        if(was_silent)
            return;
        for(uint32_t i = V * tid; i < disp * W; i += V * WAVE_THREADS)
            wave_st<V>(rows + i, dbm);
        if(tid == 0)
            a.stream_flags[stream] = sflags | WF_STREAM_LAST_SILENT;
        return;
    }
    const uint32_t wpos = a.wpos[stream];
    const uint32_t cend = a.cend[stream];
    const uint32_t R = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const uint32_t avail = wpos - cend;
    if(avail <= R) // not enough audio in advance, :293-295
        return;
    // the old rows are on their way while the time arithmetic runs
    float *old_rows = wave_lds, *new_rows = wave_lds + (size_t)a.cap_ch * W;
    for(uint32_t i = V * tid; i < a.cap_ch * W; i += V * WAVE_THREADS) {
        float v[V];
        wave_ld<V>(rows + i, v);
        wave_st<V>(old_rows + i, v);
    }
    const uint32_t max_size = a.waveform_samples + R;
    const uint32_t total = avail < max_size ? avail : max_size; // :303-304
    const uint32_t sr = a.sample_rate;
    const unsigned long long start_ts = a.audio_ts - frames_to_ns(total, sr);
    const unsigned long long stop_ts = a.audio_ts - frames_to_ns(R, sr);
    if(start_ts >= a.audio_ts || stop_ts > a.audio_ts) {
        // timestamp rollover, :316-317 (a tick before any audio has a timestamp).  The rows are untouched, but the reference
        // has already trimmed the ring to max_size on its way here (:303-304), and the samples it dropped stay dropped
        if(tid == 0 && avail > max_size)
            a.cend[stream] = wpos - max_size;
        return;
    }
    unsigned long long wts = a.wts[stream];
    if(wts < start_ts)
        wts = start_ts; // catch up
    if(wts > stop_ts && (wts - stop_ts) > a.step_ns)
        wts = start_ts; // fix desync
    // points this tick adds: ts = wts + i * step_ns while ts < stop_ts, at most W (:322-331)
    uint32_t counts = 0;
    if(wts < stop_ts) {
        const unsigned long long c = (stop_ts - wts + a.step_ns - 1ull) / a.step_ns;
        counts = c < (unsigned long long)W ? (uint32_t)c : W;
    }
    __syncthreads();
    // assemble the rotated rows (:332) and look for a non-zero value (:334-343)
    int nz0 = 0, nz1 = 0;
    const uint32_t keep = W - counts;
    for(uint32_t c = 0; c < a.cap_ch; ++c) {
        const float *x = a.ring + ((size_t)stream * a.cap_ch + c) * a.ring_stride;
        int nz = 0;
        for(uint32_t i0 = V * tid; i0 < W; i0 += V * WAVE_THREADS) {
            float v[V];
#pragma unroll
            for(int e = 0; e < V; ++e) {
                const uint32_t i = i0 + (uint32_t)e;
                if(i < keep)
                    v[e] = old_rows[c * W + i + counts];
                else {
                    const unsigned long long ts = wts + (unsigned long long)(i - keep) * a.step_ns;
                    unsigned long long index = ns_to_frames(a.audio_ts - ts, sr);
                    const unsigned long long lo = (unsigned long long)R + 1ull, hi = total;
                    index = index < lo ? lo : (hi < index ? hi : index);
                    v[e] = x[(wpos - (uint32_t)index) & a.ring_mask]; // temp[total - index]
                }
                nz |= (v[e] != 0.0f) ? 1 : 0;
            }
            wave_st<V>(new_rows + c * W + i0, v);
        }
        if(c == 0) nz0 = nz; else nz1 = nz;
    }
    const int any0 = __syncthreads_or(nz0);
    const int any1 = a.cap_ch > 1 ? __syncthreads_or(nz1) : 0;
    const bool all_silent = !any0 && (a.cap_ch == 1 || !any1); // every channel's row is zeros -> m_last_silent (:345-349)
    if(tid == 0) {
        a.cend[stream] = wpos - R;                               // everything but the reserve has been popped, :321
        a.wts[stream] = wts + (unsigned long long)counts * a.step_ns; // :351
        a.stream_flags[stream] = (sflags & ~WF_STREAM_LAST_SILENT) | (all_silent ? WF_STREAM_LAST_SILENT : 0u);
    }
    if(all_silent) { // :353-359
        for(uint32_t i = V * tid; i < disp * W; i += V * WAVE_THREADS)
            wave_st<V>(rows + i, dbm);
        // a captured channel that is not displayed keeps its (rotated) raw history
        for(uint32_t c = disp; c < a.cap_ch; ++c)
            for(uint32_t i = V * tid; i < W; i += V * WAVE_THREADS) {
                float v[V];
                wave_ld<V>(new_rows + c * W + i, v);
                wave_st<V>(rows + c * W + i, v);
            }
        return;
    }
    const float comp = a.normalize ? (a.vol_comp_stream ? a.vol_comp_stream[stream] : a.vol_comp) : 0.0f;
    for(uint32_t i0 = V * tid; i0 < W; i0 += V * WAVE_THREADS) {
        float r0[V], r1[V], o1[V];
        wave_ld<V>(new_rows + i0, r0);
        if(a.cap_ch > 1)
            wave_ld<V>(new_rows + W + i0, r1);
        if(a.out_ch > a.cap_ch)             // one captured channel shown twice: row 1 is row 0 *before* its new points are
            wave_st<V>(rows + W + i0, r0);  // converted, and its own count is 0, so they stay raw (:361-362, :364-368)
#pragma unroll
        for(int e = 0; e < V; ++e) {
            const bool fresh = i0 + (uint32_t)e >= keep;
            const float s1 = a.cap_ch > 1 ? r1[e] : r0[e];
            o1[e] = s1;
            if(fresh) {
                if(a.stereo || a.cap_ch == 1)
                    r0[e] = wave_dbfs(__builtin_fabsf(r0[e]), a.db_min);
                else
                    r0[e] = wave_dbfs(mul_unfused(add_unfused(__builtin_fabsf(r0[e]), __builtin_fabsf(s1)), 0.5f), a.db_min);
                if(a.normalize)
                    r0[e] = add_unfused(r0[e], comp);
                if(a.stereo && a.cap_ch > 1) {
                    o1[e] = wave_dbfs(__builtin_fabsf(s1), a.db_min);
                    if(a.normalize)
                        o1[e] = add_unfused(o1[e], comp);
                }
            }
        }
        wave_st<V>(rows + i0, r0);
        if(a.cap_ch > 1)
            wave_st<V>(rows + W + i0, o1); // mono display: the raw history of channel 1
    }
}

} // namespace wf
