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
// One wavefront per stream (see waveform_tick_kernel).  HBM traffic: read + write of out_ch * width floats per stream and tick plus the few samples picked from the ring.
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
    const unsigned long long *audio_ts_stream; // per stream instead (wf_hip_set_stream_audio_ts), or nullptr
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

// One WAVEFRONT per stream, WAVE_STREAMS of them per workgroup, no LDS and no barrier: lane l owns the points l + 64 k of every
// row.  The shift is in place in memory -- rows[i] = rows[i + counts] -- which a single wavefront may do in ascending batches:
// a batch's loads (all 64 lanes) are complete before its stores are issued, and what a batch stores lies below everything a
// later batch reads.  What is stored depends on whether *every* channel's rotated row is all zeros (m_last_silent), so the
// rows are looked at before anything is written: SINGLE (width <= 64 U, every default) keeps the whole stream in registers
// between the look and the store; wider rows are scanned first and fetched again (from L2) batch by batch.
// (Round 2's kernel staged the rows of a stream in LDS with one 256-thread workgroup per stream and three barriers: eight
// streams in flight per CU at most, 0.25-0.36 of the HBM roofline.)
constexpr int WAVE_STREAMS = 4;
constexpr int WAVE_THREADS = 64 * WAVE_STREAMS;
constexpr int WAVE_U = 16; // points per lane and batch: widths up to 1024 in one

struct WavePlan { // per stream, identical in every lane
    const float *x0, *x1; // ring rows of the captured channels
    float *rows;
    uint32_t W, keep, counts, wpos, R, total, sr, mask;
    unsigned long long wts, audio_ts, step_ns;
    bool two;
};
// the value of point i of the rotated row of channel c (reference :332 + :322-331): old history or the sample nearest to its time
WF_DEV void wave_gather(const WavePlan &p, uint32_t i, bool live, float &v0, float &v1)
{
    v0 = v1 = 0.0f;
    if(!live)
        return;
    if(i < p.keep) {
        v0 = p.rows[i + p.counts];
        if(p.two)
            v1 = p.rows[p.W + i + p.counts];
    } else {
        const unsigned long long ts = p.wts + (unsigned long long)(i - p.keep) * p.step_ns;
        unsigned long long index = ns_to_frames(p.audio_ts - ts, p.sr);
        const unsigned long long lo = (unsigned long long)p.R + 1ull, hi = p.total;
        index = index < lo ? lo : (hi < index ? hi : index);
        const uint32_t at = (p.wpos - (uint32_t)index) & p.mask; // temp[total - index]
        v0 = p.x0[at];
        if(p.two)
            v1 = p.x1[at];
    }
}

struct WaveOut { // how a rotated point becomes what is stored (reference :361-388)
    bool stereo, two, dup, normalize;
    float comp, db_min;
};
WF_DEV void wave_convert(const WaveOut &o, bool fresh, float r0, float r1, float &out0, float &out1, float &dup)
{
    dup = r0;  // one captured channel shown twice: row 1 is row 0 *before* its new points are converted, and its own count is 0,
               // so they stay raw (:361-362, :364-368)
    const float s1 = o.two ? r1 : r0;
    out0 = r0;
    out1 = s1; // mono display: the raw history of channel 1
    if(fresh) {
        if(o.stereo || !o.two)
            out0 = wave_dbfs(__builtin_fabsf(r0), o.db_min);
        else
            out0 = wave_dbfs(mul_unfused(add_unfused(__builtin_fabsf(r0), __builtin_fabsf(s1)), 0.5f), o.db_min);
        if(o.normalize)
            out0 = add_unfused(out0, o.comp);
        if(o.stereo && o.two) {
            out1 = wave_dbfs(__builtin_fabsf(s1), o.db_min);
            if(o.normalize)
                out1 = add_unfused(out1, o.comp);
        }
    }
}

__global__ __launch_bounds__(WAVE_THREADS) void waveform_tick_kernel(const WaveArgs a)
{
    constexpr int U = WAVE_U;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t stream = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVE_STREAMS + (threadIdx.x >> 6)));
    if(stream >= a.n_streams)
        return;
    const uint32_t W = a.width;
    const uint32_t sflags = a.stream_flags[stream];
    const bool was_silent = (sflags & WF_STREAM_LAST_SILENT) != 0;
    float *rows = a.rows + (size_t)stream * a.out_ch * W;
    const uint32_t disp = a.stereo ? 2u : 1u;
    if(sflags & WF_STREAM_PAUSED) // not ticked in this video frame: nothing of the stream moves
        return;
    const unsigned long long audio_ts = a.audio_ts_stream ? a.audio_ts_stream[stream] : a.audio_ts;

    if(sflags & WF_STREAM_HIDDEN) { // !m_show || capture timed out, :279-288
        if(was_silent)
            return;
        for(uint32_t i = lane; i < disp * W; i += 64u)
            rows[i] = a.db_min;
        if(lane == 0)
            a.stream_flags[stream] = sflags | WF_STREAM_LAST_SILENT;
        return;
    }
    const uint32_t wpos = a.wpos[stream];
    const uint32_t cend = a.cend[stream];
    const uint32_t R = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const uint32_t avail = wpos - cend;
    if(avail <= R) // not enough audio in advance, :293-295
        return;
    const uint32_t max_size = a.waveform_samples + R;
    const uint32_t total = avail < max_size ? avail : max_size; // :303-304
    const uint32_t sr = a.sample_rate;
    const unsigned long long start_ts = audio_ts - frames_to_ns(total, sr);
    const unsigned long long stop_ts = audio_ts - frames_to_ns(R, sr);
    if(start_ts >= audio_ts || stop_ts > audio_ts) {
        // timestamp rollover, :316-317 (a tick before any audio has a timestamp).  The rows are untouched, but the reference
        // has already trimmed the ring to max_size on its way here (:303-304), and the samples it dropped stay dropped
        if(lane == 0 && avail > max_size)
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
    WavePlan p;
    p.x0 = a.ring + (size_t)stream * a.cap_ch * a.ring_stride;
    p.x1 = p.x0 + a.ring_stride;
    p.rows = rows;
    p.W = W; p.keep = W - counts; p.counts = counts; p.wpos = wpos; p.R = R; p.total = total; p.sr = sr; p.mask = a.ring_mask;
    p.wts = wts; p.audio_ts = audio_ts; p.step_ns = a.step_ns;
    p.two = a.cap_ch > 1;
    WaveOut o;
    o.stereo = a.stereo != 0; o.two = p.two; o.dup = a.out_ch > a.cap_ch; o.normalize = a.normalize != 0;
    o.comp = a.normalize ? (a.vol_comp_stream ? a.vol_comp_stream[stream] : a.vol_comp) : 0.0f;
    o.db_min = a.db_min;

    const bool single = W <= 64u * U;
    float r0[U], r1[U];
    bool nz0 = false, nz1 = false;
    // the rotated rows (:332) and whether they hold a non-zero value (:334-343)
    for(uint32_t base = 0; base < W; base += 64u * U) {
#pragma unroll
        for(int k = 0; k < U; ++k) {
            const uint32_t i = base + lane + 64u * (uint32_t)k;
            wave_gather(p, i, i < W, r0[k], r1[k]);
        }
#pragma unroll
        for(int k = 0; k < U; ++k) {
            nz0 = nz0 || r0[k] != 0.0f;
            nz1 = nz1 || r1[k] != 0.0f;
        }
    }
    const bool all_silent = !__any(nz0) && !__any(nz1); // every channel's row is zeros -> m_last_silent (:345-349)
    if(lane == 0) {
        a.cend[stream] = wpos - R;                               // everything but the reserve has been popped, :321
        a.wts[stream] = wts + (unsigned long long)counts * a.step_ns; // :351
        a.stream_flags[stream] = (sflags & ~WF_STREAM_LAST_SILENT) | (all_silent ? WF_STREAM_LAST_SILENT : 0u);
    }
    for(uint32_t base = 0; base < W; base += 64u * U) {
        if(!single) { // the batch again (the scan has been through all of them)
#pragma unroll
            for(int k = 0; k < U; ++k) {
                const uint32_t i = base + lane + 64u * (uint32_t)k;
                wave_gather(p, i, i < W, r0[k], r1[k]);
            }
        }
#pragma unroll
        for(int k = 0; k < U; ++k) {
            const uint32_t i = base + lane + 64u * (uint32_t)k;
            if(i >= W)
                continue;
            if(all_silent) { // :353-359: the displayed rows read DB_MIN; a captured channel that is not displayed keeps its
                rows[i] = a.db_min; // (rotated) raw history
                if(disp > 1u)
                    rows[W + i] = a.db_min;
                else if(p.two)
                    rows[W + i] = r1[k];
            } else {
                float out0, out1, dup;
                wave_convert(o, i >= p.keep, r0[k], r1[k], out0, out1, dup);
                if(o.dup)
                    rows[W + i] = dup;
                rows[i] = out0;
                if(p.two)
                    rows[W + i] = out1;
            }
        }
    }
}

} // namespace wf
