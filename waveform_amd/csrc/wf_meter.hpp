// wf_meter.hpp -- gfx950 kernel of the level-meter tick (device code only; hipcc).
//
//   meter_tick_kernel    WAVSource*::tick_meter for a whole batch of sources (reference src/source_generic.cpp:182-269,
//                        AVX variant src/source_avx.cpp:202-322): consume the captured audio that lies before the tick
//                        time into the meter buffer, RMS or peak over the buffer, temporal smoothing, dBFS, silence flag;
//                        plus what render_bars makes of the levels (src/source.cpp:1505-1509, :1548-1557).
//
// The reference keeps a circular "meter buffer" of m_fft_size samples per channel (m_decibels, repurposed) into which
// tick_meter pops everything older than the A/V-sync point; the level is taken over the whole buffer.  Here the device
// ring *is* that buffer: the meter buffer's contents are exactly the m_fft_size samples that end at the consumption
// point `mend` (monotonic: audio once consumed stays consumed even if a later tick asks for a larger sync delay), so
// the kernel is a pure streaming reduction over ring[mend - size, mend) -- HBM-bound, 4 * size bytes per channel and
// tick (the EMA state is two floats).  One workgroup per stream; its captured channels are reduced together because
// m_last_silent couples them (:262-268).
#pragma once
#include <hip/hip_runtime.h>
#include "wf_tick_phases.hpp"

namespace wf {

struct MeterArgs {
    float *ring;               // [n_streams * cap_ch][ring_cap]  (written only by the capture-timeout branch)
    const uint32_t *wpos;      // [n_streams] samples written so far, modulo 2^32
    uint32_t *mend;            // [n_streams] consumption point: samples popped into the meter buffer so far, modulo 2^32
    uint32_t ring_mask;
    uint32_t ring_cap;
    uint32_t ring_stride;      // floats between the rings of consecutive channels (>= ring_cap)
    uint32_t delay;            // frames of captured audio that lie after the tick time (dtsize, :201-202)
    const uint32_t *delay_stream;
    uint32_t size;             // m_fft_size = meter buffer length (multiple of 16)
    float *meter_buf;          // [n_streams * cap_ch] m_meter_buf (EMA state)
    float *meter_val;          // [n_streams * cap_ch] m_meter_val (dBFS)
    uint32_t *stream_flags;    // [n_streams]
    float *bars;               // [n_streams][cap_ch] bar tops in pixels, or nullptr
    float g, g2;               // get_gravity(seconds), 1 - g
    float db_min;
    float silent_floor;        // (float)(m_floor - 10)
    float border_top, border_bottom, ceiling, dbrange;
    uint32_t n_streams;
    uint32_t cap_ch;
    uint32_t rms;              // m_meter_rms
    uint32_t tsmooth;          // m_tsmoothing != NONE
    uint32_t fast_peaks;
};

constexpr int METER_THREADS = 256;
constexpr int METER_UNROLL = 8; // 16-byte loads in flight per thread and channel

// sum of squares / max |x| of the four samples of one aligned chunk, masked to the window [head, head + size)
WF_DEV void meter_accumulate(const f4 v, uint32_t e0, uint32_t head, uint32_t stop, bool rms, float &acc)
{
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for(uint32_t e = 0; e < 4; ++e) {
        const bool in = (e0 + e) >= head && (e0 + e) < stop;
        const float s = in ? x[e] : 0.0f;
        acc = rms ? __builtin_fmaf(s, s, acc) : __builtin_fmaxf(acc, __builtin_fabsf(s));
    }
}

__global__ __launch_bounds__(METER_THREADS) void meter_tick_kernel(const MeterArgs a)
{
    __shared__ float part[2][METER_THREADS / 64];
    const uint32_t stream = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const uint32_t wpos = a.wpos[stream];
    const uint32_t sflags = a.stream_flags[stream];
    uint32_t mend = a.mend[stream];
    const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const bool was_silent = (sflags & WF_STREAM_LAST_SILENT) != 0;
    float *rows = a.ring + (size_t)stream * a.cap_ch * a.ring_stride;
    float *buf = a.meter_buf + (size_t)stream * a.cap_ch;
    float *val = a.meter_val + (size_t)stream * a.cap_ch;
    float *bar = a.bars ? a.bars + (size_t)stream * a.cap_ch : nullptr;

    // a source that was not ticked in this video frame (the plugin's batched mode: OBS ticks only active sources): nothing is
    // consumed, nothing changes
    if(sflags & WF_STREAM_PAUSED)
        return;
    // the window in aligned 16-byte chunks: chunk j covers ring positions base + 4j .. +3 (never straddles the wrap)
    const uint32_t start = mend - a.size;

    if(sflags & WF_STREAM_TIMEOUT) {
        // capture lost (:184-199): once, the meter buffer and the state are cleared; nothing is consumed
        if(was_silent)
            return;
        for(uint32_t c = 0; c < a.cap_ch; ++c)
            for(uint32_t i = (uint32_t)tid; i < a.size; i += METER_THREADS)
                rows[(size_t)c * a.ring_stride + ((start + i) & a.ring_mask)] = 0.0f;
        if(tid < (int)a.cap_ch) {
            buf[tid] = 0.0f;
            val[tid] = a.db_min;
            if(bar)
                bar[tid] = a.border_bottom;
        }
        if(tid == 0)
            a.stream_flags[stream] = sflags | WF_STREAM_LAST_SILENT;
        return;
    }

    // consume everything older than the sync point (:204-220); the consumption point never moves back
    const uint32_t cand = wpos - delay;
    if((int32_t)(cand - mend) > 0)
        mend = cand;
    if(tid == 0)
        a.mend[stream] = mend;

    if(sflags & WF_STREAM_HIDDEN) {
        // !m_show (:222-230): the audio has been consumed, the state is reset
        if(tid < (int)a.cap_ch) {
            buf[tid] = 0.0f;
            val[tid] = a.db_min;
            if(bar)
                bar[tid] = a.border_bottom;
        }
        if(tid == 0)
            a.stream_flags[stream] = sflags | WF_STREAM_LAST_SILENT;
        return;
    }

    const uint32_t s0 = mend - a.size;
    const uint32_t base = s0 & ~3u;
    const uint32_t head = s0 - base;          // 0..3 samples of the first chunk lie before the window
    const uint32_t stop = head + a.size;
    const uint32_t n_chunks = (stop + 3u) >> 2;
    const bool rms = a.rms != 0;
    const bool two = a.cap_ch > 1;
    const float *row0 = rows, *row1 = rows + a.ring_stride;

    float acc0 = 0.0f, acc1 = 0.0f;
    for(uint32_t j0 = (uint32_t)tid; j0 < n_chunks; j0 += METER_THREADS * METER_UNROLL) {
        f4 v0[METER_UNROLL], v1[METER_UNROLL];
#pragma unroll
        for(int u = 0; u < METER_UNROLL; ++u) {
            const uint32_t j = j0 + (uint32_t)u * METER_THREADS;
            const uint32_t jj = j < n_chunks ? j : n_chunks - 1; // clamp: the load is always legal, the mask drops it
            const uint32_t idx = (base + 4u * jj) & a.ring_mask;
            v0[u] = ld4(row0 + idx);
            if(two)
                v1[u] = ld4(row1 + idx);
        }
#pragma unroll
        for(int u = 0; u < METER_UNROLL; ++u) {
            const uint32_t j = j0 + (uint32_t)u * METER_THREADS;
            const uint32_t e0 = j < n_chunks ? 4u * j : stop; // out-of-range chunk: every element masked
            meter_accumulate(v0[u], e0, head, stop, rms, acc0);
            if(two)
                meter_accumulate(v1[u], e0, head, stop, rms, acc1);
        }
    }
    // workgroup reduction: butterflies inside the wavefront, then the four wavefronts through LDS
#pragma unroll
    for(int m = 32; m >= 1; m >>= 1) {
        const float o0 = __shfl_xor(acc0, m, 64), o1 = __shfl_xor(acc1, m, 64);
        acc0 = rms ? acc0 + o0 : __builtin_fmaxf(acc0, o0);
        acc1 = rms ? acc1 + o1 : __builtin_fmaxf(acc1, o1);
    }
    if((tid & 63) == 0) {
        part[0][tid >> 6] = acc0;
        part[1][tid >> 6] = acc1;
    }
    __syncthreads();
    if(tid != 0)
        return;

    uint32_t silent_channels = 0;
    for(uint32_t c = 0; c < a.cap_ch; ++c) {
        float out = part[c][0];
        for(int w = 1; w < METER_THREADS / 64; ++w)
            out = rms ? out + part[c][w] : __builtin_fmaxf(out, part[c][w]);
        if(rms)
            out = __fsqrt_rn(__fdiv_rn(out, (float)a.size)); // std::sqrt(out / m_fft_size), :243
        if(a.tsmooth) {
            const float old = buf[c];
            if(!a.fast_peaks || out <= old)
                out = meter_ema(a.g, old, a.g2, out); // (g * m_meter_buf) + (g2 * out), :255
        }
        buf[c] = out;
        const float db = (out > 0.0f) ? mul_unfused(20.0f, log10f(out)) : a.db_min; // dbfs(), src/source.hpp:293-299
        val[c] = db;
        if(db < a.silent_floor)
            ++silent_channels;
        if(bar) {
            // render_bars: m_interp_bufs[0][c] = m_meter_val[c], then the dB -> pixel mapping (src/source.cpp:1505-1509, :1548-1557)
            float tt = a.ceiling - db;
            tt = (tt < 0.0f) ? 0.0f : (a.dbrange < tt) ? a.dbrange : tt;
            bar[c] = lerp_std(a.border_top, a.border_bottom, tt / a.dbrange);
        }
    }
    const bool last_silent = silent_channels >= a.cap_ch; // :262-268
    a.stream_flags[stream] = (sflags & ~WF_STREAM_LAST_SILENT) | (last_silent ? WF_STREAM_LAST_SILENT : 0u);
}

} // namespace wf
