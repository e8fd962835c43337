/*
 * wf_oracle_wave.c -- CPU restatement of the reference's waveform display tick.  TEST INFRASTRUCTURE
 * (see wf_oracle.h: only tests/, tools/make_golden.py, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; libwaveform_hip.so never does).
 *
 * Restates, for one source in WAVEFORM display mode:
 *   setup    WAVSource::update()                 src/source.cpp:1130-1142 (m_fft_size = m_width, m_waveform_samples,
 *                                                m_waveform_ts = 0), :1172-1182 (rows = DB_MIN), :1243-1248 (width zeros)
 *   ingest   WAVSource::capture_audio()          src/source.cpp:1833-1836, :1873-1886 (trim to dtsamples + m_waveform_samples)
 *   tick     WAVSourceGeneric::tick_waveform     src/source_generic.cpp:271-390
 * Time is the caller's: the end-of-audio timestamp m_audio_ts and the A/V-sync reserve (frames) are given per tick, as
 * get_audio_sync(m_tick_ts) would derive them.  Build: gcc -O2 -std=c11 -ffp-contract=off, like wf_oracle.c.
 *
 * PARITY PIN: golden vectors generated from oracle/_ref (tests/golden/wave_*.npz) in tests/test_golden.py.
 */
#include "wf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct wfo_wave {
    wf_config cfg;
    uint32_t n;                 /* m_fft_size = m_width: points per row */
    uint32_t cap_ch, out_ch;
    size_t waveform_samples;    /* m_waveform_samples */
    float *ring[2];
    size_t ring_len[2], ring_cap[2];
    uint32_t reserve;           /* A/V-sync reserve in frames (dtaudio > 0) */
    uint64_t audio_ts;          /* m_audio_ts */
    uint64_t waveform_ts;       /* m_waveform_ts */
    int hidden;                 /* !m_show || capture timed out */
    int last_silent;
    float input_rms;
    /* volume-normalisation producer, as in wf_oracle.c (capture_audio's RMS part, sync_rms_buffer, update_input_rms) */
    float *rms_sync; size_t rms_sync_len, rms_sync_cap;
    float *rms_buf; size_t rms_size, rms_pos;
    float *rows[2];             /* m_decibels */
    float *temp; size_t temp_cap; /* m_interp_bufs[2] */
};

static float db_min_f(void) { return 20.0f * log10f(FLT_MIN); }
static float dbfs_f(float mag) { return (mag > 0.0f) ? 20.0f * log10f(mag) : db_min_f(); }
/* libobs util_mul_div64 and the two helpers built on it (media-io/audio-io.h) */
static uint64_t mul_div64(uint64_t num, uint64_t mul, uint64_t div)
{
    const uint64_t rem = num % div;
    return (num / div) * mul + (rem * mul) / div;
}
static uint64_t frames_to_ns(uint64_t sr, uint64_t frames) { return mul_div64(frames, 1000000000ULL, sr); }
static uint64_t ns_to_frames(uint64_t sr, uint64_t ns) { return mul_div64(ns, sr, 1000000000ULL); }

static void ring_push(wfo_wave *w, int ch, const float *src, size_t frames)
{
    if(w->ring_len[ch] + frames > w->ring_cap[ch]) {
        size_t cap = w->ring_cap[ch] ? w->ring_cap[ch] : 1024;
        while(cap < w->ring_len[ch] + frames)
            cap *= 2;
        w->ring[ch] = (float *)realloc(w->ring[ch], cap * sizeof(float));
        w->ring_cap[ch] = cap;
    }
    if(src != NULL)
        memcpy(w->ring[ch] + w->ring_len[ch], src, frames * sizeof(float));
    else
        memset(w->ring[ch] + w->ring_len[ch], 0, frames * sizeof(float));
    w->ring_len[ch] += frames;
}
static void ring_pop(wfo_wave *w, int ch, float *dst, size_t frames)
{
    if(frames > w->ring_len[ch])
        frames = w->ring_len[ch];
    if(dst != NULL)
        memcpy(dst, w->ring[ch], frames * sizeof(float));
    memmove(w->ring[ch], w->ring[ch] + frames, (w->ring_len[ch] - frames) * sizeof(float));
    w->ring_len[ch] -= frames;
}

wfo_wave *wfo_wave_create(const wf_config *cfg)
{
    if(cfg == NULL || !cfg->waveform || cfg->width == 0 || cfg->capture_channels < 1 || cfg->capture_channels > 2)
        return NULL;
    wfo_wave *w = (wfo_wave *)calloc(1, sizeof(*w));
    w->cfg = *cfg;
    /* get_settings()' repairs, src/source.cpp:567-579 */
    if((w->cfg.cutoff_high - w->cfg.cutoff_low) < 0) {
        w->cfg.cutoff_high = 17500;
        w->cfg.cutoff_low = 120;
    }
    if((w->cfg.ceiling_db - w->cfg.floor_db) < 1) {
        w->cfg.ceiling_db = 0;
        w->cfg.floor_db = -120;
    }
    if(!w->cfg.stereo || (((int)w->cfg.height - w->cfg.channel_spacing) < 1))
        w->cfg.channel_spacing = 0;
    w->n = cfg->width;                                                            /* :1140 */
    w->waveform_samples = (size_t)((double)cfg->sample_rate * ((double)cfg->meter_ms / 1000.0)); /* :1141 */
    w->cap_ch = cfg->capture_channels;
    w->out_ch = ((w->cap_ch > 1) || cfg->stereo) ? 2u : 1u;
    for(uint32_t c = 0; c < w->out_ch; ++c) {
        w->rows[c] = (float *)malloc(sizeof(float) * w->n);
        for(uint32_t i = 0; i < w->n; ++i)
            w->rows[c][i] = db_min_f();
    }
    for(uint32_t c = 0; c < w->cap_ch; ++c) /* :1243-1248 */
        ring_push(w, (int)c, NULL, w->n);
    if(cfg->normalize_volume) { /* src/source.cpp:1144-1152 */
        w->rms_size = (size_t)cfg->sample_rate & (size_t)-16;
        w->rms_buf = (float *)calloc(w->rms_size, sizeof(float));
    }
    return w;
}

void wfo_wave_destroy(wfo_wave *w)
{
    if(w == NULL)
        return;
    for(int c = 0; c < 2; ++c) {
        free(w->ring[c]);
        free(w->rows[c]);
    }
    free(w->temp);
    free(w->rms_sync);
    free(w->rms_buf);
    free(w);
}

void wfo_wave_set_time(wfo_wave *w, uint64_t audio_ts_ns, uint32_t reserve_frames)
{
    w->audio_ts = audio_ts_ns;
    w->reserve = reserve_frames;
}
void wfo_wave_set_hidden(wfo_wave *w, int hidden) { w->hidden = hidden; }
void wfo_wave_set_input_rms(wfo_wave *w, float rms) { w->input_rms = rms; }

/* capture_audio: the ring keeps dtsamples + m_waveform_samples samples (the caller's reserve stands in for dtsamples) */
void wfo_wave_push_audio(wfo_wave *w, const float *ch0, const float *ch1, uint32_t frames, int muted)
{
    const float *data[2] = {ch0, ch1};
    if(w->cfg.normalize_volume) { /* src/source.cpp:1842-1871 */
        if(w->rms_sync_len + frames > w->rms_sync_cap) {
            size_t cap = w->rms_sync_cap ? w->rms_sync_cap : 4096;
            while(cap < w->rms_sync_len + frames)
                cap *= 2;
            w->rms_sync = (float *)realloc(w->rms_sync, cap * sizeof(float));
            w->rms_sync_cap = cap;
        }
        for(uint32_t i = 0; i < frames; ++i) {
            float val = 0.0f;
            for(uint32_t ch = 0; ch < w->cap_ch; ++ch)
                if(data[ch] != NULL)
                    val = fmaxf(fabsf(data[ch][i]), val);
            w->rms_sync[w->rms_sync_len + i] = val * val;
        }
        w->rms_sync_len += frames;
        const size_t max_rms_size = (size_t)w->reserve + w->rms_size;
        if(w->rms_sync_len > max_rms_size) {
            const size_t drop = w->rms_sync_len - max_rms_size;
            memmove(w->rms_sync, w->rms_sync + drop, max_rms_size * sizeof(float));
            w->rms_sync_len = max_rms_size;
        }
    }
    for(uint32_t j = 0; j < w->cap_ch; ++j) {
        ring_push(w, (int)j, (muted || data[j] == NULL) ? NULL : data[j], frames);
        const size_t max_size = (size_t)w->reserve + w->waveform_samples;
        if(w->ring_len[j] > max_size)
            ring_pop(w, (int)j, NULL, w->ring_len[j] - max_size);
    }
}

/* update_input_rms + sync_rms_buffer (src/source_generic.cpp:392-403, src/source.cpp:810-835): what WAVSource::tick runs first */
float wfo_wave_update_input_rms(wfo_wave *w)
{
    if(!w->cfg.normalize_volume)
        return w->input_rms;
    const size_t dtsize = w->reserve;
    if(w->rms_sync_len <= dtsize)
        return w->input_rms;
    size_t head = 0;
    while(w->rms_sync_len - head > dtsize) {
        const size_t consume = w->rms_sync_len - head - dtsize;
        const size_t max = w->rms_size - w->rms_pos;
        const size_t n = (consume >= max) ? max : consume;
        memcpy(w->rms_buf + w->rms_pos, w->rms_sync + head, n * sizeof(float));
        head += n;
        w->rms_pos = (consume >= max) ? 0 : w->rms_pos + n;
    }
    memmove(w->rms_sync, w->rms_sync + head, (w->rms_sync_len - head) * sizeof(float));
    w->rms_sync_len -= head;
    float sum = 0.0f;
    for(size_t i = 0; i < w->rms_size; ++i)
        sum += w->rms_buf[i];
    w->input_rms = sqrtf(sum / w->rms_size);
    return w->input_rms;
}

/* WAVSourceGeneric::tick_waveform, src/source_generic.cpp:271-390 */
void wfo_wave_tick(wfo_wave *w)
{
    const size_t outsz = w->n;
    const float DB_MIN = db_min_f();
    const uint64_t sr = w->cfg.sample_rate;
    if(w->hidden) { /* :279-288 */
        if(w->last_silent)
            return;
        for(int ch = 0; ch < (w->cfg.stereo ? 2 : 1); ++ch)
            for(size_t i = 0; i < outsz; ++i)
                w->rows[ch][i] = DB_MIN;
        w->last_silent = 1;
        return;
    }
    const size_t reserve = w->reserve;                  /* in samples */
    const size_t max_size = w->waveform_samples + reserve;
    for(uint32_t i = 0; i < w->cap_ch; ++i)             /* :293-295 */
        if(w->ring_len[i] <= reserve)
            return;

    size_t counts[2] = {0, 0};
    unsigned silent_channels = 0;
    const uint64_t step_ns = ((uint64_t)w->cfg.meter_ms * 1000000u) / (uint64_t)outsz; /* :299 */
    for(uint32_t ch = 0; ch < w->cap_ch; ++ch) {
        if(w->ring_len[ch] > max_size)
            ring_pop(w, (int)ch, NULL, w->ring_len[ch] - max_size);
        if(w->temp_cap < w->ring_len[ch]) {
            w->temp = (float *)realloc(w->temp, w->ring_len[ch] * sizeof(float));
            w->temp_cap = w->ring_len[ch];
        }
        const size_t consume = w->ring_len[ch] - reserve;
        const size_t total_samples = w->ring_len[ch];
        const size_t reserve_samples = reserve;
        if(total_samples <= reserve_samples)
            return;
        const uint64_t start_ts = w->audio_ts - frames_to_ns(sr, total_samples);
        const uint64_t stop_ts = w->audio_ts - frames_to_ns(sr, reserve_samples);
        if((start_ts >= w->audio_ts) || (stop_ts > w->audio_ts))
            return; /* timestamp rollover */
        if(w->waveform_ts < start_ts)
            w->waveform_ts = start_ts;
        if((w->waveform_ts > stop_ts) && ((w->waveform_ts - stop_ts) > step_ns))
            w->waveform_ts = start_ts;
        ring_pop(w, (int)ch, w->temp, consume);
        for(size_t i = 0; i < outsz; ++i) {
            const uint64_t ts = w->waveform_ts + (i * step_ns);
            if(ts >= stop_ts)
                break;
            if(ts < w->waveform_ts)
                break;
            uint64_t index = ns_to_frames(sr, w->audio_ts - ts);
            const uint64_t lo = (uint64_t)reserve_samples + 1u, hi = (uint64_t)total_samples;
            index = (index < lo) ? lo : (hi < index) ? hi : index; /* std::clamp */
            w->rows[ch][counts[ch]++] = w->temp[total_samples - index];
        }
        /* std::rotate(first, first + counts, last): the new points move to the end */
        if(counts[ch] > 0 && counts[ch] < outsz) {
            float *tmp = (float *)malloc(counts[ch] * sizeof(float));
            memcpy(tmp, w->rows[ch], counts[ch] * sizeof(float));
            memmove(w->rows[ch], w->rows[ch] + counts[ch], (outsz - counts[ch]) * sizeof(float));
            memcpy(w->rows[ch] + (outsz - counts[ch]), tmp, counts[ch] * sizeof(float));
            free(tmp);
        }
        int silent = 1;
        for(size_t i = 0; i < outsz; ++i)
            if(w->rows[ch][i] != 0.0f) {
                silent = 0;
                w->last_silent = 0;
                break;
            }
        if(silent) {
            if(++silent_channels >= w->cap_ch)
                w->last_silent = 1;
        }
    }
    w->waveform_ts += (counts[0] * step_ns);

    if(w->last_silent) { /* :353-359 */
        for(int ch = 0; ch < (w->cfg.stereo ? 2 : 1); ++ch)
            for(size_t i = 0; i < outsz; ++i)
                w->rows[ch][i] = DB_MIN;
        return;
    }
    if(w->out_ch > w->cap_ch)
        memcpy(w->rows[1], w->rows[0], outsz * sizeof(float));
    if(w->cfg.stereo) {
        for(int ch = 0; ch < 2; ++ch)
            for(size_t i = (outsz - counts[ch]); i < outsz; ++i)
                w->rows[ch][i] = dbfs_f(fabsf(w->rows[ch][i]));
    } else if(w->cap_ch > 1) {
        for(size_t i = (outsz - counts[0]); i < outsz; ++i)
            w->rows[0][i] = dbfs_f((fabsf(w->rows[0][i]) + fabsf(w->rows[1][i])) * 0.5f);
    } else {
        for(size_t i = (outsz - counts[0]); i < outsz; ++i)
            w->rows[0][i] = dbfs_f(fabsf(w->rows[0][i]));
    }
    if(w->cfg.normalize_volume) {
        const float a = w->cfg.volume_target - dbfs_f(w->input_rms);
        const float comp = (w->cfg.max_gain < a) ? w->cfg.max_gain : a;
        for(int ch = 0; ch < (w->cfg.stereo ? 2 : 1); ++ch)
            for(size_t i = (outsz - counts[ch]); i < outsz; ++i)
                w->rows[ch][i] += comp;
    }
}

uint32_t wfo_wave_points(const wfo_wave *w) { return w->n; }
uint32_t wfo_wave_output_channels(const wfo_wave *w) { return w->out_ch; }
int wfo_wave_last_silent(const wfo_wave *w) { return w->last_silent; }
const float *wfo_wave_row(const wfo_wave *w, int ch) { return w->rows[ch & 1]; }
uint64_t wfo_wave_ts(const wfo_wave *w) { return w->waveform_ts; }
