/*
 * wf_oracle_meter.c -- CPU restatement of the reference's level-meter tick.  TEST INFRASTRUCTURE
 * (see wf_oracle.h: only tests/, tools/make_golden.py, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; libwaveform_hip.so never does).
 *
 * Restates, for one source in METER / STEPPED_METER display mode:
 *   setup    WAVSource::update()               src/source.cpp:1106-1128 (mode overrides, buffer length),
 *                                              :1181 (meter buffer = 0), :1243 (no zero pre-fill), :1257-1266
 *   ingest   WAVSource::capture_audio()        src/source.cpp:1873-1886
 *   tick     WAVSourceGeneric::tick_meter      src/source_generic.cpp:182-269
 *   bars     render_bars in meter mode         src/source.cpp:1476-1494, :1505-1509, :1548-1557
 * Build: gcc -O2 -std=c11 -ffp-contract=off, like wf_oracle.c.
 *
 * PARITY PIN: checked against the reference itself through golden vectors generated from oracle/_ref
 * (tests/golden/meter_*.npz, tools/make_golden.py) in tests/test_golden.py.
 */
#include "wf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct wfo_meter {
    wf_config cfg;
    uint32_t n;                 /* m_fft_size, repurposed: meter buffer length */
    uint32_t cap_ch;
    float *ring[2];             /* m_capturebufs (CircularBuffer), in samples */
    size_t ring_len[2], ring_cap[2];
    uint32_t sync_delay;        /* dtsize in frames */
    int state;                  /* 0 shown, 1 !m_show, 2 capture timed out */
    int last_silent;
    float *buffer[2];           /* m_decibels, repurposed: the circular meter buffer */
    size_t pos[2];              /* m_meter_pos */
    float val[2];               /* m_meter_val (dBFS) */
    float ema[2];               /* m_meter_buf */
    float bars[2];              /* m_interp_bufs[0] after render_bars */
    float border_top, border_bottom;
    int exact;                  /* wfo_meter_set_exact: the RMS sum in double (what the float sum approximates) */
};

static float db_min_f(void) { return 20.0f * log10f(FLT_MIN); }
static float dbfs_f(float mag) { return (mag > 0.0f) ? 20.0f * log10f(mag) : db_min_f(); } /* src/source.hpp:293-299 */
static float clampf(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
static float lerpf_std(float a, float b, float t) /* std::lerp (libstdc++) */
{
    if((a <= 0 && b >= 0) || (a >= 0 && b <= 0))
        return t * b + (1 - t) * a;
    if(t == 1)
        return b;
    const float x = a + t * (b - a);
    return ((t > 1) == (b > a)) ? (b < x ? x : b) : (b > x ? x : b);
}

/* get_gravity, src/source.hpp:301-312 */
static float gravity(const wfo_meter *m, float seconds)
{
    const float denom = 0.03868924705242879469662125316986f;
    const float hi = denom * 5.0f;
    const float lo = 0.0f;
    if((m->cfg.tsmoothing == WF_TSMOOTH_NONE) || (m->cfg.gravity <= 0.0f))
        return 0.0f;
    return (m->cfg.tsmoothing == WF_TSMOOTH_TVEXPONENTIAL) ? expf(-seconds / lerpf_std(lo, hi, m->cfg.gravity)) : m->cfg.gravity;
}

static void ring_push(wfo_meter *m, int ch, const float *src, size_t frames)
{
    if(m->ring_len[ch] + frames > m->ring_cap[ch]) {
        size_t cap = m->ring_cap[ch] ? m->ring_cap[ch] : 1024;
        while(cap < m->ring_len[ch] + frames)
            cap *= 2;
        m->ring[ch] = (float *)realloc(m->ring[ch], cap * sizeof(float));
        m->ring_cap[ch] = cap;
    }
    if(src != NULL)
        memcpy(m->ring[ch] + m->ring_len[ch], src, frames * sizeof(float));
    else
        memset(m->ring[ch] + m->ring_len[ch], 0, frames * sizeof(float));
    m->ring_len[ch] += frames;
}
/* CircularBuffer::pop_front(dst, n): dst may be NULL */
static void ring_pop(wfo_meter *m, int ch, float *dst, size_t frames)
{
    if(frames > m->ring_len[ch])
        frames = m->ring_len[ch];
    if(dst != NULL)
        memcpy(dst, m->ring[ch], frames * sizeof(float));
    memmove(m->ring[ch], m->ring[ch] + frames, (m->ring_len[ch] - frames) * sizeof(float));
    m->ring_len[ch] -= frames;
}

wfo_meter *wfo_meter_create(const wf_config *cfg)
{
    if(cfg == NULL || !cfg->meter || cfg->capture_channels < 1 || cfg->capture_channels > 2)
        return NULL;
    wfo_meter *m = (wfo_meter *)calloc(1, sizeof(*m));
    m->cfg = *cfg;
    /* get_settings()' repairs, src/source.cpp:567-579 */
    if((m->cfg.cutoff_high - m->cfg.cutoff_low) < 0) {
        m->cfg.cutoff_high = 17500;
        m->cfg.cutoff_low = 120;
    }
    if((m->cfg.ceiling_db - m->cfg.floor_db) < 1) {
        m->cfg.ceiling_db = 0;
        m->cfg.floor_db = -120;
    }
    if(!m->cfg.stereo || (((int)m->cfg.height - m->cfg.channel_spacing) < 1))
        m->cfg.channel_spacing = 0;
    /* src/source.cpp:1108-1121 */
    m->cfg.stereo = 0;
    m->cfg.slope = 0.0f;
    m->cfg.normalize_volume = 0;
    m->cfg.mirror_freq_axis = 0;
    m->n = (uint32_t)((size_t)((double)cfg->sample_rate * ((double)cfg->meter_ms / 1000.0)) & (size_t)-16);
    m->cap_ch = cfg->capture_channels;
    for(uint32_t c = 0; c < 2; ++c) {
        m->buffer[c] = (float *)calloc(m->n ? m->n : 1, sizeof(float)); /* :1181: meter mode fills 0.0f */
        m->ema[c] = db_min_f();                                          /* :1124-1127 */
        m->val[c] = db_min_f();
    }
    /* render_bars geometry, src/source.cpp:1476-1494 (m_stereo is false in meter mode) */
    const float bottom = (float)m->cfg.height;
    const float cpos = bottom;
    const float cap_radius = (float)m->cfg.bar_width / 2.0f;
    const float channel_offset = m->cfg.channel_spacing * 0.5f;
    float border_top = m->cfg.rounded_caps ? cap_radius : 0.0f;
    float border_bottom = m->cfg.rounded_caps ? cpos - cap_radius : cpos;
    if(m->cfg.channel_spacing > 0)
        border_bottom -= channel_offset;
    if(m->cfg.min_bar_height > 0)
        border_bottom -= m->cfg.min_bar_height;
    m->border_top = border_top;
    m->border_bottom = clampf(border_bottom, border_top, cpos);
    m->bars[0] = m->bars[1] = m->border_bottom;
    return m;
}

void wfo_meter_destroy(wfo_meter *m)
{
    if(m == NULL)
        return;
    for(int c = 0; c < 2; ++c) {
        free(m->ring[c]);
        free(m->buffer[c]);
    }
    free(m);
}

void wfo_meter_set_sync_delay(wfo_meter *m, uint32_t frames) { m->sync_delay = frames; }
void wfo_meter_set_state(wfo_meter *m, int state) { m->state = state; }
void wfo_meter_set_exact(wfo_meter *m, int exact) { m->exact = exact; }

/* capture_audio, src/source.cpp:1873-1886 */
void wfo_meter_push_audio(wfo_meter *m, const float *ch0, const float *ch1, uint32_t frames, int muted)
{
    const float *data[2] = {ch0, ch1};
    for(uint32_t j = 0; j < m->cap_ch; ++j) {
        ring_push(m, (int)j, (muted || data[j] == NULL) ? NULL : data[j], frames);
        const size_t max_size = (size_t)m->sync_delay + m->n;
        if(m->ring_len[j] > max_size)
            ring_pop(m, (int)j, NULL, m->ring_len[j] - max_size);
    }
}

/* WAVSourceGeneric::tick_meter, src/source_generic.cpp:182-269 */
void wfo_meter_tick(wfo_meter *m, float seconds)
{
    const float DB_MIN = db_min_f();
    if(m->state == 2) { /* dtcapture > CAPTURE_TIMEOUT, :184-199 */
        if(m->last_silent)
            return;
        for(uint32_t ch = 0; ch < m->cap_ch; ++ch)
            for(size_t i = 0; i < m->n; ++i)
                m->buffer[ch][i] = 0.0f;
        for(int i = 0; i < 2; ++i) {
            m->ema[i] = 0.0f;
            m->val[i] = DB_MIN;
        }
        m->last_silent = 1;
        return;
    }
    const size_t outsz = m->n;
    const size_t dtsize = m->sync_delay; /* in samples */
    for(uint32_t ch = 0; ch < m->cap_ch; ++ch) { /* :204-220 */
        while(m->ring_len[ch] > dtsize) {
            const size_t consume = m->ring_len[ch] - dtsize;
            const size_t max = m->n - m->pos[ch];
            if(consume >= max) {
                ring_pop(m, (int)ch, &m->buffer[ch][m->pos[ch]], max);
                m->pos[ch] = 0;
            } else {
                ring_pop(m, (int)ch, &m->buffer[ch][m->pos[ch]], consume);
                m->pos[ch] += consume;
            }
        }
    }
    if(m->state == 1) { /* !m_show, :222-230 */
        for(int i = 0; i < 2; ++i) {
            m->ema[i] = 0.0f;
            m->val[i] = DB_MIN;
        }
        m->last_silent = 1;
        return;
    }
    for(uint32_t ch = 0; ch < m->cap_ch; ++ch) { /* :232-260 */
        float out = 0.0f;
        if(m->cfg.meter_rms && m->exact) {
            /* the value the reference's sequential float sum approximates: every term and the sum in double, rounded once */
            double acc = 0.0;
            for(size_t i = 0; i < outsz; ++i)
                acc += (double)m->buffer[ch][i] * (double)m->buffer[ch][i];
            out = (float)sqrt(acc / (double)m->n);
        } else if(m->cfg.meter_rms) {
            for(size_t i = 0; i < outsz; ++i) {
                const float v = m->buffer[ch][i];
                out += v * v;
            }
            out = sqrtf(out / m->n);
        } else {
            for(size_t i = 0; i < outsz; ++i)
                out = fmaxf(out, fabsf(m->buffer[ch][i]));
        }
        if(m->cfg.tsmoothing != WF_TSMOOTH_NONE) {
            const float g = gravity(m, seconds);
            const float g2 = 1.0f - g;
            if(!m->cfg.fast_peaks || (out <= m->ema[ch]))
                out = (g * m->ema[ch]) + (g2 * out);
        }
        m->ema[ch] = out;
        m->val[ch] = dbfs_f(out);
    }
    unsigned silent_channels = 0;
    for(uint32_t ch = 0; ch < m->cap_ch; ++ch)
        if(m->val[ch] < (float)(m->cfg.floor_db - 10))
            ++silent_channels;
    m->last_silent = (silent_channels >= m->cap_ch);
}

/* render_bars in meter mode: m_interp_bufs[0][i] = m_meter_val[i], then the dB -> pixel mapping
 * (src/source.cpp:1505-1509, :1548-1557) */
void wfo_meter_render(wfo_meter *m)
{
    const int dbrange = m->cfg.ceiling_db - m->cfg.floor_db;
    for(uint32_t i = 0; i < m->cap_ch; ++i)
        m->bars[i] = lerpf_std(m->border_top, m->border_bottom, clampf(m->cfg.ceiling_db - m->val[i], 0.0f, (float)dbrange) / dbrange);
}

uint32_t wfo_meter_size(const wfo_meter *m) { return m->n; }
int wfo_meter_last_silent(const wfo_meter *m) { return m->last_silent; }
float wfo_meter_val(const wfo_meter *m, int ch) { return m->val[ch & 1]; }
float wfo_meter_ema(const wfo_meter *m, int ch) { return m->ema[ch & 1]; }
float wfo_meter_bar(const wfo_meter *m, int ch) { return m->bars[ch & 1]; }
