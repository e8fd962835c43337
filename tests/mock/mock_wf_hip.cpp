// tests/mock/mock_wf_hip.cpp -- TEST HARNESS: a host-only stand-in for libwaveform_hip.so with the entry points the reference-side
// binding (host/wav_source_hip.cpp) resolves, so that the binding -- its process-wide group registry, the registry mutex against
// every source's m_mtx, members joining and leaving batches, the double-buffered staging -- runs on a GPU-less box under
// ThreadSanitizer (tests/test_sanitizers.py; the HIP runtime itself does not start under that tool).  A stream's rows are a
// deterministic function of the audio it has received since its last reset, so the stress test's "a source that survived the
// chaos equals a fresh one" check means something here too.  Nothing here is product code.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "wf_hip.h"

struct wf_hip {
    wf_config cfg{};
    uint32_t n = 0, N = 0, M = 0, cap = 1, out = 1, disp = 1, bars = 0;
    std::vector<uint64_t> sum;      // per stream: a checksum of everything pushed since the reset
    std::vector<uint8_t> state, silent;
    std::vector<float> rows;        // [n][out][M]
    std::string err;
};
static thread_local std::string g_err;

static uint64_t mix(uint64_t h, const float *p, size_t n)
{
    for(size_t i = 0; i < n; ++i) {
        uint32_t u;
        std::memcpy(&u, p + i, 4);
        h = (h ^ u) * 1099511628211ull;
    }
    return h;
}

extern "C" {
int wf_hip_abi_version(void) { return WF_HIP_ABI_VERSION; }
int wf_hip_device_count(void) { return 1; }
const char *wf_hip_last_error(const wf_hip *h) { return h ? h->err.c_str() : g_err.c_str(); }
int wf_hip_create(const wf_config *cfg, int, uint32_t max_streams, uint32_t, wf_hip **out)
{
    auto *h = new wf_hip;
    h->cfg = *cfg;
    h->n = max_streams;
    h->N = cfg->fft_size;
    h->M = cfg->waveform ? cfg->fft_size : cfg->fft_size / 2;
    h->cap = cfg->capture_channels;
    h->out = (cfg->capture_channels > 1 || cfg->stereo) ? 2u : 1u;
    h->disp = cfg->stereo ? 2u : 1u;
    h->bars = cfg->bars ? 26u : (cfg->curve ? cfg->width : 0u);
    h->sum.assign(h->n, 1469598103934665603ull);
    h->state.assign(h->n, 0);
    h->silent.assign(h->n, 0);
    h->rows.assign((size_t)h->n * h->out * h->M, -758.0f);
    *out = h;
    return WF_HIP_OK;
}
void wf_hip_destroy(wf_hip *h) { delete h; }
int wf_hip_reset(wf_hip *h, uint32_t first, uint32_t count)
{
    for(uint32_t s = first; s < first + count; ++s) {
        h->sum[s] = 1469598103934665603ull;
        h->state[s] = 0;
        h->silent[s] = 0;
        for(size_t k = 0; k < (size_t)h->out * h->M; ++k)
            h->rows[(size_t)s * h->out * h->M + k] = -758.0f;
    }
    return WF_HIP_OK;
}
uint32_t wf_hip_output_channels(const wf_hip *h) { return h->out; }
uint32_t wf_hip_display_channels(const wf_hip *h) { return h->disp; }
uint32_t wf_hip_num_bars(const wf_hip *h) { return h->bars; }
uint32_t wf_hip_num_vertices(const wf_hip *h) { return (h->cfg.vertices && h->cfg.bars) ? h->bars * 6u : 0u; }
void *wf_hip_host_alloc(size_t bytes) { return std::calloc(bytes ? bytes : 1, 1); }
void wf_hip_host_free(void *p) { std::free(p); }
int wf_hip_push_audio(wf_hip *h, uint32_t first, uint32_t count, const float *samples, uint32_t frames)
{
    for(uint32_t s = 0; s < count; ++s)
        h->sum[first + s] = mix(h->sum[first + s], samples + (size_t)s * h->cap * frames, (size_t)h->cap * frames);
    return WF_HIP_OK;
}
int wf_hip_push_audio_ragged_async(wf_hip *h, uint32_t first, uint32_t count, const float *p, const uint32_t *frames, uint32_t max_frames, uint32_t)
{
    for(uint32_t s = 0; s < count; ++s)
        for(uint32_t c = 0; c < h->cap; ++c)
            h->sum[first + s] = mix(h->sum[first + s], p + ((size_t)s * h->cap + c) * max_frames, frames[s]);
    return WF_HIP_OK;
}
int wf_hip_ingest_done(wf_hip *, uint32_t) { return WF_HIP_OK; }
int wf_hip_readback_done(wf_hip *, uint32_t) { return WF_HIP_OK; }
int wf_hip_set_hidden(wf_hip *h, uint32_t first, uint32_t count, const uint8_t *mask)
{
    for(uint32_t i = 0; i < count; ++i)
        h->state[first + i] = mask[i];
    return WF_HIP_OK;
}
int wf_hip_set_input_rms(wf_hip *, uint32_t, uint32_t, const float *) { return WF_HIP_OK; }
int wf_hip_enable_input_rms(wf_hip *h, int) { h->err = "mock: no device RMS producer"; return WF_HIP_ERR_UNSUPPORTED; }
int wf_hip_push_rms_ragged_async(wf_hip *, uint32_t, uint32_t, const float *, const uint32_t *, uint32_t, uint32_t) { return WF_HIP_OK; }
int wf_hip_set_stream_delay(wf_hip *, uint32_t, uint32_t, const uint32_t *) { return WF_HIP_OK; }
int wf_hip_set_stream_audio_ts(wf_hip *, uint32_t, uint32_t, const uint64_t *) { return WF_HIP_OK; }
int wf_hip_tick(wf_hip *h, const wf_hip_tick_params *)
{
    for(uint32_t s = 0; s < h->n; ++s) {
        if(h->state[s] == WF_HIP_PAUSED)
            continue;
        const bool hidden = h->state[s] == WF_HIP_HIDDEN || h->state[s] == WF_HIP_HIDDEN_TIMEOUT;
        h->silent[s] = hidden ? 1 : 0;
        float *r = h->rows.data() + (size_t)s * h->out * h->M;
        for(size_t k = 0; k < (size_t)h->out * h->M; ++k)
            r[k] = hidden ? -758.0f : -20.0f - (float)((h->sum[s] + k) % 97u);
    }
    return WF_HIP_OK;
}
uint32_t wf_hip_ring_frames(const wf_hip *) { return 1u << 20; }
int wf_hip_set_bars_mirrors(wf_hip *, uint32_t, void *const *, void *const *) { return WF_HIP_ERR_UNSUPPORTED; }
int wf_hip_bars_mirror_ready(wf_hip *, void *, void **) { return WF_HIP_ERR_INVALID; }
// the one reader (wf_hip_read): every output a deterministic function of what the stream has received
int wf_hip_read(wf_hip *h, wf_hip_output what, uint32_t first, uint32_t count, void *out_)
{
    switch(what) {
    case WF_HIP_OUT_DECIBELS:
        std::memcpy(out_, h->rows.data() + (size_t)first * h->out * h->M, (size_t)count * h->out * h->M * sizeof(float));
        return WF_HIP_OK;
    case WF_HIP_OUT_LAST_SILENT:
        std::memcpy(out_, h->silent.data() + first, count);
        return WF_HIP_OK;
    case WF_HIP_OUT_PREMIRROR:
    case WF_HIP_OUT_INPUT_RMS: {
        const size_t n = what == WF_HIP_OUT_PREMIRROR ? (size_t)count * h->disp : count;
        std::memset(out_, 0, n * sizeof(float));
        return WF_HIP_OK;
    }
    case WF_HIP_OUT_WAVEFORM_TS:
        std::memset(out_, 0, (size_t)count * sizeof(uint64_t));
        return WF_HIP_OK;
    case WF_HIP_OUT_METER: {
        float *out = static_cast<float *>(out_);
        for(uint32_t i = 0; i < count * h->cap; ++i)
            out[i] = -20.0f - (float)(h->sum[first + i / h->cap] % 31u);
        return WF_HIP_OK;
    }
    case WF_HIP_OUT_BARS: {
        float *out = static_cast<float *>(out_);
        for(size_t i = 0; i < (size_t)count * h->disp * h->bars; ++i)
            out[i] = 10.0f + (float)((h->sum[first + i / ((size_t)h->disp * h->bars)] + i) % 200u);
        return WF_HIP_OK;
    }
    case WF_HIP_OUT_VERTICES:
        std::memset(out_, 0, (size_t)count * h->disp * wf_hip_num_vertices(h) * 4 * sizeof(float));
        return WF_HIP_OK;
    case WF_HIP_OUT_VERTEX_COUNTS: {
        uint32_t *out = static_cast<uint32_t *>(out_);
        for(uint32_t i = 0; i < count * h->disp; ++i)
            out[i] = wf_hip_num_vertices(h);
        return WF_HIP_OK;
    }
    default: return WF_HIP_ERR_INVALID;
    }
}
// the pipelined reader: the mock copies at once
int wf_hip_read_async(wf_hip *h, uint32_t first, uint32_t count, const wf_hip_readback *d, uint32_t)
{
    if(d->rows) wf_hip_read(h, WF_HIP_OUT_DECIBELS, first, count, d->rows);
    if(d->last_silent) wf_hip_read(h, WF_HIP_OUT_LAST_SILENT, first, count, d->last_silent);
    if(d->bars) wf_hip_read(h, WF_HIP_OUT_BARS, first, count, d->bars);
    if(d->premirror) wf_hip_read(h, WF_HIP_OUT_PREMIRROR, first, count, d->premirror);
    if(d->vertices) wf_hip_read(h, WF_HIP_OUT_VERTICES, first, count, d->vertices);
    if(d->vertex_counts) wf_hip_read(h, WF_HIP_OUT_VERTEX_COUNTS, first, count, d->vertex_counts);
    if(d->input_rms) wf_hip_read(h, WF_HIP_OUT_INPUT_RMS, first, count, d->input_rms);
    if(d->meter) wf_hip_read(h, WF_HIP_OUT_METER, first, count, d->meter);
    return WF_HIP_OK;
}
}
