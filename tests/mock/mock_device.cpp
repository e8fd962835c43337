// tests/mock/mock_device.cpp -- TEST HARNESS: a host-only stand-in for the HIP runtime calls and the wf_hip_* entry points that
// waveform_amd/csrc/wf_hip_multi.cpp uses, so that the multi-device group -- its worker threads, shard arithmetic, peer-copy
// gather, double buffering and failure handling -- runs on a GPU-less box under AddressSanitizer / UndefinedBehaviorSanitizer and
// ThreadSanitizer (tests/test_sanitizers.py).  "Device memory" is host memory, streams execute at once, events are no-ops; a
// mock handle's bars are a function of (global stream, channel, bar, ticks so far), so every gathered copy can be checked exactly.
// Nothing here is product code; the product library links none of it.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "wf_hip.h"

static int g_devices = 4;
extern "C" void mock_set_device_count(int n) { g_devices = n; }

// ---- HIP runtime ------------------------------------------------------------------------------------------------------------
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemset(void *p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "mock"; }
}

// ---- wf_hip_* --------------------------------------------------------------------------------------------------------------
struct wf_hip {
    wf_config cfg{};
    uint32_t streams = 0, bars = 26, disp = 2;
    uint32_t id0 = 0;   // global id of stream 0 (wf_hip_push_synth's stream_id0)
    uint32_t ticks = 0;
    std::vector<uint8_t> hidden;
    std::vector<std::vector<float *>> mirrors; // wf_hip_set_bars_mirrors: all buffers of the two sets; every tick writes set mirror_next,
    uint32_t mirror_next = 0;                  // wf_hip_bars_mirror_ready hands it over and switches (as the library does: ABI 13)
    bool mirror_fresh = false;
    std::string err;
};
static thread_local std::string g_err;

extern "C" float mock_bar_value(uint32_t global_stream, uint32_t ch, uint32_t bar, uint32_t ticks)
{
    return (float)(global_stream % 4096u) + 0.25f * (float)ch + (float)bar * 0.001f + 5000.0f * (float)(ticks % 64u);
}

extern "C" {
int wf_hip_device_count(void) { return g_devices; }
const char *wf_hip_last_error(const wf_hip *h) { return h ? h->err.c_str() : g_err.c_str(); }
int wf_hip_create(const wf_config *cfg, int device, uint32_t max_streams, uint32_t, wf_hip **out)
{
    if(device < 0 || device >= g_devices) {
        g_err = "no such device";
        return WF_HIP_ERR_INVALID;
    }
    auto *h = new wf_hip;
    h->cfg = *cfg;
    h->streams = max_streams;
    h->bars = cfg->bars ? 26u : 0u;
    h->disp = cfg->stereo ? 2u : 1u;
    h->hidden.assign(max_streams, 0);
    *out = h;
    return WF_HIP_OK;
}
void wf_hip_destroy(wf_hip *h) { delete h; }
uint32_t wf_hip_display_channels(const wf_hip *h) { return h->disp; }
uint32_t wf_hip_num_bars(const wf_hip *h) { return h->bars; }
uint32_t wf_hip_capture_channels(const wf_hip *h) { return h->cfg.capture_channels; }
uint32_t wf_hip_output_channels(const wf_hip *h) { return 2; }
uint32_t wf_hip_fft_size(const wf_hip *h) { return h->cfg.fft_size; }
int wf_hip_push_audio(wf_hip *, uint32_t, uint32_t, const float *, uint32_t) { return WF_HIP_OK; }
int wf_hip_push_synth(wf_hip *h, uint32_t first, uint32_t, uint64_t, uint32_t stream_id0, uint64_t, uint32_t)
{
    h->id0 = stream_id0 - first;
    return WF_HIP_OK;
}
int wf_hip_set_hidden(wf_hip *h, uint32_t first, uint32_t count, const uint8_t *mask)
{
    for(uint32_t i = 0; i < count; ++i)
        h->hidden[first + i] = mask[i];
    return WF_HIP_OK;
}
int wf_hip_reset(wf_hip *h, uint32_t, uint32_t) { h->ticks = 0; return WF_HIP_OK; }
static void fill_bars(const wf_hip *h, uint32_t first, uint32_t count, float *out);
int wf_hip_tick(wf_hip *h, const wf_hip_tick_params *)
{
    ++h->ticks;
    if(!h->mirrors.empty() && !h->mirrors[0].empty()) {
        for(float *p : h->mirrors[h->mirror_next])
            fill_bars(h, 0, h->streams, p);
        h->mirror_fresh = true;
    }
    return WF_HIP_OK;
}
int wf_hip_set_bars_mirrors(wf_hip *h, uint32_t n, void *const *a, void *const *b)
{
    if(getenv("WF_MOCK_NO_MIRROR"))
        return WF_HIP_ERR_UNSUPPORTED;
    h->mirrors.assign(2, std::vector<float *>());
    for(uint32_t j = 0; j < n; ++j) {
        h->mirrors[0].push_back(static_cast<float *>(a[j]));
        h->mirrors[1].push_back(static_cast<float *>(b[j]));
    }
    h->mirror_next = 0;
    h->mirror_fresh = false;
    return WF_HIP_OK;
}
int wf_hip_bars_mirror_ready(wf_hip *h, void *, void **out)
{
    if(h->mirrors.empty() || h->mirrors[0].empty())
        return WF_HIP_ERR_INVALID;
    if(!h->mirror_fresh) // no tick has written the set: filled from the handle's own bars
        for(float *p : h->mirrors[h->mirror_next])
            fill_bars(h, 0, h->streams, p);
    *out = h->mirrors[h->mirror_next][0];
    h->mirror_next ^= 1u;
    h->mirror_fresh = false;
    return WF_HIP_OK;
}
int wf_hip_sync(wf_hip *) { return WF_HIP_OK; }
int wf_hip_wait_event(wf_hip *, void *) { return WF_HIP_OK; }
int wf_hip_time_begin(wf_hip *) { return WF_HIP_OK; }
int wf_hip_time_end(wf_hip *, float *ms) { *ms = 1.0f; return WF_HIP_OK; }
static void fill_bars(const wf_hip *h, uint32_t first, uint32_t count, float *out)
{
    for(uint32_t s = 0; s < count; ++s)
        for(uint32_t c = 0; c < h->disp; ++c)
            for(uint32_t b = 0; b < h->bars; ++b)
                out[((size_t)s * h->disp + c) * h->bars + b] = mock_bar_value(h->id0 + first + s, c, b, h->ticks);
}
int wf_hip_copy_bars_device_async(wf_hip *h, uint32_t first, uint32_t count, void *d_out, void *)
{
    fill_bars(h, first, count, static_cast<float *>(d_out));
    return WF_HIP_OK;
}
size_t wf_hip_output_bytes(const wf_hip *h, wf_hip_output what)
{
    switch(what) {
    case WF_HIP_OUT_BARS: return (size_t)h->disp * h->bars * sizeof(float);
    case WF_HIP_OUT_DECIBELS: return 2u * (h->cfg.fft_size / 2u) * sizeof(float);
    case WF_HIP_OUT_LAST_SILENT: return 1;
    default: return 0;
    }
}
int wf_hip_read(wf_hip *h, wf_hip_output what, uint32_t first, uint32_t count, void *out_)
{
    if(what == WF_HIP_OUT_BARS) {
        fill_bars(h, first, count, static_cast<float *>(out_));
        return WF_HIP_OK;
    }
    if(what == WF_HIP_OUT_DECIBELS) {
        float *out = static_cast<float *>(out_);
        const size_t per = 2u * (h->cfg.fft_size / 2u);
        for(size_t i = 0; i < (size_t)count * per; ++i)
            out[i] = (float)(h->id0 + first) + (float)(i / per);
        return WF_HIP_OK;
    }
    if(what == WF_HIP_OUT_LAST_SILENT) {
        uint8_t *out = static_cast<uint8_t *>(out_);
        for(uint32_t i = 0; i < count; ++i)
            out[i] = h->hidden[first + i] ? 1 : 0;
        return WF_HIP_OK;
    }
    return WF_HIP_ERR_INVALID;
}
}
