/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE (see wfref.h).
 *
 * Drives the reference plugin's WAVSource exactly as OBS does, through the
 * reference's own public entry points:
 *   obs_module_load()            src/module.cpp:35-39  -> WAVSource::register_source()
 *   WAVSource{Generic,AVX,AVX2}  src/source.hpp:349-386 (chosen like callbacks::create,
 *                                src/source.cpp:87-102, but by the caller, not CPUID)
 *   update(settings)             src/source.cpp:1077-1322
 *   capture_audio callback       src/source.cpp:490-493 / 1817-1888
 *   tick(seconds)                src/source.cpp:1324-1344 -> tick_spectrum
 *   render(effect)               src/source.cpp:1346-1358 -> render_bars / render_curve
 * The only liberty taken is reading protected members (this TU alone is compiled
 * with `protected` spelled `public`; the reference TUs are compiled untouched).
 */
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <numbers>
#include <shared_mutex>
#include <sstream>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include <cassert>
#include <cstdint>

#define protected public
#define private public
#include "source.hpp"
#undef protected
#undef private

#include "wav_source_hip.hpp" // the reference-side binding under test when isa == "hip"
#include "fake_obs_world.hpp"
#include "wfref.h"
#include "wf_synth.h"

namespace {

std::mutex g_create_mtx; // FFTW's planner and the fake world registry are not thread-safe
std::once_flag g_load_once;
std::atomic<uint64_t> g_next_id{1};

void ensure_loaded()
{
    std::call_once(g_load_once, [] { obs_module_load(); });
}

void apply_settings(obs_data *d, const char *settings)
{
    if(settings == nullptr)
        return;
    std::string s(settings);
    size_t pos = 0;
    while(pos < s.size()) {
        auto end = s.find(';', pos);
        if(end == std::string::npos)
            end = s.size();
        auto item = s.substr(pos, end - pos);
        auto eq = item.find('=');
        if(eq != std::string::npos) {
            auto key = item.substr(0, eq);
            auto val = item.substr(eq + 1);
            fakeobs::data_set_from_text(d, key.c_str(), val.c_str());
        }
        pos = end + 1;
    }
}

} // namespace

struct wfref {
    WAVSource *obj = nullptr;
    obs_source *self = nullptr;   // the visualiser source OBS would own
    obs_source *audio = nullptr;  // the audio source it captures
    obs_data *settings = nullptr;
    uint32_t sample_rate = 48000;
    int channels = 2;
};

extern "C" {

wfref_t *wfref_create(const char *isa, const char *settings, uint32_t sample_rate, int channels, uint32_t fps_num, uint32_t fps_den)
{
    ensure_loaded();
    auto info = fakeobs::registered_source_info();
    if(info == nullptr)
        return nullptr;

    std::lock_guard lock(g_create_mtx);
    auto h = new wfref();
    h->sample_rate = sample_rate;
    h->channels = channels;
    const auto id = g_next_id.fetch_add(1);
    const auto self_name = "wf_self_" + std::to_string(id);
    const auto audio_name = "wf_audio_" + std::to_string(id);
    h->self = fakeobs::create_source(self_name.c_str(), OBS_SOURCE_VIDEO | OBS_SOURCE_CUSTOM_DRAW);
    h->audio = fakeobs::create_source(audio_name.c_str(), OBS_SOURCE_AUDIO);

    h->settings = fakeobs::data_create();
    info->get_defaults(h->settings);
    fakeobs::data_set_from_text(h->settings, "audio_source", audio_name.c_str());
    apply_settings(h->settings, settings);

    fakeobs::set_audio_info(sample_rate, channels);
    fakeobs::set_video_fps(fps_num ? fps_num : 60, fps_den ? fps_den : 1);

    std::string which = isa ? isa : "generic";
    if(which == "create") {
        // the plugin's own factory (obs_source_info::create = callbacks::create, src/source.cpp:87-102), which also runs update()
        h->obj = static_cast<WAVSource *>(info->create(h->settings, h->self));
        return h;
    }
    if(which == "hip")
        h->obj = new WAVSourceHIP(h->self);
    else if(which == "avx2")
        h->obj = new WAVSourceAVX2(h->self);
    else if(which == "avx")
        h->obj = new WAVSourceAVX(h->self);
    else
        h->obj = new WAVSourceGeneric(h->self);
    h->obj->update(h->settings);
    return h;
}

void wfref_destroy(wfref_t *h)
{
    if(h == nullptr)
        return;
    std::lock_guard lock(g_create_mtx);
    delete h->obj;
    fakeobs::destroy_source(h->audio);
    fakeobs::destroy_source(h->self);
    fakeobs::data_destroy(h->settings);
    delete h;
}

void wfref_update(wfref_t *h, const char *settings)
{
    std::lock_guard lock(g_create_mtx);
    apply_settings(h->settings, settings);
    fakeobs::set_audio_info(h->sample_rate, h->channels);
    h->obj->update(h->settings);
}

void wfref_set_clock_ns(uint64_t ns) { fakeobs::set_clock_ns(ns); }
uint64_t wfref_clock_ns(void) { return fakeobs::clock_ns(); }

void wfref_push_audio(wfref_t *h, const float *ch0, const float *ch1, uint32_t frames, uint64_t timestamp_ns, int muted)
{
    audio_data pkt{};
    pkt.data[0] = (uint8_t *)ch0;
    pkt.data[1] = (uint8_t *)ch1;
    pkt.frames = frames;
    pkt.timestamp = timestamp_ns;
    fakeobs::push_audio(h->audio, &pkt, muted != 0);
}

void wfref_feed_and_tick(wfref_t *h, const float *ch0, const float *ch1, uint32_t frames, uint64_t now_ns, float seconds)
{
    fakeobs::set_clock_ns(now_ns);
    if(frames > 0) {
        const auto len = audio_frames_to_ns(h->sample_rate, frames);
        wfref_push_audio(h, ch0, ch1, frames, now_ns - len, 0);
    }
    h->obj->tick(seconds);
}

void wfref_tick(wfref_t *h, float seconds) { h->obj->tick(seconds); }
void wfref_render(wfref_t *h)
{
    fakeobs::clear_draws();
    h->obj->render(nullptr);
}
/* the gs_draw calls of the last wfref_render(): render_bars / render_curve draw once per displayed channel */
int wfref_draw_count(wfref_t *) { return (int)fakeobs::draws().size(); }
size_t wfref_draw(wfref_t *, int i, int *mode, const float **points)
{
    auto &d = fakeobs::draws();
    if(i < 0 || i >= (int)d.size())
        return 0;
    if(mode) *mode = d[(size_t)i].mode;
    if(points) *points = d[(size_t)i].points.data();
    return d[(size_t)i].num; /* vertices drawn; points holds 4 floats per vertex */
}
/* the last value a shader parameter was set to (set_shader_vars, src/source.cpp:1693-1770): returns 0 if it never was */
int wfref_shader_value(wfref_t *, const char *name, float out[4])
{
    auto &m = fakeobs::shader_values();
    auto it = m.find(name ? name : "");
    if(it == m.end())
        return 0;
    for(int i = 0; i < 4; ++i)
        out[i] = it->second[(size_t)i];
    return 1;
}
void wfref_show(wfref_t *h, int show)
{
    h->self->showing = (show != 0);
    if(show)
        h->obj->show();
    else
        h->obj->hide();
}

size_t wfref_fft_size(wfref_t *h) { return h->obj->m_fft_size; }
uint32_t wfref_capture_channels(wfref_t *h) { return h->obj->m_capture_channels; }
uint32_t wfref_output_channels(wfref_t *h) { return h->obj->m_output_channels; }
int wfref_stereo(wfref_t *h) { return h->obj->m_stereo ? 1 : 0; }
int wfref_last_silent(wfref_t *h) { return h->obj->m_last_silent ? 1 : 0; }
size_t wfref_ring_bytes(wfref_t *h, int ch) { return h->obj->m_capturebufs[ch & 1].size(); }
float wfref_gravity(wfref_t *h, float seconds) { return h->obj->get_gravity(seconds); }
float wfref_db_min(void) { return WAVSource::DB_MIN; }
const float *wfref_decibels(wfref_t *h, int ch) { return h->obj->m_decibels[ch & 1].get(); }
const float *wfref_tsmooth(wfref_t *h, int ch) { return h->obj->m_tsmooth_buf[ch & 1].get(); }
const float *wfref_window(wfref_t *h) { return h->obj->m_window_coefficients.get(); }
float wfref_window_sum(wfref_t *h) { return h->obj->m_window_sum; }
const float *wfref_slope(wfref_t *h) { return h->obj->m_slope_modifiers.get(); }
const float *wfref_rolloff(wfref_t *h) { return h->obj->m_rolloff_modifiers.get(); }
int wfref_num_bars(wfref_t *h) { return h->obj->m_num_bars; }
size_t wfref_interp_indices(wfref_t *h, const float **out)
{
    *out = h->obj->m_interp_indices.data();
    return h->obj->m_interp_indices.size();
}
size_t wfref_band_widths(wfref_t *h, const int **out)
{
    *out = h->obj->m_band_widths.data();
    return h->obj->m_band_widths.size();
}
size_t wfref_interp_kernel(wfref_t *h, const float **out, int *radius, int *size)
{
    auto &k = h->obj->m_interp_kernel;
    *out = k.weights.get();
    if(radius) *radius = k.radius;
    if(size) *size = k.size;
    return (k.weights.get() == nullptr) ? 0 : h->obj->m_interp_indices.size() * (size_t)k.size;
}
size_t wfref_bars(wfref_t *h, int ch, const float **out)
{
    auto &v = h->obj->m_interp_bufs[ch & 1];
    *out = v.data();
    return v.size();
}

int wfref_meter_mode(wfref_t *h) { return h->obj->m_meter_mode ? 1 : 0; }
float wfref_meter_val(wfref_t *h, int ch) { return h->obj->m_meter_val[ch & 1]; }
float wfref_meter_buf(wfref_t *h, int ch) { return h->obj->m_meter_buf[ch & 1]; }
float wfref_input_rms(wfref_t *h) { return h->obj->m_input_rms; }
size_t wfref_decibels_size(wfref_t *h)
{
    const bool spectrum = !h->obj->m_meter_mode && (h->obj->m_display_mode != DisplayMode::WAVEFORM);
    return spectrum ? h->obj->m_fft_size / 2 : h->obj->m_fft_size;
}

/* the class callbacks::create (or the harness) instantiated: "hip", "avx2", "avx", "generic" */
const char *wfref_class_name(wfref_t *h)
{
    if(dynamic_cast<WAVSourceHIP *>(h->obj)) return "hip";
    if(dynamic_cast<WAVSourceAVX2 *>(h->obj)) return "avx2";
    if(dynamic_cast<WAVSourceAVX *>(h->obj)) return "avx";
    return "generic";
}

int wfref_using_hip(wfref_t *h)
{
    auto p = dynamic_cast<WAVSourceHIP *>(h->obj);
    return (p != nullptr && p->using_hip()) ? 1 : 0;
}

uint64_t wfref_hip_fallback_ticks(void) { return WAVSourceHIP::fallback_ticks(); }
uint64_t wfref_hip_host_rms_updates(void) { return WAVSourceHIP::host_rms_updates(); }
uint64_t wfref_hip_device_renders(void) { return WAVSourceHIP::device_renders(); }
uint64_t wfref_hip_host_renders(void) { return WAVSourceHIP::host_renders(); }

float wfref_noise(uint64_t seed, uint32_t stream, uint32_t channel, uint64_t index)
{
    return wf_synth_noise(seed, stream, channel, index);
}

// ---- threads, as OBS runs a source (src/source.hpp:98-101) ---------------------------------------------------------------------
// One "audio" thread per source pushes 10 ms packets through the capture callback (capture_audio try_locks m_mtx for 10 ms and
// drops the packet otherwise, src/source.cpp:1822-1824), one "video" thread ticks and renders every source once per frame, one
// "UI" thread calls update() on random sources -- with a new FFT size now and then, so that buffers, plans and (WAVSourceHIP)
// the device group membership change under the other threads' feet -- and destroys / re-creates random sources (the harness
// keeps a source alive while a thread is inside it, as libobs' reference counting does).  After `seconds` everything stops and
// the sources are checked against fresh ones of the same class on a deterministic serial script: after the chaos a source
// (and, for WAVSourceHIP, its group) must compute exactly what a new one computes.
// Returns 0; 1..: that many sources differ from a fresh one; -1: a source could not be created.  stats[6] = ticks, packets,
// updates, re-creations, render calls, sources compared.
int wfref_thread_stress(const char *isa, const char *settings, int n_sources, double seconds, uint64_t seed, uint64_t *stats)
{
    struct Slot {
        std::shared_mutex life; // shared: a thread is inside the source; exclusive: it is being destroyed / re-created
        wfref_t *h = nullptr;
    };
    std::vector<Slot> slots((size_t)n_sources);
    for(auto &sl : slots) {
        sl.h = wfref_create(isa, settings, 48000, 2, 60, 1);
        if(sl.h == nullptr)
            return -1;
    }
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> n_ticks{0}, n_packets{0}, n_updates{0}, n_recreate{0}, n_render{0};
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ns = [&] { return 1000000000ull + (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::vector<std::thread> threads;
    for(int i = 0; i < n_sources; ++i)
        threads.emplace_back([&, i] { // audio thread of source i: 480 frames every millisecond (ten times real time)
            std::vector<float> a(2 * 480);
            const auto key0 = wf_synth_key(seed, (uint32_t)i, 0), key1 = wf_synth_key(seed, (uint32_t)i, 1);
            uint64_t pos = 0;
            while(!stop.load(std::memory_order_relaxed)) {
                for(int k = 0; k < 480; ++k) {
                    a[(size_t)k] = wf_synth_sample(key0, pos + (uint64_t)k);
                    a[480 + (size_t)k] = wf_synth_sample(key1, pos + (uint64_t)k);
                }
                pos += 480;
                {
                    std::shared_lock alive(slots[(size_t)i].life);
                    const uint64_t now = now_ns();
                    fakeobs::set_clock_ns(now);
                    wfref_push_audio(slots[(size_t)i].h, a.data(), a.data() + 480, 480, now - audio_frames_to_ns(48000, 480), 0);
                }
                n_packets.fetch_add(1, std::memory_order_relaxed);
                std::this_thread::sleep_for(std::chrono::microseconds(1000));
            }
        });
    threads.emplace_back([&] { // video thread: tick every source, then render every source
        while(!stop.load(std::memory_order_relaxed)) {
            for(auto &sl : slots) {
                std::shared_lock alive(sl.life);
                fakeobs::set_clock_ns(now_ns());
                wfref_tick(sl.h, 1.0f / 60.0f);
                n_ticks.fetch_add(1, std::memory_order_relaxed);
            }
            for(auto &sl : slots) {
                std::shared_lock alive(sl.life);
                wfref_render(sl.h);
                n_render.fetch_add(1, std::memory_order_relaxed);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(1500));
        }
    });
    threads.emplace_back([&] { // UI thread
        uint64_t r = seed * 6364136223846793005ull + 1442695040888963407ull;
        auto next = [&] { r = r * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(r >> 33); };
        const char *sizes[] = {"fft_size=1024", "fft_size=2048", "fft_size=4096", "fft_size=4096", "fft_size=800"};
        while(!stop.load(std::memory_order_relaxed)) {
            auto &sl = slots[next() % (uint32_t)n_sources];
            const uint32_t what = next() % 8u;
            if(what == 0) { // destroy + create
                std::unique_lock gone(sl.life);
                wfref_destroy(sl.h);
                sl.h = wfref_create(isa, settings, 48000, 2, 60, 1);
                n_recreate.fetch_add(1, std::memory_order_relaxed);
            } else {
                std::shared_lock alive(sl.life);
                fakeobs::set_clock_ns(now_ns());
                if(what < 4)
                    wfref_update(sl.h, sizes[next() % 5u]);
                else if(what == 4)
                    wfref_show(sl.h, (int)(next() & 1u));
                else
                    wfref_update(sl.h, "");
                n_updates.fetch_add(1, std::memory_order_relaxed);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(2500));
        }
    });
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for(auto &t : threads)
        t.join();
    // after the chaos: every survivor, reset by update(), against a fresh source of the same class on one serial script
    int differ = 0;
    const int hop = 800, script_ticks = 12;
    std::vector<float> a(2 * (size_t)hop);
    std::vector<wfref_t *> fresh((size_t)n_sources, nullptr);
    for(int i = 0; i < n_sources; ++i) {
        fresh[(size_t)i] = wfref_create(isa, settings, 48000, 2, 60, 1);
        if(fresh[(size_t)i] == nullptr)
            return -1;
        wfref_show(slots[(size_t)i].h, 1);
        wfref_update(slots[(size_t)i].h, "fft_size=2048");
        wfref_update(fresh[(size_t)i], "fft_size=2048");
    }
    uint64_t now = now_ns() + 1000000000ull;
    for(int t = 0; t < script_ticks; ++t) {
        now += audio_frames_to_ns(48000, (uint64_t)hop);
        for(int i = 0; i < n_sources; ++i) {
            const auto key0 = wf_synth_key(seed ^ 0x55, (uint32_t)i, 0), key1 = wf_synth_key(seed ^ 0x55, (uint32_t)i, 1);
            for(int k = 0; k < hop; ++k) {
                a[(size_t)k] = wf_synth_sample(key0, (uint64_t)t * (uint64_t)hop + (uint64_t)k);
                a[(size_t)hop + (size_t)k] = wf_synth_sample(key1, (uint64_t)t * (uint64_t)hop + (uint64_t)k);
            }
            wfref_feed_and_tick(slots[(size_t)i].h, a.data(), a.data() + hop, (uint32_t)hop, now, 1.0f / 60.0f);
            wfref_feed_and_tick(fresh[(size_t)i], a.data(), a.data() + hop, (uint32_t)hop, now, 1.0f / 60.0f);
        }
    }
    for(int i = 0; i < n_sources; ++i) {
        const size_t n = wfref_fft_size(slots[(size_t)i].h) / 2;
        bool same = wfref_fft_size(fresh[(size_t)i]) / 2 == n && wfref_last_silent(slots[(size_t)i].h) == wfref_last_silent(fresh[(size_t)i]);
        for(int c = 0; c < 2 && same; ++c)
            same = std::memcmp(wfref_decibels(slots[(size_t)i].h, c), wfref_decibels(fresh[(size_t)i], c), n * sizeof(float)) == 0;
        differ += same ? 0 : 1;
    }
    if(stats) {
        stats[0] = n_ticks.load(); stats[1] = n_packets.load(); stats[2] = n_updates.load(); stats[3] = n_recreate.load();
        stats[4] = n_render.load(); stats[5] = (uint64_t)n_sources;
    }
    for(int i = 0; i < n_sources; ++i) {
        wfref_destroy(fresh[(size_t)i]);
        wfref_destroy(slots[(size_t)i].h);
    }
    return differ;
}

// wfref_bench with a video_render behind every frame's ticks (what OBS does: tick every source, then render every source)
static std::atomic<int> g_bench_render{0};
void wfref_bench_set_render(int on) { g_bench_render.store(on); }

double wfref_bench(const char *isa, const char *settings, uint32_t sample_rate, int channels, int n_streams, int n_threads,
                   int warmup_ticks, int timed_ticks, int hop, uint64_t seed, double *elapsed_s)
{
    if(n_streams <= 0 || n_threads <= 0 || hop <= 0)
        return 0.0;
    n_threads = std::min(n_threads, n_streams);
    std::vector<wfref_t *> streams((size_t)n_streams, nullptr);
    for(int s = 0; s < n_streams; ++s) {
        streams[(size_t)s] = wfref_create(isa, settings, sample_rate, channels, 60, 1);
        if(streams[(size_t)s] == nullptr)
            return 0.0;
    }
    const int cap_ch = (int)wfref_capture_channels(streams[0]);
    const int total_ticks = warmup_ticks + timed_ticks;
    const uint64_t tick_ns = audio_frames_to_ns(sample_rate, (uint64_t)hop);

    // Noise is generated outside the timed region: one pool per thread, long enough that
    // every (stream, tick) sees a different slice.  Generating it is not part of the
    // reference's path; pushing it through capture_audio (CircularBuffer) is.
    const size_t pool = (size_t)hop * 64 + 1024;
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<double> t_elapsed((size_t)n_threads, 0.0);
    std::vector<std::thread> threads;
    for(int t = 0; t < n_threads; ++t) {
        threads.emplace_back([&, t] {
            const int lo = (int)((int64_t)n_streams * t / n_threads);
            const int hi = (int)((int64_t)n_streams * (t + 1) / n_threads);
            std::vector<float> noise[2];
            for(int c = 0; c < 2; ++c) {
                noise[c].resize(pool);
                const auto key = wf_synth_key(seed, (uint32_t)(0x40000000u + (uint32_t)t), (uint32_t)c);
                for(size_t i = 0; i < pool; ++i)
                    noise[c][i] = wf_synth_sample(key, i);
            }
            uint64_t now = 1000000000ull;
            auto run = [&](int ticks, int tick0) {
                for(int k = 0; k < ticks; ++k) {
                    now += tick_ns;
                    for(int s = lo; s < hi; ++s) {
                        const size_t off = ((size_t)(tick0 + k) * 7919u + (size_t)s * 104729u) % (pool - (size_t)hop);
                        wfref_feed_and_tick(streams[(size_t)s], noise[0].data() + off, (cap_ch > 1) ? noise[1].data() + off : nullptr,
                                            (uint32_t)hop, now, 1.0f / 60.0f);
                    }
                    if(g_bench_render.load(std::memory_order_relaxed))
                        for(int s = lo; s < hi; ++s)
                            wfref_render(streams[(size_t)s]);
                }
            };
            run(warmup_ticks, 0);
            ready.fetch_add(1);
            while(!go.load(std::memory_order_acquire))
                std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            run(timed_ticks, warmup_ticks);
            const auto t1 = std::chrono::steady_clock::now();
            t_elapsed[(size_t)t] = std::chrono::duration<double>(t1 - t0).count();
        });
    }
    while(ready.load() < n_threads)
        std::this_thread::yield();
    const auto w0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for(auto &th : threads)
        th.join();
    const auto w1 = std::chrono::steady_clock::now();
    (void)total_ticks;
    const double wall = std::chrono::duration<double>(w1 - w0).count();
    if(elapsed_s != nullptr)
        *elapsed_s = wall;
    for(auto s : streams)
        wfref_destroy(s);
    const double spectra = (double)n_streams * (double)cap_ch * (double)timed_ticks;
    return (wall > 0.0) ? spectra / wall : 0.0;
}

} // extern "C"
