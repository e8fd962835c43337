/*
 * wav_source_hip.cpp -- see wav_source_hip.hpp.  Reference-side binding (compiled with the plugin, or here with the
 * oracle harness).  libwaveform_hip.so is loaded with dlopen so that the plugin still loads on machines without it;
 * when it is missing, or no gfx950 device is present, every call falls through to WAVSourceGeneric -- the reference's
 * own CPU path -- which is the reference's failure style (degrade and log, never throw across the C boundary,
 * src/source_generic.cpp:105-108).
 */
#include "wav_source_hip.hpp"
#include "log.hpp"

#include <dlfcn.h>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>

namespace {

struct HipApi {
    void *lib = nullptr;
    decltype(&wf_hip_device_count) device_count = nullptr;
    decltype(&wf_hip_create) create = nullptr;
    decltype(&wf_hip_destroy) destroy = nullptr;
    decltype(&wf_hip_push_audio) push_audio = nullptr;
    decltype(&wf_hip_tick) tick = nullptr;
    decltype(&wf_hip_set_hidden) set_hidden = nullptr;
    decltype(&wf_hip_read) read = nullptr;
    decltype(&wf_hip_read_async) read_async = nullptr;
    decltype(&wf_hip_enable_input_rms) enable_input_rms = nullptr;
    decltype(&wf_hip_last_error) last_error = nullptr;
    decltype(&wf_hip_reset) reset = nullptr;
    decltype(&wf_hip_push_audio_ragged_async) push_audio_ragged_async = nullptr;
    decltype(&wf_hip_ingest_done) ingest_done = nullptr;
    decltype(&wf_hip_readback_done) readback_done = nullptr;
    decltype(&wf_hip_set_input_rms) set_input_rms = nullptr;
    decltype(&wf_hip_host_alloc) host_alloc = nullptr;
    decltype(&wf_hip_host_free) host_free = nullptr;
    decltype(&wf_hip_push_rms_ragged_async) push_rms_ragged_async = nullptr;
    decltype(&wf_hip_set_stream_delay) set_stream_delay = nullptr;
    decltype(&wf_hip_set_stream_audio_ts) set_stream_audio_ts = nullptr;
    decltype(&wf_hip_output_channels) output_channels = nullptr;
    decltype(&wf_hip_num_vertices) num_vertices = nullptr;
    decltype(&wf_hip_num_bars) num_bars = nullptr;
    decltype(&wf_hip_display_channels) display_channels = nullptr;
    decltype(&wf_hip_ring_frames) ring_frames = nullptr;
    bool ok = false;
};

HipApi &api()
{
    static HipApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *path = std::getenv("WF_HIP_LIBRARY");
        a.lib = dlopen(path ? path : "libwaveform_hip.so", RTLD_NOW | RTLD_LOCAL);
        if(a.lib == nullptr)
            return;
#define WF_SYM(name)                                                           \
    a.name = reinterpret_cast<decltype(a.name)>(dlsym(a.lib, "wf_hip_" #name)); \
    if(a.name == nullptr)                                                      \
        return;
        WF_SYM(device_count)
        WF_SYM(create)
        WF_SYM(destroy)
        WF_SYM(push_audio)
        WF_SYM(tick)
        WF_SYM(set_hidden)
        WF_SYM(read)
        WF_SYM(read_async)
        WF_SYM(enable_input_rms)
        WF_SYM(last_error)
        WF_SYM(reset)
        WF_SYM(push_audio_ragged_async)
        WF_SYM(ingest_done)
        WF_SYM(readback_done)
        WF_SYM(set_input_rms)
        WF_SYM(host_alloc)
        WF_SYM(host_free)
        WF_SYM(push_rms_ragged_async)
        WF_SYM(set_stream_delay)
        WF_SYM(set_stream_audio_ts)
        WF_SYM(output_channels)
        WF_SYM(num_vertices)
        WF_SYM(num_bars)
        WF_SYM(display_channels)
        WF_SYM(ring_frames)
#undef WF_SYM
        // struct wf_config and the entry points above must be the ones this file was compiled against
        auto abi = reinterpret_cast<decltype(&wf_hip_abi_version)>(dlsym(a.lib, "wf_hip_abi_version"));
        if(abi == nullptr || abi() != WF_HIP_ABI_VERSION)
            return;
        a.ok = true;
    });
    return a;
}

std::atomic<uint64_t> g_fallback_ticks{0};
std::atomic<uint64_t> g_host_rms_updates{0}; // update_input_rms calls that ran the reference's host loop
std::atomic<uint64_t> g_device_renders{0}, g_host_renders{0}; // render() of HIP-configured spectrum sources: from the device / by the reference's loops

bool device_render_mode()
{
    const char *e = std::getenv("WF_HIP_RENDER"); // 0: the reference's render() keeps interpolating and filling vertices on the host
    return e == nullptr || e[0] != '0';
}

bool batched_mode()
{
    const char *e = std::getenv("WF_HIP_BATCHED"); // 0: every source its own handle, results inside the call
    return e == nullptr || e[0] != '0';
}

uint32_t group_capacity()
{
    const char *e = std::getenv("WF_HIP_BATCH_CAPACITY");
    const long v = e ? std::atol(e) : 64;
    return (uint32_t)std::min<long>(std::max<long>(v, 1), 16384);
}

} // namespace

// One handle shared by the sources of one configuration: sources are streams of its batch.
//
// A video frame, seen from the group: OBS ticks its active sources one after the other on the video thread
// (reference callbacks::tick, src/source.cpp:481-484).  Every member's tick_spectrum
//   1. collects: copies its row of the PREVIOUS frame's tick out of the page-locked readback buffer (one frame of latency:
//      the copy was enqueued 16 ms ago and has long landed), and
//   2. submits: writes the samples its window gained since its last tick into the page-locked staging block of the batch
//      being assembled, with its show / hide / timeout state and its m_input_rms.
// The member that completes the frame -- or one that comes round again while an incomplete batch is waiting -- flushes:
// state masks and RMS values that changed, ONE ragged ingest (wf_hip_push_audio_ragged_async), ONE wf_hip_tick, ONE
// readback (wf_hip_read_async); members that did not show up are paused for that tick.  Nothing in a flush waits for
// the device.
struct WFHipGroup {
    wf_config cfg{};
    wf_hip *h = nullptr;
    uint32_t capacity = 0, cap_ch = 0, out_ch = 0, N = 0, M = 0;
    std::vector<WAVSourceHIP *> member;   // [capacity] or nullptr
    uint32_t members = 0;
    uint64_t batch = 1;                   // the batch being assembled (batch - 1: the last one flushed)
    std::vector<uint64_t> submitted;      // [capacity] the batch a member last submitted to
    uint32_t n_submitted = 0;
    float *stage[2] = {nullptr, nullptr}; // page-locked [capacity][cap_ch][N]
    float *rows[2] = {nullptr, nullptr};  // page-locked [capacity][out_ch][M]
    uint8_t *silent[2] = {nullptr, nullptr};
    bool rows_valid[2] = {false, false};
    std::vector<uint32_t> frames;         // [capacity] frames each member staged for this batch
    std::vector<uint8_t> state, state_dev;
    std::vector<float> rms, rms_dev;
    // volume normalisation produced on the device (WAVSourceHIP::update_input_rms): the squared peaks every member's
    // sync_rms_buffer consumed this frame, page-locked [capacity][sq_max] per ingest slot, and m_input_rms coming back
    bool rms_feed = false;
    uint32_t sq_max = 0;
    float *sq_stage[2] = {nullptr, nullptr};
    float *rms_back[2] = {nullptr, nullptr};
    std::vector<uint32_t> sq_frames;      // [capacity] values staged for the batch being assembled
    float seconds = 1.0f / 60.0f;
    bool failed = false;
    int device = 0;
    // the display from the device (cfg.bars / cfg.curve with cfg.vertices): bar tops / curve points, vertices and vertex counts
    // of every member come back with the rows
    bool display = false;
    uint32_t disp_ch = 0, points = 0, per_row = 0;
    float *bars[2] = {nullptr, nullptr};      // page-locked [capacity][disp_ch][points]
    float *verts[2] = {nullptr, nullptr};     // page-locked [capacity][disp_ch][per_row][4]
    uint32_t *vcounts[2] = {nullptr, nullptr}; // page-locked [capacity][disp_ch]
    float *pre[2] = {nullptr, nullptr};       // page-locked [capacity][disp_ch]: mirrored axis, the value above the middle before the mirror (WF_HIP_OUT_PREMIRROR)

    bool create(const wf_config &c, int dev)
    {
        auto &a = api();
        cfg = c;
        device = dev;
        capacity = group_capacity();
        if(a.create(&c, dev, capacity, 0, &h) != WF_HIP_OK) {
            h = nullptr;
            return false;
        }
        cap_ch = c.capture_channels;
        N = c.fft_size;
        M = c.fft_size / 2;
        out_ch = ((cap_ch > 1) || c.stereo) ? 2u : 1u; // src/source.cpp:1171
        display = (c.bars != 0 || c.curve != 0) && a.num_bars(h) > 0 && (c.curve != 0 || a.num_vertices(h) > 0);
        if(display) {
            disp_ch = a.display_channels(h);
            points = a.num_bars(h);
            per_row = a.num_vertices(h); // 0 for a curve: its points come back, the strip is written on the host
            for(int i = 0; i < 2; ++i) {
                bars[i] = static_cast<float *>(a.host_alloc((size_t)capacity * disp_ch * points * sizeof(float)));
                if(bars[i] == nullptr)
                    return false;
                if(c.mirror_freq_axis) {
                    pre[i] = static_cast<float *>(a.host_alloc((size_t)capacity * disp_ch * sizeof(float)));
                    if(pre[i] == nullptr)
                        return false;
                }
                if(per_row == 0)
                    continue;
                verts[i] = static_cast<float *>(a.host_alloc((size_t)capacity * disp_ch * per_row * 4 * sizeof(float)));
                vcounts[i] = static_cast<uint32_t *>(a.host_alloc((size_t)capacity * disp_ch * sizeof(uint32_t)));
                if(verts[i] == nullptr || vcounts[i] == nullptr)
                    return false;
            }
        }
        member.assign(capacity, nullptr);
        submitted.assign(capacity, 0);
        frames.assign(capacity, 0);
        state.assign(capacity, WF_HIP_PAUSED);
        state_dev.assign(capacity, WF_HIP_SHOWN);
        rms.assign(capacity, 0.0f);
        rms_dev.assign(capacity, -1.0f);
        sq_frames.assign(capacity, 0);
        if(c.normalize_volume && std::getenv("WF_HIP_HOST_RMS") == nullptr && a.enable_input_rms(h, 1) == WF_HIP_OK) {
            rms_feed = true;
            sq_max = c.sample_rate & ~15u; // m_input_rms_size: more than a window's worth per frame is never needed
            for(int i = 0; i < 2; ++i) {
                sq_stage[i] = static_cast<float *>(a.host_alloc((size_t)capacity * sq_max * sizeof(float)));
                rms_back[i] = static_cast<float *>(a.host_alloc((size_t)capacity * sizeof(float)));
                if(sq_stage[i] == nullptr || rms_back[i] == nullptr)
                    return false;
            }
        }
        for(int i = 0; i < 2; ++i) {
            stage[i] = static_cast<float *>(a.host_alloc((size_t)capacity * cap_ch * N * sizeof(float)));
            rows[i] = static_cast<float *>(a.host_alloc((size_t)capacity * out_ch * M * sizeof(float)));
            silent[i] = static_cast<uint8_t *>(a.host_alloc(capacity));
            if(stage[i] == nullptr || rows[i] == nullptr || silent[i] == nullptr)
                return false;
        }
        return true;
    }

    ~WFHipGroup()
    {
        auto &a = api();
        if(h)
            a.destroy(h); // synchronises its streams first
        for(int i = 0; i < 2; ++i) {
            if(stage[i]) a.host_free(stage[i]);
            if(rows[i]) a.host_free(rows[i]);
            if(silent[i]) a.host_free(silent[i]);
            if(sq_stage[i]) a.host_free(sq_stage[i]);
            if(rms_back[i]) a.host_free(rms_back[i]);
            if(bars[i]) a.host_free(bars[i]);
            if(verts[i]) a.host_free(verts[i]);
            if(vcounts[i]) a.host_free(vcounts[i]);
        }
    }

    // everything the frame's members staged goes to the device as one batch; returns false when the device path failed
    bool flush()
    {
        auto &a = api();
        const uint32_t b = (uint32_t)(batch & 1);
        for(uint32_t i = 0; i < capacity; ++i)
            if(submitted[i] != batch) { // not ticked in this frame (inactive source, free slot): left exactly as it is
                state[i] = WF_HIP_PAUSED;
                frames[i] = 0;
            }
        bool ok = true;
        if(state != state_dev) {
            ok = a.set_hidden(h, 0, capacity, state.data()) == WF_HIP_OK;
            state_dev = state;
        }
        if(ok && rms_feed) {
            ok = a.push_rms_ragged_async(h, 0, capacity, sq_stage[b], sq_frames.data(), sq_max, b) == WF_HIP_OK;
            std::fill(sq_frames.begin(), sq_frames.end(), 0u);
        } else if(ok && cfg.normalize_volume && rms != rms_dev) {
            ok = a.set_input_rms(h, 0, capacity, rms.data()) == WF_HIP_OK;
            rms_dev = rms;
        }
        ok = ok && a.push_audio_ragged_async(h, 0, capacity, stage[b], frames.data(), N, b) == WF_HIP_OK;
        wf_hip_tick_params p{};
        p.seconds = seconds;
        ok = ok && a.tick(h, &p) == WF_HIP_OK;
        if(ok) { // the frame: rows + m_last_silent, and behind them what the members draw from one frame later
            wf_hip_readback dst{};
            dst.rows = rows[b];
            dst.last_silent = silent[b];
            if(rms_feed)
                dst.input_rms = rms_back[b];
            if(display) {
                dst.bars = bars[b];
                dst.vertices = per_row ? verts[b] : nullptr;
                dst.vertex_counts = per_row ? vcounts[b] : nullptr;
                dst.premirror = pre[b];
            }
            ok = a.read_async(h, 0, capacity, &dst, b) == WF_HIP_OK;
        }
        rows_valid[b] = ok;
        ++batch;
        n_submitted = 0;
        std::fill(frames.begin(), frames.end(), 0u);
        if(!ok) {
            LogWarn << "HIP batch tick failed (" << a.last_error(h) << "); its sources fall back to the CPU path";
            failed = true;
        }
        return ok;
    }
};

// The level meter's batch (same idea as WFHipGroup; reference tick_meter src/source_generic.cpp:182-269): every member's
// tick_meter collects the levels the PREVIOUS frame's tick left for it and submits what its own tick_meter would consume this
// frame -- every captured frame older than the A/V-sync point, :201-220 -- with its show / hide / timeout state; the member that
// completes the frame flushes: ONE ragged ingest, ONE wf_hip_tick (meter_tick_kernel over all streams), ONE readback.
struct WFHipMeterGroup {
    wf_config cfg{};
    wf_hip *h = nullptr;
    int device = 0;
    uint32_t capacity = 0, cap_ch = 0, ring_cap = 0;
    std::vector<WAVSourceHIP *> member;
    uint32_t members = 0;
    uint64_t batch = 1;
    std::vector<uint64_t> submitted;
    uint32_t n_submitted = 0;
    std::vector<std::vector<float>> pending; // [capacity] what the member consumed this frame: [cap_ch][frames[i]]
    std::vector<uint32_t> frames;
    std::vector<uint8_t> state, state_dev;
    float *stage[2] = {nullptr, nullptr};    // page-locked [capacity][cap_ch][stage_frames[b]], rebuilt at every flush
    size_t stage_floats[2] = {0, 0};
    float *levels[2] = {nullptr, nullptr};   // page-locked [capacity][cap_ch]
    uint8_t *silent[2] = {nullptr, nullptr};
    bool valid[2] = {false, false};
    float seconds = 1.0f / 60.0f;
    bool failed = false;
    // waveform display (cfg.waveform): the same frame-by-frame hand-over, with every member's A/V-sync reserve and audio
    // timestamp, and the rows of the whole group coming back instead of the levels
    bool wave = false;
    uint32_t out_ch = 0, width = 0;
    std::vector<uint32_t> delay, delay_dev;  // [capacity] the reserve tick_waveform leaves in the ring (:291-292), frames
    std::vector<uint64_t> audio_ts;          // [capacity] m_audio_ts
    float *rows[2] = {nullptr, nullptr};     // page-locked [capacity][out_ch][width]

    bool create(const wf_config &c, int dev)
    {
        auto &a = api();
        cfg = c;
        device = dev;
        capacity = group_capacity();
        if(a.create(&c, dev, capacity, 0, &h) != WF_HIP_OK) {
            h = nullptr;
            return false;
        }
        cap_ch = c.capture_channels;
        ring_cap = a.ring_frames(h);
        wave = c.waveform != 0;
        member.assign(capacity, nullptr);
        submitted.assign(capacity, 0);
        pending.assign(capacity, {});
        frames.assign(capacity, 0);
        state.assign(capacity, WF_HIP_PAUSED);
        state_dev.assign(capacity, WF_HIP_SHOWN);
        if(wave) {
            out_ch = a.output_channels(h);
            width = c.fft_size;
            delay.assign(capacity, 0);
            delay_dev.assign(capacity, 0);
            audio_ts.assign(capacity, 0);
        }
        for(int i = 0; i < 2; ++i) {
            silent[i] = static_cast<uint8_t *>(a.host_alloc(capacity));
            if(wave)
                rows[i] = static_cast<float *>(a.host_alloc((size_t)capacity * out_ch * width * sizeof(float)));
            else
                levels[i] = static_cast<float *>(a.host_alloc((size_t)capacity * cap_ch * sizeof(float)));
            if(silent[i] == nullptr || (wave ? rows[i] : levels[i]) == nullptr)
                return false;
        }
        return true;
    }

    ~WFHipMeterGroup()
    {
        auto &a = api();
        if(h)
            a.destroy(h);
        for(int i = 0; i < 2; ++i) {
            if(stage[i]) a.host_free(stage[i]);
            if(levels[i]) a.host_free(levels[i]);
            if(rows[i]) a.host_free(rows[i]);
            if(silent[i]) a.host_free(silent[i]);
        }
    }

    bool flush()
    {
        auto &a = api();
        const uint32_t b = (uint32_t)(batch & 1);
        uint32_t maxf = 0;
        for(uint32_t i = 0; i < capacity; ++i) {
            if(submitted[i] != batch) { // not ticked in this frame: left exactly as it is
                state[i] = WF_HIP_PAUSED;
                frames[i] = 0;
            }
            maxf = std::max(maxf, frames[i]);
        }
        bool ok = true;
        if(state != state_dev) {
            ok = a.set_hidden(h, 0, capacity, state.data()) == WF_HIP_OK;
            state_dev = state;
        }
        if(ok && maxf > 0) {
            a.ingest_done(h, b); // the slot's block has been copied out (two frames ago)
            const size_t need = (size_t)capacity * cap_ch * maxf;
            if(stage_floats[b] < need) {
                if(stage[b]) a.host_free(stage[b]);
                stage_floats[b] = need + need / 2;
                stage[b] = static_cast<float *>(a.host_alloc(stage_floats[b] * sizeof(float)));
                if(stage[b] == nullptr) {
                    stage_floats[b] = 0;
                    ok = false;
                }
            }
            if(ok) {
                for(uint32_t i = 0; i < capacity; ++i)
                    for(uint32_t c = 0; c < cap_ch && frames[i]; ++c)
                        std::memcpy(stage[b] + ((size_t)i * cap_ch + c) * maxf, pending[i].data() + (size_t)c * frames[i], (size_t)frames[i] * sizeof(float));
                ok = a.push_audio_ragged_async(h, 0, capacity, stage[b], frames.data(), maxf, b) == WF_HIP_OK;
            }
        }
        if(ok && wave) {
            if(delay != delay_dev) {
                ok = a.set_stream_delay(h, 0, capacity, delay.data()) == WF_HIP_OK;
                delay_dev = delay;
            }
            ok = ok && a.set_stream_audio_ts(h, 0, capacity, audio_ts.data()) == WF_HIP_OK;
        }
        wf_hip_tick_params p{};
        p.seconds = seconds;
        ok = ok && a.tick(h, &p) == WF_HIP_OK;
        if(ok) {
            wf_hip_readback dst{};
            dst.last_silent = silent[b];
            if(wave)
                dst.rows = rows[b];
            else
                dst.meter = levels[b];
            ok = a.read_async(h, 0, capacity, &dst, b) == WF_HIP_OK;
        }
        valid[b] = ok;
        ++batch;
        n_submitted = 0;
        std::fill(frames.begin(), frames.end(), 0u);
        if(!ok) {
            LogWarn << "HIP " << (wave ? "waveform" : "meter") << " batch tick failed (" << a.last_error(h) << "); its sources fall back to the CPU path";
            failed = true;
        }
        return ok;
    }
};

namespace {

struct Registry {
    std::mutex mtx;
    std::vector<std::unique_ptr<WFHipGroup>> groups;
    std::vector<std::unique_ptr<WFHipMeterGroup>> mgroups;
};
Registry &registry()
{
    static Registry *r = new Registry; // never destroyed: sources an exiting host leaks must not call into an unloaded runtime
    return *r;
}

} // namespace

uint64_t WAVSourceHIP::fallback_ticks() { return g_fallback_ticks.load(); }
uint64_t WAVSourceHIP::host_rms_updates() { return g_host_rms_updates.load(); }
uint64_t WAVSourceHIP::device_renders() { return g_device_renders.load(); }
uint64_t WAVSourceHIP::host_renders() { return g_host_renders.load(); }

bool WAVSourceHIP::available()
{
    auto &a = api();
    return a.ok && a.device_count() > 0;
}

WAVSourceHIP::~WAVSourceHIP()
{
    std::lock_guard lock(m_mtx);
    hip_release();
}

void WAVSourceHIP::hip_release()
{
    if(m_hip != nullptr) {
        api().destroy(m_hip);
        m_hip = nullptr;
    }
    if(m_mgroup != nullptr) {
        auto &r = registry();
        std::lock_guard lock(r.mtx);
        auto g = m_mgroup;
        m_mgroup = nullptr;
        g->member[m_slot] = nullptr;
        if(g->submitted[m_slot] == g->batch && g->n_submitted > 0)
            --g->n_submitted;
        g->submitted[m_slot] = 0;
        g->frames[m_slot] = 0;
        g->state[m_slot] = WF_HIP_PAUSED;
        if(--g->members == 0)
            r.mgroups.erase(std::remove_if(r.mgroups.begin(), r.mgroups.end(), [g](const auto &p) { return p.get() == g; }), r.mgroups.end());
    }
    if(m_group != nullptr) {
        auto &r = registry();
        std::lock_guard lock(r.mtx);
        auto g = m_group;
        m_group = nullptr;
        g->member[m_slot] = nullptr;
        if(g->submitted[m_slot] == g->batch && g->n_submitted > 0)
            --g->n_submitted; // what it staged for the batch in assembly is dropped with it
        g->submitted[m_slot] = 0;
        g->frames[m_slot] = 0;
        g->sq_frames[m_slot] = 0;
        g->state[m_slot] = WF_HIP_PAUSED;
        if(--g->members == 0)
            r.groups.erase(std::remove_if(r.groups.begin(), r.groups.end(), [g](const auto &p) { return p.get() == g; }), r.groups.end());
    }
}

// WAVSource members -> wf_config (include/wf_config.h lists the member behind every field)
bool WAVSourceHIP::hip_configure()
{
    hip_release();
    if(!available() || (m_capture_channels == 0))
        return false;
    wf_config c{};
    c.waveform = (m_display_mode == DisplayMode::WAVEFORM) ? 1u : 0u; // m_fft_size is m_width in this mode
    c.meter = m_meter_mode ? 1u : 0u;   // update() has already applied the mode's overrides to the members below
    c.meter_rms = m_meter_rms ? 1u : 0u;
    c.meter_ms = m_meter_ms;
    c.fft_size = (uint32_t)m_fft_size;
    c.sample_rate = m_audio_info.samples_per_sec;
    c.capture_channels = m_capture_channels;
    c.stereo = m_stereo ? 1u : 0u;
    c.window = (int32_t)m_window_func;       // FFTWindow and wf_window share their numbering
    c.sine_exponent = m_sine_exponent;
    c.tsmoothing = (int32_t)m_tsmoothing;    // TSmoothingMode / wf_tsmoothing likewise
    c.gravity = m_gravity;
    c.fast_peaks = m_fast_peaks ? 1u : 0u;
    c.slope = m_slope;
    c.rolloff_q = m_rolloff_q;
    c.rolloff_rate = m_rolloff_rate;
    c.cutoff_low = m_cutoff_low;
    c.cutoff_high = m_cutoff_high;
    c.floor_db = m_floor;
    c.ceiling_db = m_ceiling;
    c.normalize_volume = m_normalize_volume ? 1u : 0u;
    c.volume_target = m_volume_target;
    c.max_gain = m_max_gain;
    c.bars = 0; // (hip_display_config below sets the display fields where render() is served from the device)
    c.interp_mode = (int32_t)m_interp_mode;
    c.log_scale = m_log_scale ? 1u : 0u;
    c.mirror_freq_axis = m_mirror_freq_axis ? 1u : 0u;
    c.width = m_width;
    c.height = m_height;
    c.bar_width = m_bar_width;
    c.bar_gap = m_bar_gap;
    c.channel_spacing = m_channel_spacing;
    c.min_bar_height = m_min_bar_height;
    c.rounded_caps = m_rounded_caps ? 1u : 0u;
    m_hip_have_prev = false;
    m_hip_display = m_hip_display_valid = false;
    const wf_config rows_only = c;
    if(!c.waveform && !c.meter && device_render_mode() && m_vbuf != nullptr)
        hip_display_config(c);
    const auto has_display = [](const wf_config &k) { return k.bars != 0 || k.curve != 0; };
    for(int attempt = 0; attempt < 2; ++attempt) {
    // (second attempt: the same configuration without the display -- e.g. a filtered curve whose staging the device refuses:
    // the rows still come from the device, render() keeps interpolating on the host)
    if(attempt == 1) {
        if(!has_display(c))
            break;
        c = rows_only;
    }
    if(!c.waveform && !c.meter && batched_mode()) {
        // spectrum display: join (or open) the batch of this configuration
        auto &r = registry();
        std::lock_guard lock(r.mtx);
        WFHipGroup *g = nullptr;
        for(auto &p : r.groups)
            if(!p->failed && p->members < p->capacity && std::memcmp(&p->cfg, &c, sizeof(c)) == 0) {
                g = p.get();
                break;
            }
        bool opened = false;
        if(g == nullptr) {
            // batches spread over the devices the box has: a new group goes to the device that serves the fewest
            // (sources share nothing; WF_HIP_DEVICE pins everything to one device)
            const int ndev = std::max(api().device_count(), 1);
            std::vector<int> load((size_t)ndev, 0);
            for(auto &p : r.groups)
                if(p->device >= 0 && p->device < ndev)
                    ++load[(size_t)p->device];
            int dev = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            if(const char *e = std::getenv("WF_HIP_DEVICE"))
                dev = std::min(std::max(std::atoi(e), 0), ndev - 1);
            auto fresh = std::make_unique<WFHipGroup>();
            if(!fresh->create(c, dev)) {
                if(has_display(c))
                    continue; // once more without the display
                LogWarn << "HIP spectrum path unavailable for this configuration (" << api().last_error(nullptr) << "); using the CPU path";
                return false;
            }
            g = fresh.get();
            r.groups.push_back(std::move(fresh));
            opened = true;
        }
        uint32_t slot = 0;
        while(g->member[slot] != nullptr)
            ++slot;
        if(api().reset(g->h, slot, 1) != WF_HIP_OK) { // the stream starts as update() leaves a source
            if(opened) // nobody joined: the group would stay in the registry with no member for ever
                r.groups.erase(std::remove_if(r.groups.begin(), r.groups.end(), [g](const auto &p) { return p.get() == g; }), r.groups.end());
            return false;
        }
        // reset has put the device's view of the slot back to SHOWN with no per-stream RMS: the cache of what the device holds
        // must say so, or a flush before this member's first submit would skip re-sending PAUSED and the device would tick the
        // fresh stream on its zero ring (m_last_silent = 1 where the reference still has false)
        g->state_dev[slot] = WF_HIP_SHOWN;
        g->rms_dev[slot] = -1.0f;
        g->member[slot] = this;
        m_hip_joined = g->batch; // rows read back for earlier batches belong to whoever held the slot then
        g->submitted[slot] = 0;
        g->state[slot] = WF_HIP_PAUSED;
        ++g->members;
        m_group = g;
        m_slot = slot;
        m_hip_window.assign((size_t)m_capture_channels * m_fft_size, 0.0f);
        m_hip_prev.assign((size_t)m_capture_channels * m_fft_size, 0.0f);
        m_hip_display = g->display;
        if(m_hip_display) {
            m_hip_points = g->points;
            m_hip_per_row = g->per_row;
            m_hip_bars.assign((size_t)g->disp_ch * g->points, 0.0f);
            m_hip_verts.assign((size_t)g->disp_ch * g->per_row * 4, 0.0f);
            m_hip_vcounts.assign(g->per_row ? g->disp_ch : 0u, 0u);
            m_hip_pre.assign(g->disp_ch, 0.0f);
        }
        return true;
    }
    // level meter: join (or open) the batch of this meter configuration.  Meter buffers beyond 65536 samples (meter_buf above
    // ~1.3 s; the reference allows 600 s) stay synchronous: a member's first hand-over is the whole buffer.
    const char *mb = std::getenv("WF_HIP_BATCHED_METER"); // 0: the meter stays synchronous while the spectrum batches
    const char *wb = std::getenv("WF_HIP_BATCHED_WAVE");  // 0: so does the waveform display
    // waveform display: the same kind of batch (a group's configuration has either cfg.meter or cfg.waveform set, so the two never
    // share one).  With volume normalisation a source stays synchronous: its m_input_rms is computed on the host per call.
    const bool wave_batch = c.waveform && !c.normalize_volume && !(wb && wb[0] == '0');
    if(((c.meter && m_fft_size <= 65536 && !(mb && mb[0] == '0')) || wave_batch) && batched_mode()) {
        auto &r = registry();
        std::lock_guard lock(r.mtx);
        WFHipMeterGroup *g = nullptr;
        for(auto &p : r.mgroups)
            if(!p->failed && p->members < p->capacity && std::memcmp(&p->cfg, &c, sizeof(c)) == 0) {
                g = p.get();
                break;
            }
        bool opened = false;
        if(g == nullptr) {
            const int ndev = std::max(api().device_count(), 1);
            std::vector<int> load((size_t)ndev, 0);
            for(auto &p : r.groups)
                if(p->device >= 0 && p->device < ndev)
                    ++load[(size_t)p->device];
            for(auto &p : r.mgroups)
                if(p->device >= 0 && p->device < ndev)
                    ++load[(size_t)p->device];
            int dev = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            if(const char *e = std::getenv("WF_HIP_DEVICE"))
                dev = std::min(std::max(std::atoi(e), 0), ndev - 1);
            auto fresh = std::make_unique<WFHipMeterGroup>();
            if(!fresh->create(c, dev)) {
                LogWarn << "HIP meter path unavailable for this configuration (" << api().last_error(nullptr) << "); using the CPU path";
                return false;
            }
            g = fresh.get();
            r.mgroups.push_back(std::move(fresh));
            opened = true;
        }
        uint32_t slot = 0;
        while(g->member[slot] != nullptr)
            ++slot;
        if(api().reset(g->h, slot, 1) != WF_HIP_OK) {
            if(opened)
                r.mgroups.erase(std::remove_if(r.mgroups.begin(), r.mgroups.end(), [g](const auto &p) { return p.get() == g; }), r.mgroups.end());
            return false;
        }
        g->state_dev[slot] = WF_HIP_SHOWN; // (what reset leaves on the device: the next flush re-sends the slot's mask)
        g->member[slot] = this;
        m_hip_joined = g->batch;
        g->submitted[slot] = 0;
        g->state[slot] = WF_HIP_PAUSED;
        ++g->members;
        m_mgroup = g;
        m_slot = slot;
        if(g->wave) {
            g->delay[slot] = 0;
            g->audio_ts[slot] = 0;
            m_hip_pushed = m_fft_size; // update() pre-fills m_fft_size zeros; so does the device (wf_hip_reset)
            m_hip_prev.assign((size_t)m_capture_channels * m_hip_pushed, 0.0f);
        }
        return true;
    }
    int dev = 0;
    if(const char *e = std::getenv("WF_HIP_DEVICE"))
        dev = std::min(std::max(std::atoi(e), 0), std::max(api().device_count(), 1) - 1);
    const int rc = api().create(&c, dev, 1, 0, &m_hip);
    if(rc != WF_HIP_OK) {
        m_hip = nullptr;
        if(has_display(c))
            continue; // once more without the display
        // e.g. WF_HIP_ERR_UNSUPPORTED for a configuration the device refuses
        LogWarn << "HIP spectrum path unavailable for this configuration (" << api().last_error(nullptr) << "); using the CPU path";
        return false;
    }
    m_hip_display = has_display(c) && api().num_bars(m_hip) > 0 && (c.curve != 0 || api().num_vertices(m_hip) > 0);
    if(m_hip_display) {
        const uint32_t dch = api().display_channels(m_hip);
        m_hip_points = api().num_bars(m_hip);
        m_hip_per_row = api().num_vertices(m_hip);
        m_hip_bars.assign((size_t)dch * m_hip_points, 0.0f);
        m_hip_verts.assign((size_t)dch * m_hip_per_row * 4, 0.0f);
        m_hip_vcounts.assign(m_hip_per_row ? dch : 0u, 0u);
        m_hip_pre.assign(dch, 0.0f);
    }
    m_hip_window.assign((size_t)m_capture_channels * m_fft_size, 0.0f);
    m_hip_out.assign((size_t)m_output_channels * (m_fft_size / 2), 0.0f);
    m_hip_hidden = false;
    m_hip_state = WF_HIP_SHOWN;
    m_hip_pushed = (m_display_mode == DisplayMode::WAVEFORM) ? m_fft_size : 0; // update() pre-fills m_fft_size zeros; so does the device
    m_hip_prev.assign((size_t)m_capture_channels * m_hip_pushed, 0.0f);          // ... and those zeros are what tick_waveform expects at the front
    return true;
    } // (attempts)
    return false;
}

// The display part of wf_config (include/wf_config.h) from the members get_settings / update() have set: which of render_bars /
// render_curve runs (src/source.cpp:1354-1357), its interpolation, filter and geometry.
void WAVSourceHIP::hip_display_config(wf_config &c) const
{
    const bool stepped = m_display_mode == DisplayMode::STEPPED_BAR;
    if(m_display_mode == DisplayMode::BAR || stepped) {
        c.bars = 1;
        c.vertices = stepped ? 3u : 1u;
    } else if(m_display_mode == DisplayMode::CURVE) {
        // the points come from the device; render_curve's vertex loop only writes their y into a strip whose x never changes
        // (src/source.cpp:1443-1460) and stays here: its vertices would be eight times the bytes of the points on the way back
        c.curve = 1;
        c.vertices = 0;
    } else
        return;
    c.filter_mode = (m_filter_mode != FilterMode::NONE) ? WF_FILTER_GAUSS : WF_FILTER_NONE;
    c.filter_radius = m_filter_radius;
    c.step_width = m_step_width;
    c.step_gap = m_step_gap;
    c.radial = m_radial ? 1u : 0u;
}

// a member's (or the synchronous handle's) display of the frame just read back -> the members render() draws from; the bar tops
// also go where the reference keeps them (m_interp_bufs after render_bars / render_curve), for whoever looks there
void WAVSourceHIP::hip_collect_display(const float *bars, const float *verts, const uint32_t *counts)
{
    std::memcpy(m_hip_bars.data(), bars, m_hip_bars.size() * sizeof(float));
    if(m_hip_per_row) {
        std::memcpy(m_hip_verts.data(), verts, m_hip_verts.size() * sizeof(float));
        std::memcpy(m_hip_vcounts.data(), counts, m_hip_vcounts.size() * sizeof(uint32_t));
    }
    hip_publish_display();
}

// m_hip_bars / m_hip_verts / m_hip_vcounts hold this frame's display: the bar tops (curve points) also go where render() and the
// reference's own members expect them
void WAVSourceHIP::hip_publish_display()
{
    const size_t channels = m_hip_points ? m_hip_bars.size() / m_hip_points : 0;
    for(size_t channel = 0; channel < channels; ++channel)
        if(m_interp_bufs[channel].size() >= m_hip_points)
            std::memcpy(m_interp_bufs[channel].data(), m_hip_bars.data() + channel * m_hip_points, (size_t)m_hip_points * sizeof(float));
    m_hip_display_valid = true;
}

void WAVSourceHIP::render([[maybe_unused]] gs_effect_t *effect)
{
    std::lock_guard lock(m_mtx);
    // the display modes the device does not draw (level meter: two values; waveform), sources on the CPU path, a frame before
    // the first device result: the reference's own render
    const bool spectrum = !m_meter_mode && m_display_mode != DisplayMode::WAVEFORM;
    // The shader's gradient height / pulse colour follow the smallest y BEFORE the mirror image replaces the upper half
    // (src/source.cpp:1548-1567, :1411-1424).  Without the Gaussian filter every output above the middle has ONE pre-mirror value
    // (init_interp clamps their positions to the highest bin, src/source.cpp:857-862), which the device keeps
    // (WF_HIP_OUT_PREMIRROR).  With the filter on (apply_filter runs before that loop, :1541-1547) outputs half+1 ... half+radius
    // blend that constant with lower bars and the minimum may sit at any of them: those two render modes then keep the
    // reference's own loops over m_decibels.
    const bool miny_on_device = !m_mirror_freq_axis || m_filter_mode == FilterMode::NONE ||
                                (m_render_mode != RenderMode::GRADIENT && m_render_mode != RenderMode::PULSE);
    if(!spectrum || !using_hip() || !m_hip_display || !m_hip_display_valid || !miny_on_device) {
        if(spectrum && using_hip())
            g_host_renders.fetch_add(1);
        WAVSourceGeneric::render(effect);
        return;
    }
    if(m_last_silent && m_hide_on_silent) // src/source.cpp:1349-1352
        return;
    if(m_vbuf == nullptr)
        return;
    g_device_renders.fetch_add(1);
    const bool curve = m_display_mode == DisplayMode::CURVE;
    auto tech = get_shader_tech();
    // the constants render_bars / render_curve hand to set_shader_vars (src/source.cpp:1368-1373, :1480-1493)
    const auto center = (float)m_height / 2;
    const auto bottom = (float)m_height;
    const auto cpos = m_stereo ? center : bottom;
    const auto channel_offset = m_channel_spacing * 0.5f;
    auto border_top = 0.0f, border_bottom = cpos - channel_offset;
    if(!curve) {
        border_top = (m_rounded_caps) ? m_cap_radius : 0.0f;
        border_bottom = (m_rounded_caps && (!m_stereo || (m_channel_spacing > 0))) ? cpos - m_cap_radius : cpos;
        if(m_channel_spacing > 0)
            border_bottom -= channel_offset;
        if(m_min_bar_height > 0)
            border_bottom -= m_min_bar_height;
        border_bottom = std::clamp(border_bottom, border_top, cpos);
    }
    const auto channels = m_stereo ? 2u : 1u;
    auto miny = cpos;
    auto minpos = 0u;
    // The shader's gradient height / pulse colour follow the smallest y of the rows BEFORE the mirror image replaces their upper
    // halves (src/source.cpp:1548-1567, :1411-1424).  The device hands back the mirrored rows -- whose upper halves repeat lower
    // values: a strict "<" never picks them -- and, per row, the one value every output above the middle had before the mirror
    // (WF_HIP_OUT_PREMIRROR): first seen at output num_bars / 2 + 1.
    const auto half = m_hip_points / 2u;
    for(auto channel = 0u; channel < channels; ++channel) {
        const auto upto = m_mirror_freq_axis ? std::min(half + 1u, m_hip_points) : m_hip_points;
        for(auto i = 0u; i < upto; ++i) {
            const auto val = m_hip_bars[(size_t)channel * m_hip_points + i];
            if(val < miny) {
                miny = val;
                minpos = i;
            }
        }
        if(m_mirror_freq_axis && half + 1u < m_hip_points && channel < m_hip_pre.size() && m_hip_pre[channel] < miny) {
            miny = m_hip_pre[channel];
            minpos = half + 1u;
        }
    }
    set_shader_vars(cpos, miny, (float)minpos, channel_offset, border_top, border_bottom);

    gs_technique_begin(tech);
    gs_technique_begin_pass(tech, 0);
    gs_load_vertexbuffer(m_vbuf);
    gs_load_indexbuffer(nullptr);
    auto vbdata = gs_vertexbuffer_get_data(m_vbuf);
    static_assert(sizeof(vec3) == 4 * sizeof(float), "libobs' vec3 is four floats: what the device's vertex fill writes");
    for(auto channel = 0u; channel < channels; ++channel) {
        if(curve) {
            // render_curve's own loop (src/source.cpp:1436-1461): the points' y into the strip
            auto offset = channel_offset;
            if(channel)
                offset = -offset;
            const auto bot = cpos - offset;
            const float *pts = m_hip_bars.data() + (size_t)channel * m_hip_points;
            const auto n = std::min<size_t>(m_hip_points, m_width);
            for(size_t i = 0; i < n; ++i) {
                const auto val = pts[i];
                if(m_render_mode == RenderMode::LINE)
                    vbdata->points[i].y = channel == 0 ? val : bottom - val;
                else {
                    vbdata->points[i * 2].y = channel == 0 ? val : bottom - val;
                    vbdata->points[(i * 2) + 1].y = bot;
                }
            }
            gs_vertexbuffer_flush(m_vbuf);
            gs_draw((m_render_mode != RenderMode::LINE) ? GS_TRISTRIP : GS_LINESTRIP, 0, (uint32_t)vbdata->num);
            continue;
        }
        const auto count = std::min<size_t>(m_hip_vcounts[channel], std::min<size_t>(m_hip_per_row, vbdata->num));
        std::memcpy(vbdata->points, m_hip_verts.data() + (size_t)channel * m_hip_per_row * 4, count * sizeof(vec3));
        gs_vertexbuffer_flush(m_vbuf);
        if(count > 0)
            gs_draw(GS_TRIS, 0, (uint32_t)count);
    }
    gs_load_vertexbuffer(nullptr);
    gs_technique_end_pass(tech);
    gs_technique_end(tech);
}

void WAVSourceHIP::update(obs_data_t *settings)
{
    std::lock_guard lock(m_mtx);
    WAVSourceGeneric::update(settings); // tables, rings, render state: unchanged reference code
    hip_configure();
}

// Same observable behaviour as WAVSourceGeneric::tick_spectrum (src/source_generic.cpp:26-180): the A/V-synchronised
// window of each channel goes to the device, the device runs the whole per-tick state machine (hidden/timeout reset,
// silence detection, window, FFT, magnitude, slope, smoothing, dBFS, normalisation, roll-off) and m_decibels /
// m_last_silent come back.
// reference :50-59: keep dtsize bytes in every channel's ring, copy the first fft_size samples of what is left -- the
// A/V-synchronised window -- into m_hip_window.  false: a channel holds less than that (the reference skips it).
bool WAVSourceHIP::hip_window(size_t &dtframes)
{
    const auto bufsz = m_fft_size * sizeof(float);
    const int64_t dtaudio = get_audio_sync(m_tick_ts);
    dtframes = (dtaudio > 0) ? size_t(ns_to_audio_frames(m_audio_info.samples_per_sec, (uint64_t)dtaudio)) : 0;
    const size_t dtsize = dtframes * sizeof(float) + bufsz;
    for(auto channel = 0u; channel < m_capture_channels; ++channel)
        if(m_capturebufs[channel].size() < dtsize)
            return false;
    for(auto channel = 0u; channel < m_capture_channels; ++channel) {
        m_capturebufs[channel].pop_front(nullptr, m_capturebufs[channel].size() - dtsize);
        m_capturebufs[channel].peek_front(m_hip_window.data() + (size_t)channel * m_fft_size, bufsz);
    }
    return true;
}

// The batched path (struct WFHipGroup).  m_decibels / m_last_silent are the device's results for the previous video frame.
void WAVSourceHIP::tick_spectrum_batched(float seconds)
{
    auto &a = api();
    auto &r = registry();
    std::unique_lock lock(r.mtx);
    WFHipGroup *g = m_group;
    const uint32_t slot = m_slot;
    const size_t N = m_fft_size, outsz = m_fft_size / 2;
    bool ok = !g->failed;
    // a member that comes round again while its last hand-over is still waiting for the rest of the frame completes it
    if(ok && g->submitted[slot] == g->batch)
        ok = g->flush();
    // 1. collect the row the last flushed batch left for this source
    const uint32_t last = (uint32_t)((g->batch - 1) & 1);
    // (whatever the stream holds after that batch: a source that sat a frame out finds the row of its last tick, unchanged)
    if(ok && g->batch > 1 && g->rows_valid[last] && g->batch - 1 >= m_hip_joined) {
        ok = a.readback_done(g->h, last) == WF_HIP_OK;
        if(ok) {
            const float *row = g->rows[last] + (size_t)slot * g->out_ch * g->M;
            for(auto channel = 0u; channel < m_output_channels; ++channel)
                std::memcpy(m_decibels[channel].get(), row + (size_t)channel * outsz, outsz * sizeof(float));
            m_last_silent = g->silent[last][slot] != 0;
            if(g->rms_feed)
                m_input_rms = g->rms_back[last][slot]; // as of that batch's tick (for observers; the device uses its own)
            if(g->display && m_hip_display && g->pre[last])
                m_hip_pre.assign(g->pre[last] + (size_t)slot * g->disp_ch, g->pre[last] + (size_t)(slot + 1) * g->disp_ch);
            if(g->display && m_hip_display)
                hip_collect_display(g->bars[last] + (size_t)slot * g->disp_ch * g->points,
                                    g->per_row ? g->verts[last] + (size_t)slot * g->disp_ch * g->per_row * 4 : nullptr,
                                    g->per_row ? g->vcounts[last] + (size_t)slot * g->disp_ch : nullptr);
        }
    }
    if(!ok) {
        lock.unlock();
        LogWarn << "HIP batch unavailable; this source continues on the CPU path";
        if(g->rms_feed && m_input_rms_buf && m_input_rms_size > 0) {
            // the device kept the one-second window of squared peaks; the host's ring was never advanced.  Seed it so that its
            // sum reproduces the last m_input_rms that came back (every entry = the mean square), instead of a window of zeros
            // that would clamp the gain to max_gain for the next second
            const float ms = m_input_rms * m_input_rms;
            for(size_t i = 0; i < m_input_rms_size; ++i)
                m_input_rms_buf[i] = ms;
            m_input_rms_pos = 0;
        }
        hip_release();
        g_fallback_ticks.fetch_add(1);
        WAVSourceGeneric::tick_spectrum(seconds);
        return;
    }
    // 2. submit this frame
    const uint32_t b = (uint32_t)(g->batch & 1);
    if(g->n_submitted == 0)
        a.ingest_done(g->h, b); // the staging block of this slot has been copied out (two frames ago)
    const auto dtcapture = m_tick_ts - m_capture_ts;
    const bool timed_out = dtcapture > CAPTURE_TIMEOUT;
    const bool hidden = !m_show || timed_out; // reference :34
    uint32_t fresh = 0;
    uint8_t st = hidden ? (timed_out ? WF_HIP_HIDDEN_TIMEOUT : WF_HIP_HIDDEN) : WF_HIP_SHOWN;
    size_t dtframes = 0;
    if(!hidden) {
        if(!hip_window(dtframes)) {
            st = WF_HIP_STARVED; // underflow (:55-61): no channel is processed, the end-of-tick pass still runs (see tick_spectrum)
        } else {
            // Which samples are new?  The device ring of this stream ends with the window handed over last time; the new
            // window overlaps it by fft_size - shift samples.  The shift the timestamps suggest is checked against the
            // samples themselves (and its neighbours tried: the timestamps round); if no overlap is found the whole window
            // goes -- always correct, since the device analyses the newest fft_size samples of its ring.
            long est = (long)N;
            if(m_hip_have_prev) {
                const int64_t dts = (int64_t)(m_audio_ts - m_hip_prev_audio_ts);
                est = (long)std::llround((double)dts * (double)m_audio_info.samples_per_sec / 1e9) - ((long)dtframes - (long)m_hip_prev_sync);
            }
            auto overlaps = [&](long s) {
                if(s < 0 || s > (long)N)
                    return false;
                for(auto channel = 0u; channel < m_capture_channels; ++channel)
                    if(std::memcmp(m_hip_prev.data() + (size_t)channel * N + (size_t)s, m_hip_window.data() + (size_t)channel * N,
                                   (N - (size_t)s) * sizeof(float)) != 0)
                        return false;
                return true;
            };
            long shift = (long)N;
            if(m_hip_have_prev) {
                for(long d : {0L, 1L, -1L, 2L, -2L})
                    if(overlaps(est + d)) {
                        shift = est + d;
                        break;
                    }
            }
            fresh = (uint32_t)shift;
            float *dst = g->stage[b] + (size_t)slot * g->cap_ch * N;
            for(auto channel = 0u; channel < m_capture_channels; ++channel)
                std::memcpy(dst + (size_t)channel * N, m_hip_window.data() + (size_t)channel * N + (N - fresh), (size_t)fresh * sizeof(float));
            m_hip_prev.swap(m_hip_window);
            m_hip_have_prev = true;
            m_hip_prev_audio_ts = m_audio_ts;
            m_hip_prev_sync = (int64_t)dtframes;
        }
    }
    g->frames[slot] = fresh;
    g->state[slot] = st;
    g->rms[slot] = m_input_rms;
    g->seconds = seconds;
    g->submitted[slot] = g->batch;
    ++g->n_submitted;
    // 3. the frame's last member sends the batch off
    if(g->n_submitted >= g->members)
        g->flush(); // (a failure shows at the members' next tick)
}

// WAVSourceGeneric::update_input_rms (src/source_generic.cpp:392-403) = sync_rms_buffer (src/source.cpp:810-835: everything
// in m_rms_sync_buf older than the A/V-sync point moves into the circular one-second m_input_rms_buf) + the sum of its
// m_input_rms_size squares.  Batched mode: the values sync_rms_buffer would move are staged for the device instead, which
// keeps the window and its (two-level) sum per stream; no 48000-float loop per source and frame on the host.  Called by
// WAVSource::tick (src/source.cpp:1330-1331) ahead of tick_spectrum, under m_mtx.
void WAVSourceHIP::update_input_rms()
{
    WFHipGroup *g = m_group;
    if(g == nullptr || !g->rms_feed) {
        g_host_rms_updates.fetch_add(1);
        WAVSourceGeneric::update_input_rms();
        return;
    }
    auto &r = registry();
    std::lock_guard lock(r.mtx);
    if(g->failed)
        return; // this source leaves the group at its tick_spectrum and continues on the host
    // a member that comes round again while its last hand-over is still waiting for the rest of the frame completes it
    if(g->submitted[m_slot] == g->batch)
        g->flush();
    const int64_t dtaudio = get_audio_sync(m_tick_ts);
    const size_t dtsize = (dtaudio > 0) ? size_t(ns_to_audio_frames(m_audio_info.samples_per_sec, (uint64_t)dtaudio)) * sizeof(float) : 0;
    if(m_rms_sync_buf.size() <= dtsize)
        return; // sync_rms_buffer returns false: m_input_rms stays
    const uint32_t b = (uint32_t)(g->batch & 1);
    if(g->n_submitted == 0 && std::all_of(g->sq_frames.begin(), g->sq_frames.end(), [](uint32_t v) { return v == 0; }))
        api().ingest_done(g->h, b); // first writer of this frame: the slot's staging has been copied out (two frames ago)
    size_t consume = (m_rms_sync_buf.size() - dtsize) / sizeof(float);
    const size_t room = g->sq_max - g->sq_frames[m_slot];
    if(consume > room) {
        // more than a window's worth since the last hand-over: only the newest values can still be inside the window
        m_rms_sync_buf.pop_front(nullptr, (consume - room) * sizeof(float));
        consume = room;
    }
    float *dst = g->sq_stage[b] + (size_t)m_slot * g->sq_max + g->sq_frames[m_slot];
    m_rms_sync_buf.pop_front(dst, consume * sizeof(float));
    g->sq_frames[m_slot] += (uint32_t)consume;
}

void WAVSourceHIP::tick_spectrum(float seconds)
{
    if(m_group != nullptr) {
        tick_spectrum_batched(seconds);
        return;
    }
    if(m_hip == nullptr) {
        g_fallback_ticks.fetch_add(1);
        WAVSourceGeneric::tick_spectrum(seconds);
        return;
    }
    auto &a = api();
    const auto outsz = m_fft_size / 2;

    const auto dtcapture = m_tick_ts - m_capture_ts;
    const bool hidden = !m_show || (dtcapture > CAPTURE_TIMEOUT); // reference :34
    bool ok = true;
    size_t dtframes = 0;
    // Underflow (fewer samples than window + A/V-sync delay, :55-61): every channel is skipped, but unless m_last_silent the
    // reference's end-of-tick pass still runs over the rows as they are -- dbfs of a stale dB value is DB_MIN, and the volume
    // normalisation gain goes on top (:138-179).  The device does the same for a stream marked WF_HIP_STARVED.
    const bool starved = !hidden && !hip_window(dtframes);
    const int state = hidden ? WF_HIP_HIDDEN : (starved ? WF_HIP_STARVED : WF_HIP_SHOWN);
    if(state != m_hip_state) {
        const uint8_t mask = (uint8_t)state;
        ok = a.set_hidden(m_hip, 0, 1, &mask) == WF_HIP_OK;
        m_hip_state = state;
    }
    if(ok && state == WF_HIP_SHOWN) // the whole window replaces the device ring's newest fft_size samples
        ok = a.push_audio(m_hip, 0, 1, m_hip_window.data(), (uint32_t)m_fft_size) == WF_HIP_OK;
    wf_hip_tick_params p{};
    p.seconds = seconds;
    p.delay_frames = 0;
    p.input_rms = m_input_rms;
    p.flags = 0;
    uint8_t silent = 0;
    if(!ok || a.tick(m_hip, &p) != WF_HIP_OK || a.read(m_hip, WF_HIP_OUT_DECIBELS, 0, 1, m_hip_out.data()) != WF_HIP_OK ||
       a.read(m_hip, WF_HIP_OUT_LAST_SILENT, 0, 1, &silent) != WF_HIP_OK) {
        LogWarn << "HIP tick failed (" << a.last_error(m_hip) << "); falling back to the CPU path";
        hip_release();
        g_fallback_ticks.fetch_add(1);
        WAVSourceGeneric::tick_spectrum(seconds);
        return;
    }
    for(auto channel = 0u; channel < m_output_channels; ++channel)
        std::memcpy(m_decibels[channel].get(), m_hip_out.data() + (size_t)channel * outsz, outsz * sizeof(float));
    m_last_silent = silent != 0;
    if(m_hip_display) {
        // straight into the members render() draws from (no per-tick allocations, no staging copy)
        if(a.read(m_hip, WF_HIP_OUT_BARS, 0, 1, m_hip_bars.data()) == WF_HIP_OK &&
           (m_hip_per_row == 0 || (a.read(m_hip, WF_HIP_OUT_VERTICES, 0, 1, m_hip_verts.data()) == WF_HIP_OK &&
                                   a.read(m_hip, WF_HIP_OUT_VERTEX_COUNTS, 0, 1, m_hip_vcounts.data()) == WF_HIP_OK)) &&
           (!m_mirror_freq_axis || a.read(m_hip, WF_HIP_OUT_PREMIRROR, 0, 1, m_hip_pre.data()) == WF_HIP_OK))
            hip_publish_display();
        else
            m_hip_display_valid = false; // render() goes back to the host loops over m_decibels
    }
}

// Same observable behaviour as WAVSourceGeneric::tick_meter (src/source_generic.cpp:182-269): the audio that tick_meter
// would pop into its meter buffer this tick (everything older than the A/V-sync point, :201-220) goes to the device ring
// instead; the device takes the level over the last m_fft_size consumed samples, smooths it, converts to dBFS and
// decides m_last_silent; m_meter_val / m_last_silent come back for render_bars (src/source.cpp:1505-1509).
// The batched meter path (struct WFHipMeterGroup): m_meter_val / m_last_silent are the device's results for the previous frame.
void WAVSourceHIP::tick_meter_batched(float seconds)
{
    auto &a = api();
    auto &r = registry();
    std::unique_lock lock(r.mtx);
    WFHipMeterGroup *g = m_mgroup;
    const uint32_t slot = m_slot;
    bool ok = !g->failed;
    if(ok && g->submitted[slot] == g->batch) // came round again while the frame was still being assembled: complete it
        ok = g->flush();
    // 1. collect the levels the last flushed batch left for this source
    const uint32_t last = (uint32_t)((g->batch - 1) & 1);
    if(ok && g->batch > 1 && g->valid[last] && g->batch - 1 >= m_hip_joined) {
        ok = a.readback_done(g->h, last) == WF_HIP_OK;
        if(ok) {
            for(auto channel = 0u; channel < m_capture_channels; ++channel)
                m_meter_val[channel] = g->levels[last][(size_t)slot * g->cap_ch + channel];
            m_last_silent = g->silent[last][slot] != 0;
        }
    }
    if(!ok) {
        lock.unlock();
        LogWarn << "HIP meter batch unavailable; this source continues on the CPU path";
        hip_release();
        g_fallback_ticks.fetch_add(1);
        WAVSourceGeneric::tick_meter(seconds);
        return;
    }
    // 2. submit this frame: state, and what tick_meter would pop into its meter buffer (:201-220)
    const auto dtcapture = m_tick_ts - m_capture_ts;
    const bool timed_out = dtcapture > CAPTURE_TIMEOUT; // :184
    uint32_t frames = 0;
    if(!timed_out) {
        const int64_t dtaudio = get_audio_sync(m_tick_ts);
        const size_t dtsize = (dtaudio > 0) ? size_t(ns_to_audio_frames(m_audio_info.samples_per_sec, (uint64_t)dtaudio)) * sizeof(float) : 0;
        size_t n_min = 0;
        for(auto channel = 0u; channel < m_capture_channels; ++channel) {
            const auto sz = m_capturebufs[channel].size();
            const size_t n = (sz > dtsize) ? (sz - dtsize) / sizeof(float) : 0;
            n_min = (channel == 0) ? n : std::min(n_min, n);
        }
        frames = (uint32_t)n_min;
        auto &dst = g->pending[slot];
        dst.resize((size_t)m_capture_channels * frames);
        for(auto channel = 0u; channel < m_capture_channels && frames; ++channel)
            m_capturebufs[channel].pop_front(dst.data() + (size_t)channel * frames, (size_t)frames * sizeof(float));
    }
    g->frames[slot] = frames;
    g->state[slot] = timed_out ? WF_HIP_HIDDEN_TIMEOUT : (m_show ? WF_HIP_SHOWN : WF_HIP_HIDDEN);
    g->seconds = seconds;
    g->submitted[slot] = g->batch;
    ++g->n_submitted;
    if(g->n_submitted >= g->members)
        g->flush();
}

void WAVSourceHIP::tick_meter(float seconds)
{
    if(m_mgroup != nullptr) {
        tick_meter_batched(seconds);
        return;
    }
    if(m_hip == nullptr) {
        g_fallback_ticks.fetch_add(1);
        WAVSourceGeneric::tick_meter(seconds);
        return;
    }
    auto &a = api();
    const auto dtcapture = m_tick_ts - m_capture_ts;
    const bool timed_out = dtcapture > CAPTURE_TIMEOUT; // :184
    const int state = timed_out ? WF_HIP_HIDDEN_TIMEOUT : (m_show ? WF_HIP_SHOWN : WF_HIP_HIDDEN);
    bool ok = true;
    if(state != m_hip_state) {
        const uint8_t mask = (uint8_t)state;
        ok = a.set_hidden(m_hip, 0, 1, &mask) == WF_HIP_OK;
        m_hip_state = state;
    }
    if(ok && !timed_out) {
        // :201-220: everything beyond dtsize bytes is consumed -- here: moved to the device ring, oldest first
        const int64_t dtaudio = get_audio_sync(m_tick_ts);
        const size_t dtsize = (dtaudio > 0) ? size_t(ns_to_audio_frames(m_audio_info.samples_per_sec, (uint64_t)dtaudio)) * sizeof(float) : 0;
        size_t frames = 0;
        for(auto channel = 0u; channel < m_capture_channels; ++channel) {
            const auto sz = m_capturebufs[channel].size();
            const size_t n = (sz > dtsize) ? (sz - dtsize) / sizeof(float) : 0;
            frames = (channel == 0) ? n : std::min(frames, n); // the channels of a source are pushed together by capture_audio
        }
        if(frames > 0) {
            m_hip_window.resize((size_t)m_capture_channels * frames);
            for(auto channel = 0u; channel < m_capture_channels; ++channel)
                m_capturebufs[channel].pop_front(m_hip_window.data() + (size_t)channel * frames, frames * sizeof(float));
            // capture_audio keeps at most dtsamples + m_fft_size samples, so this is never more than the ring holds
            ok = a.push_audio(m_hip, 0, 1, m_hip_window.data(), (uint32_t)frames) == WF_HIP_OK;
        }
    }
    wf_hip_tick_params p{};
    p.seconds = seconds;
    float levels[2] = {DB_MIN, DB_MIN};
    uint8_t silent = 0;
    if(!ok || a.tick(m_hip, &p) != WF_HIP_OK || a.read(m_hip, WF_HIP_OUT_METER, 0, 1, levels) != WF_HIP_OK ||
       a.read(m_hip, WF_HIP_OUT_LAST_SILENT, 0, 1, &silent) != WF_HIP_OK) {
        LogWarn << "HIP meter tick failed (" << a.last_error(m_hip) << "); falling back to the CPU path";
        hip_release();
        g_fallback_ticks.fetch_add(1);
        WAVSourceGeneric::tick_meter(seconds);
        return;
    }
    for(auto channel = 0u; channel < m_capture_channels; ++channel)
        m_meter_val[channel] = levels[channel];
    m_last_silent = silent != 0;
}

// The batched waveform path (struct WFHipMeterGroup with cfg.waveform): m_decibels / m_last_silent are the device's results for
// the previous video frame.  What this frame hands over is exactly what the synchronous path below pushes inside the call.
void WAVSourceHIP::tick_waveform_batched(float seconds)
{
    auto &a = api();
    auto &r = registry();
    std::unique_lock lock(r.mtx);
    WFHipMeterGroup *g = m_mgroup;
    const uint32_t slot = m_slot;
    const size_t outsz = m_fft_size;
    bool ok = !g->failed;
    if(ok && g->submitted[slot] == g->batch) // came round again while the frame was still being assembled: complete it
        ok = g->flush();
    // 1. collect the rows the last flushed batch left for this source
    const uint32_t last = (uint32_t)((g->batch - 1) & 1);
    if(ok && g->batch > 1 && g->valid[last] && g->batch - 1 >= m_hip_joined) {
        ok = a.readback_done(g->h, last) == WF_HIP_OK;
        if(ok) {
            const float *row = g->rows[last] + (size_t)slot * g->out_ch * g->width;
            for(auto channel = 0u; channel < m_output_channels; ++channel)
                std::memcpy(m_decibels[channel].get(), row + (size_t)channel * outsz, outsz * sizeof(float));
            m_last_silent = g->silent[last][slot] != 0;
        }
    }
    if(!ok) {
        lock.unlock();
        LogWarn << "HIP waveform batch unavailable; this source continues on the CPU path";
        hip_release();
        g_fallback_ticks.fetch_add(1);
        WAVSourceGeneric::tick_waveform(seconds);
        return;
    }
    // 2. submit this frame
    const auto dtcapture = m_tick_ts - m_capture_ts;
    const bool hidden = !m_show || (dtcapture > CAPTURE_TIMEOUT); // :279
    uint8_t st = hidden ? WF_HIP_HIDDEN : WF_HIP_SHOWN;
    size_t fresh = 0;
    if(!hidden) {
        const int64_t dtaudio = get_audio_sync(m_tick_ts);
        const size_t reserve = (dtaudio > 0) ? size_t(ns_to_audio_frames(m_audio_info.samples_per_sec, (uint64_t)dtaudio)) : 0; // frames
        const size_t max_frames = m_waveform_samples + reserve;
        if(max_frames > g->ring_cap) {
            // wf_hip_set_stream_delay would reject the whole batch for this one stream (an A/V-sync reserve beyond what the
            // device rings were sized for): only THIS source leaves -- nothing of it has been staged or popped yet, so the
            // reference's tick_waveform takes the frame over from m_capturebufs as they are, with m_decibels as step 1 left
            // them (the previous frame's rows) and the sweep position the device kept for it (one 8-byte read that waits for
            // the last batch: this happens once) -- and the group's other members never notice
            uint64_t wts = 0;
            if(a.read(g->h, WF_HIP_OUT_WAVEFORM_TS, slot, 1, &wts) == WF_HIP_OK)
                m_waveform_ts = (size_t)wts; // (else 0: the reference catches up by itself, src/source_generic.cpp:318-321)
            lock.unlock();
            LogWarn << "HIP waveform batch: this source's A/V-sync reserve (" << reserve << " frames) does not fit the device ring ("
                    << g->ring_cap << "); it continues on the CPU path";
            hip_release();
            g_fallback_ticks.fetch_add(1);
            WAVSourceGeneric::tick_waveform(seconds);
            return;
        }
        bool enough = true;
        for(auto i = 0u; i < m_capture_channels; ++i)
            if(m_capturebufs[i].size() <= reserve * sizeof(float)) // :293-295: the reference returns before it touches anything
                enough = false;
        if(!enough) {
            st = WF_HIP_PAUSED;
        } else {
            // (see the synchronous path below for what is pushed and why)
            size_t frames = 0;
            bool front_kept = true;
            std::vector<std::vector<float>> all(m_capture_channels);
            for(auto channel = 0u; channel < m_capture_channels; ++channel) {
                auto &buf = m_capturebufs[channel];
                if(buf.size() > max_frames * sizeof(float)) // :303-304
                    buf.pop_front(nullptr, buf.size() - max_frames * sizeof(float));
                const size_t s = buf.size() / sizeof(float);
                all[channel].resize(s);
                buf.peek_front(all[channel].data(), s * sizeof(float));
                if(channel == 0)
                    frames = s;
                front_kept = front_kept && s >= m_hip_pushed && m_hip_prev.size() == (size_t)m_capture_channels * m_hip_pushed &&
                             std::memcmp(all[channel].data(), m_hip_prev.data() + (size_t)channel * m_hip_pushed, m_hip_pushed * sizeof(float)) == 0;
            }
            fresh = front_kept ? frames - m_hip_pushed : frames;
            const auto start_ts = m_audio_ts - audio_frames_to_ns(m_audio_info.samples_per_sec, frames);
            const auto stop_ts = m_audio_ts - audio_frames_to_ns(m_audio_info.samples_per_sec, reserve);
            const bool rollover = (start_ts >= m_audio_ts) || (stop_ts > m_audio_ts); // :314-317
            const size_t stay = rollover ? frames : reserve; // frames left in m_capturebufs behind this tick
            auto &dst = g->pending[slot];
            dst.resize((size_t)m_capture_channels * fresh);
            m_hip_prev.assign((size_t)m_capture_channels * stay, 0.0f);
            for(auto channel = 0u; channel < m_capture_channels; ++channel) {
                auto &buf = m_capturebufs[channel];
                const size_t s = all[channel].size();
                std::memcpy(dst.data() + (size_t)channel * fresh, all[channel].data() + (s - fresh), fresh * sizeof(float));
                std::memcpy(m_hip_prev.data() + (size_t)channel * stay, all[channel].data() + (s - stay), stay * sizeof(float));
                buf.pop_front(nullptr, (s - stay) * sizeof(float)); // :321: only the reserve stays
            }
            m_hip_pushed = stay;
            g->delay[slot] = (uint32_t)reserve;
            g->audio_ts[slot] = m_audio_ts;
        }
    }
    g->frames[slot] = (uint32_t)fresh;
    g->state[slot] = st;
    g->seconds = seconds;
    g->submitted[slot] = g->batch;
    ++g->n_submitted;
    if(g->n_submitted >= g->members)
        g->flush();
}

// Same observable behaviour as WAVSourceGeneric::tick_waveform (src/source_generic.cpp:271-390).  The device ring mirrors
// m_capturebufs: every tick the frames captured since the last one are appended to it, the device picks the new points
// from its ring exactly where the reference picks them from its scratch copy (both count back from the newest sample),
// and m_capturebufs is popped down to the A/V-sync reserve as the reference does (:321).  m_waveform_ts lives on the device.
void WAVSourceHIP::tick_waveform(float seconds)
{
    if(m_mgroup != nullptr) {
        tick_waveform_batched(seconds);
        return;
    }
    if(m_hip == nullptr) {
        g_fallback_ticks.fetch_add(1);
        WAVSourceGeneric::tick_waveform(seconds);
        return;
    }
    auto &a = api();
    const auto outsz = m_fft_size;
    const auto dtcapture = m_tick_ts - m_capture_ts;
    const bool hidden = !m_show || (dtcapture > CAPTURE_TIMEOUT); // :279
    bool ok = true;
    if(hidden != m_hip_hidden) {
        const uint8_t mask = hidden ? 1 : 0;
        ok = a.set_hidden(m_hip, 0, 1, &mask) == WF_HIP_OK;
        m_hip_hidden = hidden;
    }
    wf_hip_tick_params p{};
    p.seconds = seconds;
    p.input_rms = m_input_rms;
    p.audio_ts_ns = m_audio_ts;
    if(ok && !hidden) {
        const int64_t dtaudio = get_audio_sync(m_tick_ts);
        const size_t reserve = (dtaudio > 0) ? size_t(ns_to_audio_frames(m_audio_info.samples_per_sec, (uint64_t)dtaudio)) : 0; // frames
        const size_t max_frames = m_waveform_samples + reserve;
        for(auto i = 0u; i < m_capture_channels; ++i)
            if(m_capturebufs[i].size() <= reserve * sizeof(float)) // :293-295
                return;
        // The device ring holds every frame pushed so far and counts "what the reference's ring holds" as (frames pushed since
        // the last tick) + (the reserve that tick left behind), :321.  Frames at the front of m_capturebufs that are already
        // there -- the last tick's reserve, m_hip_pushed of them -- must not be pushed twice; but capture_audio drops from the
        // front once a ring exceeds dtsamples + m_waveform_samples (src/source.cpp:1883-1886), and then what is left is
        // re-sent whole: after a drop the ring is at its cap, the reference trims it to max_size and so does the device
        // (avail > max_size), and the re-sent frames are contiguous.  Whether the old reserve is still at the front is
        // checked against a copy of it (all-zero audio can match by accident: then both ways of pushing give zeros).
        size_t frames = 0, fresh = 0;
        bool front_kept = true;
        std::vector<std::vector<float>> all(m_capture_channels);
        for(auto channel = 0u; channel < m_capture_channels; ++channel) {
            auto &buf = m_capturebufs[channel];
            if(buf.size() > max_frames * sizeof(float)) // :303-304
                buf.pop_front(nullptr, buf.size() - max_frames * sizeof(float));
            const size_t s = buf.size() / sizeof(float);
            all[channel].resize(s);
            buf.peek_front(all[channel].data(), s * sizeof(float));
            if(channel == 0)
                frames = s;
            front_kept = front_kept && s >= m_hip_pushed && m_hip_prev.size() == (size_t)m_capture_channels * m_hip_pushed &&
                         std::memcmp(all[channel].data(), m_hip_prev.data() + (size_t)channel * m_hip_pushed, m_hip_pushed * sizeof(float)) == 0;
        }
        fresh = front_kept ? frames - m_hip_pushed : frames;
        // "timestamp rollover, give up" (:314-317; a tick before any audio has a timestamp): the reference has trimmed its
        // ring but consumes nothing.  The device takes the same exit from the same timestamps (and records the trim); here the
        // ring keeps what it holds, all of it now on the device.
        const auto start_ts = m_audio_ts - audio_frames_to_ns(m_audio_info.samples_per_sec, frames);
        const auto stop_ts = m_audio_ts - audio_frames_to_ns(m_audio_info.samples_per_sec, reserve);
        const bool rollover = (start_ts >= m_audio_ts) || (stop_ts > m_audio_ts);
        const size_t stay = rollover ? frames : reserve; // frames left in m_capturebufs behind this tick
        m_hip_window.resize((size_t)m_capture_channels * fresh);
        m_hip_prev.assign((size_t)m_capture_channels * stay, 0.0f);
        for(auto channel = 0u; channel < m_capture_channels; ++channel) {
            auto &buf = m_capturebufs[channel];
            const size_t s = all[channel].size();
            std::memcpy(m_hip_window.data() + (size_t)channel * fresh, all[channel].data() + (s - fresh), fresh * sizeof(float));
            std::memcpy(m_hip_prev.data() + (size_t)channel * stay, all[channel].data() + (s - stay), stay * sizeof(float));
            buf.pop_front(nullptr, (s - stay) * sizeof(float)); // :321: only the reserve stays
        }
        if(fresh > 0)
            ok = a.push_audio(m_hip, 0, 1, m_hip_window.data(), (uint32_t)fresh) == WF_HIP_OK;
        m_hip_pushed = stay;
        p.delay_frames = (uint32_t)reserve;
    }
    m_hip_out.resize((size_t)m_output_channels * outsz);
    uint8_t silent = 0;
    if(!ok || a.tick(m_hip, &p) != WF_HIP_OK || a.read(m_hip, WF_HIP_OUT_DECIBELS, 0, 1, m_hip_out.data()) != WF_HIP_OK ||
       a.read(m_hip, WF_HIP_OUT_LAST_SILENT, 0, 1, &silent) != WF_HIP_OK) {
        LogWarn << "HIP waveform tick failed (" << a.last_error(m_hip) << "); falling back to the CPU path";
        hip_release();
        g_fallback_ticks.fetch_add(1);
        return; // the audio of this tick has been consumed; the CPU path takes over at the next one
    }
    for(auto channel = 0u; channel < m_output_channels; ++channel)
        std::memcpy(m_decibels[channel].get(), m_hip_out.data() + (size_t)channel * outsz, outsz * sizeof(float));
    m_last_silent = silent != 0; // (no device display here: hip_configure builds one for spectrum displays only, the waveform's points are the rows)
}
