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
#include <cstring>
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
    decltype(&wf_hip_read_decibels) read_decibels = nullptr;
    decltype(&wf_hip_read_last_silent) read_last_silent = nullptr;
    decltype(&wf_hip_read_meter) read_meter = nullptr;
    decltype(&wf_hip_last_error) last_error = nullptr;
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
        WF_SYM(read_decibels)
        WF_SYM(read_last_silent)
        WF_SYM(read_meter)
        WF_SYM(last_error)
#undef WF_SYM
        // struct wf_config and the entry points above must be the ones this file was compiled against
        auto abi = reinterpret_cast<decltype(&wf_hip_abi_version)>(dlsym(a.lib, "wf_hip_abi_version"));
        if(abi == nullptr || abi() != WF_HIP_ABI_VERSION)
            return;
        a.ok = true;
    });
    return a;
}

} // namespace

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
    c.bars = 0; // render_bars keeps running on the host from m_decibels in drop-in mode
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
    const int rc = api().create(&c, 0, 1, 0, &m_hip);
    if(rc != WF_HIP_OK) {
        // e.g. WF_HIP_ERR_UNSUPPORTED for an FFT size outside the implemented set (powers of two 128..32768, other multiples of 16 up to 10912)
        LogWarn << "HIP spectrum path unavailable for this configuration (" << api().last_error(nullptr) << "); using the CPU path";
        m_hip = nullptr;
        return false;
    }
    m_hip_window.assign((size_t)m_capture_channels * m_fft_size, 0.0f);
    m_hip_out.assign((size_t)m_output_channels * (m_fft_size / 2), 0.0f);
    m_hip_hidden = false;
    m_hip_state = WF_HIP_SHOWN;
    m_hip_pushed = (m_display_mode == DisplayMode::WAVEFORM) ? m_fft_size : 0; // update() pre-fills m_fft_size zeros; so does the device
    return true;
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
void WAVSourceHIP::tick_spectrum(float seconds)
{
    if(m_hip == nullptr) {
        WAVSourceGeneric::tick_spectrum(seconds);
        return;
    }
    auto &a = api();
    const auto bufsz = m_fft_size * sizeof(float);
    const auto outsz = m_fft_size / 2;

    const auto dtcapture = m_tick_ts - m_capture_ts;
    const bool hidden = !m_show || (dtcapture > CAPTURE_TIMEOUT); // reference :34
    if(hidden != m_hip_hidden) {
        const uint8_t mask = hidden ? 1 : 0;
        a.set_hidden(m_hip, 0, 1, &mask);
        m_hip_hidden = hidden;
    }
    if(!hidden) {
        // reference :50-59: keep dtsize bytes, look at the first fft_size samples of what is left
        const int64_t dtaudio = get_audio_sync(m_tick_ts);
        const size_t dtsize = ((dtaudio > 0) ? size_t(ns_to_audio_frames(m_audio_info.samples_per_sec, (uint64_t)dtaudio)) * sizeof(float) : 0) + bufsz;
        for(auto channel = 0u; channel < m_capture_channels; ++channel) {
            if(m_capturebufs[channel].size() < dtsize) {
                // underflow: the reference leaves this channel untouched; without a full window there is nothing to send
                WAVSourceGeneric::tick_spectrum(seconds);
                return;
            }
            m_capturebufs[channel].pop_front(nullptr, m_capturebufs[channel].size() - dtsize);
            m_capturebufs[channel].peek_front(m_hip_window.data() + (size_t)channel * m_fft_size, bufsz);
        }
        // the whole window replaces the device ring's newest fft_size samples
        if(a.push_audio(m_hip, 0, 1, m_hip_window.data(), (uint32_t)m_fft_size) != WF_HIP_OK) {
            WAVSourceGeneric::tick_spectrum(seconds);
            return;
        }
    }
    wf_hip_tick_params p{};
    p.seconds = seconds;
    p.delay_frames = 0;
    p.input_rms = m_input_rms;
    p.flags = 0;
    uint8_t silent = 0;
    if(a.tick(m_hip, &p) != WF_HIP_OK || a.read_decibels(m_hip, 0, 1, m_hip_out.data()) != WF_HIP_OK ||
       a.read_last_silent(m_hip, 0, 1, &silent) != WF_HIP_OK) {
        LogWarn << "HIP tick failed (" << a.last_error(m_hip) << "); falling back to the CPU path";
        hip_release();
        WAVSourceGeneric::tick_spectrum(seconds);
        return;
    }
    for(auto channel = 0u; channel < m_output_channels; ++channel)
        std::memcpy(m_decibels[channel].get(), m_hip_out.data() + (size_t)channel * outsz, outsz * sizeof(float));
    m_last_silent = silent != 0;
}

// Same observable behaviour as WAVSourceGeneric::tick_meter (src/source_generic.cpp:182-269): the audio that tick_meter
// would pop into its meter buffer this tick (everything older than the A/V-sync point, :201-220) goes to the device ring
// instead; the device takes the level over the last m_fft_size consumed samples, smooths it, converts to dBFS and
// decides m_last_silent; m_meter_val / m_last_silent come back for render_bars (src/source.cpp:1505-1509).
void WAVSourceHIP::tick_meter(float seconds)
{
    if(m_hip == nullptr) {
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
    if(!ok || a.tick(m_hip, &p) != WF_HIP_OK || a.read_meter(m_hip, 0, 1, levels) != WF_HIP_OK ||
       a.read_last_silent(m_hip, 0, 1, &silent) != WF_HIP_OK) {
        LogWarn << "HIP meter tick failed (" << a.last_error(m_hip) << "); falling back to the CPU path";
        hip_release();
        WAVSourceGeneric::tick_meter(seconds);
        return;
    }
    for(auto channel = 0u; channel < m_capture_channels; ++channel)
        m_meter_val[channel] = levels[channel];
    m_last_silent = silent != 0;
}

// Same observable behaviour as WAVSourceGeneric::tick_waveform (src/source_generic.cpp:271-390).  The device ring mirrors
// m_capturebufs: every tick the frames captured since the last one are appended to it, the device picks the new points
// from its ring exactly where the reference picks them from its scratch copy (both count back from the newest sample),
// and m_capturebufs is popped down to the A/V-sync reserve as the reference does (:321).  m_waveform_ts lives on the device.
void WAVSourceHIP::tick_waveform(float seconds)
{
    if(m_hip == nullptr) {
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
        size_t frames = 0, fresh = 0;
        for(auto channel = 0u; channel < m_capture_channels; ++channel) {
            auto &buf = m_capturebufs[channel];
            if(buf.size() > max_frames * sizeof(float)) // :303-304
                buf.pop_front(nullptr, buf.size() - max_frames * sizeof(float));
            const size_t s = buf.size() / sizeof(float);
            // capture_audio drops from the front only once the ring exceeds dtsamples + m_waveform_samples; below that
            // nothing was dropped and the first m_hip_pushed frames are on the device already.  Above it the whole buffer
            // is re-sent: the device only looks at the newest m_waveform_samples + reserve frames, which are then contiguous.
            const size_t n = (s < m_waveform_samples && s >= m_hip_pushed) ? s - m_hip_pushed : s;
            if(channel == 0) {
                frames = s;
                fresh = n;
                m_hip_window.resize((size_t)m_capture_channels * fresh);
            }
            std::vector<float> all(s);
            buf.peek_front(all.data(), s * sizeof(float));
            std::memcpy(m_hip_window.data() + (size_t)channel * fresh, all.data() + (s - fresh), fresh * sizeof(float));
            buf.pop_front(nullptr, (s - reserve) * sizeof(float)); // :321: only the reserve stays
        }
        (void)frames;
        if(fresh > 0)
            ok = a.push_audio(m_hip, 0, 1, m_hip_window.data(), (uint32_t)fresh) == WF_HIP_OK;
        m_hip_pushed = reserve;
        p.delay_frames = (uint32_t)reserve;
    }
    m_hip_out.resize((size_t)m_output_channels * outsz);
    uint8_t silent = 0;
    if(!ok || a.tick(m_hip, &p) != WF_HIP_OK || a.read_decibels(m_hip, 0, 1, m_hip_out.data()) != WF_HIP_OK ||
       a.read_last_silent(m_hip, 0, 1, &silent) != WF_HIP_OK) {
        LogWarn << "HIP waveform tick failed (" << a.last_error(m_hip) << "); falling back to the CPU path";
        hip_release();
        return; // the audio of this tick has been consumed; the CPU path takes over at the next one
    }
    for(auto channel = 0u; channel < m_output_channels; ++channel)
        std::memcpy(m_decibels[channel].get(), m_hip_out.data() + (size_t)channel * outsz, outsz * sizeof(float));
    m_last_silent = silent != 0;
}
