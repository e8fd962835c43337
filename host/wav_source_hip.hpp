/*
 * wav_source_hip.hpp -- the reference-side binding: a fourth WAVSource implementation that runs
 * tick_spectrum on an MI355X through the C ABI of libwaveform_hip.so (include/wf_hip.h).
 *
 * This file is written against the REFERENCE's own headers (phandasm/waveform src/source.hpp); it is what a
 * maintainer adds to the plugin tree.  It is not part of libwaveform_hip.so.  In this repository
 * it is compiled only into the oracle harness (oracle/ref, `isa = "hip"`), so that the drop-in can be tested
 * end to end: the reference's update() / capture_audio() / tick() / render() run verbatim and only the
 * per-tick DSP virtual is replaced.
 *
 * Interfaces replaced: WAVSource::update_input_rms() (src/source.hpp:273; WAVSourceGeneric src/source_generic.cpp:392-403),
 * WAVSource::tick_waveform(float) (src/source.hpp:277; WAVSourceGeneric src/source_generic.cpp:271-390),
 * WAVSource::tick_meter(float) (src/source.hpp:276; WAVSourceGeneric src/source_generic.cpp:182-269,
 * WAVSourceAVX src/source_avx.cpp:202-322) and WAVSource::tick_spectrum(float) (pure virtual, src/source.hpp:275), implemented in the
 * reference by WAVSourceGeneric (src/source_generic.cpp:26-180), WAVSourceAVX (src/source_avx.cpp:29-200) and
 * WAVSourceAVX2 (src/source_avx2.cpp:24-209); selected in callbacks::create (src/source.cpp:87-102).
 */
#pragma once
#include "source.hpp"   // the reference's
#include "wf_hip.h"

class WAVSourceHIP : public WAVSourceGeneric
{
protected:
    // update_input_rms (src/source.hpp:273; WAVSourceGeneric src/source_generic.cpp:392-403): in batched mode with volume
    // normalisation the squared peaks sync_rms_buffer would consume go to the device, which keeps the one-second window
    // and its sum per stream (wf_hip_enable_input_rms with feed = 1); otherwise the reference's host loop runs
    void update_input_rms() override;
    void tick_spectrum(float seconds) override;
    void tick_meter(float seconds) override;   // level meter: src/source_generic.cpp:182-269 on the device
    void tick_waveform(float seconds) override; // waveform display: src/source_generic.cpp:271-390 on the device

    // Two ways to the device (see wav_source_hip.cpp):
    //  * batched (the default for the spectrum display): sources that share a configuration share one handle, as streams of
    //    one batch; every video frame each source hands over the audio its capture buffers gained, the frame's last source
    //    enqueues ONE tick for all of them and every source picks up its row one frame later (m_group / m_slot);
    //    The level meter batches the same way (m_mgroup): one ragged ingest of what every source's tick_meter consumes, one
    //    meter_tick_kernel per video frame, levels one frame later;
    //    and so does the waveform display (its group hands over every member's A/V-sync reserve and audio timestamp too; sources
    //    with volume normalisation stay synchronous);
    //  * synchronous (WF_HIP_BATCHED=0): a handle of one stream per source,
    //    push -> tick -> read inside the call, zero latency (m_hip).
    struct WFHipGroup *m_group = nullptr;
    struct WFHipMeterGroup *m_mgroup = nullptr; // level meter, batched: the sources of one meter configuration share a handle too
    uint32_t m_slot = 0;
    uint64_t m_hip_joined = 0;         // the group's batch counter when this source took its slot
    bool m_hip_have_prev = false;      // m_hip_prev holds the window handed over at the previous tick
    std::vector<float> m_hip_prev;     // [capture_channels][fft_size]
    uint64_t m_hip_prev_audio_ts = 0;  // m_audio_ts / A/V-sync frames of that window: where the next one is expected to start
    int64_t m_hip_prev_sync = 0;
    wf_hip *m_hip = nullptr;
    bool m_hip_hidden = false;
    std::vector<float> m_hip_window;   // [capture_channels][fft_size] staging for the H2D copy
    std::vector<float> m_hip_out;      // [output_channels][fft_size/2]
    size_t m_hip_pushed = 0;           // waveform: frames at the front of m_capturebufs that are already in the device ring
    int m_hip_state = 0;               // what the device was last told: WF_HIP_SHOWN / WF_HIP_HIDDEN / WF_HIP_HIDDEN_TIMEOUT

    // The display from the device (render() override): what render_bars / render_curve would compute from m_decibels -- bar tops
    // or curve points in pixels, and the vertices they write into the vertex buffer -- comes back with the rows (one frame late
    // in batched mode); render() only hands it to libobs.  m_hip_display: the handle was created with cfg.bars / curve / vertices.
    bool m_hip_display = false;
    bool m_hip_display_valid = false;  // a device frame has been collected since the last update()
    uint32_t m_hip_points = 0;         // outputs per displayed row (m_num_bars or m_width)
    uint32_t m_hip_per_row = 0;        // vertices per displayed row (wf_hip_num_vertices)
    std::vector<float> m_hip_bars;     // [display channels][m_hip_points] pixel y
    std::vector<float> m_hip_verts;    // [display channels][m_hip_per_row][4]
    std::vector<uint32_t> m_hip_vcounts; // [display channels] vertices of each channel's draw call
    std::vector<float> m_hip_pre;        // [display channels] mirrored axis: the value the outputs above the middle had before the mirror

    void hip_release();
    bool hip_configure();              // (re)creates m_hip from the members update() has just set
    void hip_display_config(struct wf_config &c) const; // the display fields of wf_config from the members update() has set
    void hip_collect_display(const float *bars, const float *verts, const uint32_t *counts);
    void hip_publish_display();

public:
    using WAVSourceGeneric::WAVSourceGeneric;
    ~WAVSourceHIP() override;

    void update(obs_data_t *settings) override;
    // WAVSource::render (src/source.cpp:1346-1358 -> render_bars :1473-1670 / render_curve :1360-1470) with the interpolation, the
    // Gaussian filter, the dB -> pixel mapping, the mirror and the vertex loops taken from the device's results: what is left
    // here is what needs libobs (shader parameters, the vertex buffer, gs_draw).  Falls back to the reference's render() where
    // the device does not produce the display (level meter, waveform, a failed group, no frame collected yet).
    void render(gs_effect_t *effect) override;

    // true when a gfx950 device and libwaveform_hip.so are available (callbacks::create asks this first)
    static bool available();
    bool using_hip() const { return m_hip != nullptr || m_group != nullptr || m_mgroup != nullptr; }
    // ticks a HIP-configured source had to hand to the CPU class (underflow excepted: the reference skips those too)
    static uint64_t fallback_ticks();
    // update_input_rms calls served by the reference's host loop (0 for batched sources: the device keeps the RMS window)
    static uint64_t host_rms_updates();
    // render() calls of HIP-configured spectrum sources: drawn from the device's vertices / handed to the reference's render()
    // (whose apply_interp_filter*, apply_filter* and vertex loops then ran on the host)
    static uint64_t device_renders();
    static uint64_t host_renders();

private:
    void tick_spectrum_batched(float seconds);
    void tick_meter_batched(float seconds);
    void tick_waveform_batched(float seconds);
    bool hip_window(size_t &dtframes);  // the A/V-synchronised window of every channel -> m_hip_window; false on underflow
    friend struct WFHipGroup;
    friend struct WFHipMeterGroup;
};
