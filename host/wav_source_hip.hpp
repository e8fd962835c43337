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
 * Interfaces replaced: WAVSource::tick_waveform(float) (src/source.hpp:277; WAVSourceGeneric src/source_generic.cpp:271-390),
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
    void tick_spectrum(float seconds) override;
    void tick_meter(float seconds) override;   // level meter: src/source_generic.cpp:182-269 on the device
    void tick_waveform(float seconds) override; // waveform display: src/source_generic.cpp:271-390 on the device

    // one batch of 1 stream: a plugin that hosts several sources would share one handle per configuration
    wf_hip *m_hip = nullptr;
    bool m_hip_hidden = false;
    std::vector<float> m_hip_window;   // [capture_channels][fft_size] staging for the H2D copy
    std::vector<float> m_hip_out;      // [output_channels][fft_size/2]
    size_t m_hip_pushed = 0;           // waveform: frames at the front of m_capturebufs that are already in the device ring
    int m_hip_state = 0;               // what the device was last told: WF_HIP_SHOWN / WF_HIP_HIDDEN / WF_HIP_HIDDEN_TIMEOUT

    void hip_release();
    bool hip_configure();              // (re)creates m_hip from the members update() has just set

public:
    using WAVSourceGeneric::WAVSourceGeneric;
    ~WAVSourceHIP() override;

    void update(obs_data_t *settings) override;

    // true when a gfx950 device and libwaveform_hip.so are available (callbacks::create asks this first)
    static bool available();
    bool using_hip() const { return m_hip != nullptr; }
};
