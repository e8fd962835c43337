"""Scenario scripts shared by every parity test.

A scenario is a configuration plus a list of steps that drive ONE source the way OBS drives
the plugin (audio packets in, video ticks, show/hide).  The same script can be played on
  * RefBackend     the reference itself (oracle/_ref/libwfref.so)     -> tools/make_golden.py
  * OracleBackend  the CPU restatement (oracle/libwforacle.so)         -> CPU tests
  * HipBackend     libwaveform_hip.so through its C ABI               -> GPU tests
and every backend records the same observables after each tick:
  db    float32 [display_channels, fft_size/2]   m_decibels
  bars  float32 [display_channels, num_bars]     m_interp_bufs after render_bars / render_curve (if cfg.bars or cfg.curve;
                                                  num_bars = m_width for the curve)
  silent bool                                    m_last_silent
Level-meter configurations (cfg.meter) record instead
  db    float32 [1, capture_channels]            m_meter_val (dBFS), one level per captured channel
  bars  float32 [1, capture_channels]            m_interp_bufs[0] after render_bars (src/source.cpp:1505-1509, :1548-1557)
Waveform configurations (cfg.waveform) record
  db    float32 [display_channels, width]        m_decibels: the history of dBFS points (tick_waveform)
A scenario may carry sync_ms: the source's audio sync offset (P_AUDIO_SYNC_OFFSET), i.e. a constant A/V-sync reserve of
sync_ms * 48 frames that every backend applies.
Audio is the counter-hash noise of include/wf_synth.h (tools/synth.py), so fixtures only store
outputs.
"""
from __future__ import annotations

import numpy as np

from tools import synth

SEED = 0x5741564546524D31


def _steps(n_ticks, hop=800, seconds=1.0 / 60.0):
    out = []
    for _ in range(n_ticks):
        out += [("noise", hop), ("tick", seconds)]
    return out


# name -> dict(cfg=<wf_config overrides>, steps=[...], record=<"all" | int: last k ticks>)
SCENARIOS = {
    # BASELINE.json configs[0]: 1ch mono, FFT=1024, Hann (OBS mono layout: one captured channel)
    "cfg1_mono_1024": dict(cfg=dict(fft_size=1024, stereo=0, capture_channels=1), steps=_steps(6), record="all"),
    # configs[1]: stereo, FFT=2048, Hann + magnitude + dB, no smoothing
    "cfg2_stereo_2048_nosmooth": dict(cfg=dict(fft_size=2048, stereo=1, tsmoothing=0), steps=_steps(4), record="all"),
    # configs[2]: FFT=4096, EMA + slope
    "cfg3_stereo_4096_ema_slope": dict(cfg=dict(fft_size=4096, stereo=1, slope=1.0), steps=_steps(8), record=3),
    # configs[3]: FFT=16384, gravity (TV-EMA) smoothing + Lanczos bars
    "cfg4_16384_tv_lanczos_bars": dict(cfg=dict(fft_size=16384, stereo=1, tsmoothing=2, bars=1, interp_mode=1),
                                       steps=_steps(4), record=1),
    # configs[4] per-stream shape: FFT=4096 EMA + slope + bars (the all-gathered quantity)
    "cfg5_4096_bars": dict(cfg=dict(fft_size=4096, stereo=1, slope=1.0, bars=1, interp_mode=1), steps=_steps(6), record=2),
    # mono mixdown of two captured channels after smoothing, TV-EMA, fast peaks (Appendix C.5)
    "mono_mix_4096_tv_fastpeaks": dict(cfg=dict(fft_size=4096, stereo=0, tsmoothing=2, fast_peaks=1, slope=0.5),
                                       steps=_steps(6), record=2),
    "fft8192_stereo": dict(cfg=dict(fft_size=8192, stereo=1, slope=2.0), steps=_steps(4), record=1),
    # every window function
    "win_hamming": dict(cfg=dict(fft_size=2048, stereo=0, capture_channels=1, window=2), steps=_steps(3), record=1),
    "win_blackman": dict(cfg=dict(fft_size=2048, stereo=0, capture_channels=1, window=3), steps=_steps(3), record=1),
    "win_blackman_harris": dict(cfg=dict(fft_size=2048, stereo=0, capture_channels=1, window=4), steps=_steps(3), record=1),
    "win_sine3": dict(cfg=dict(fft_size=2048, stereo=0, capture_channels=1, window=5, sine_exponent=3), steps=_steps(3), record=1),
    "win_none": dict(cfg=dict(fft_size=2048, stereo=0, capture_channels=1, window=0), steps=_steps(3), record=1),
    # roll-off + catmull-rom bars on a linear axis, single captured channel shown as stereo (channel dup)
    "rolloff_catrom_linear": dict(cfg=dict(fft_size=2048, stereo=1, capture_channels=1, rolloff_q=1.5, rolloff_rate=12.0, bars=1,
                                           interp_mode=2, log_scale=0), steps=_steps(4), record=2),
    "point_bars_mirror": dict(cfg=dict(fft_size=4096, stereo=0, bars=1, interp_mode=0, mirror_freq_axis=1, bar_width=10, bar_gap=2),
                              steps=_steps(4), record=1),
    # many narrow bars: more bars than threads per spectrum and more products than one LDS scratch holds (chunked reduction)
    "many_bars_1024": dict(cfg=dict(fft_size=1024, stereo=1, bars=1, interp_mode=1, width=1920, bar_width=1, bar_gap=0, log_scale=0),
                           steps=_steps(3), record=1),
    "many_bars_4096_catrom": dict(cfg=dict(fft_size=4096, stereo=0, bars=1, interp_mode=2, width=600, bar_width=2, bar_gap=1),
                                  steps=_steps(3), record=1),
    # curve display (render_curve): one interpolated point per pixel column; Gaussian filter across the points / bars
    "curve_4096_lanczos_gauss": dict(cfg=dict(fft_size=4096, stereo=1, slope=1.0, curve=1, interp_mode=1, filter_mode=1, filter_radius=1.5),
                                     steps=_steps(4), record=1),
    "curve_2048_catrom_mirror": dict(cfg=dict(fft_size=2048, stereo=0, curve=1, interp_mode=2, mirror_freq_axis=1, width=600),
                                     steps=_steps(3), record=1),
    "curve_1024_point_linear_gauss": dict(cfg=dict(fft_size=1024, stereo=1, capture_channels=1, curve=1, interp_mode=0, log_scale=0,
                                                   filter_mode=1, filter_radius=4.0, channel_spacing=8), steps=_steps(3), record=1),
    # the plugin's default configuration (src/source.cpp:119-174): mono mixdown of two channels, 800-point Catmull-Rom curve
    "plugin_defaults_4096": dict(cfg=dict(fft_size=4096, stereo=0, curve=1, interp_mode=2), steps=_steps(5), record=2),
    # curves wider than a thread's registers hold (more than 8 points per thread at two wavefronts): points streamed; with the
    # Gaussian filter they are staged behind the row
    "curve_4096_catrom_wide_gauss": dict(cfg=dict(fft_size=4096, stereo=1, slope=1.0, curve=1, interp_mode=2, width=1900, filter_mode=1,
                                                  filter_radius=1.5), steps=_steps(4), record=1),
    "curve_2048_lanczos_wide_mirror": dict(cfg=dict(fft_size=2048, stereo=1, curve=1, interp_mode=1, width=2560, mirror_freq_axis=1),
                                           steps=_steps(3), record=1),
    "curve_1024_catrom_3840_linear": dict(cfg=dict(fft_size=1024, stereo=0, capture_channels=1, curve=1, interp_mode=2, width=3840, log_scale=0),
                                          steps=_steps(3), record=1),
    # ---- vertex fill (cfg.vertices): what render_bars / render_curve hand to gs_draw, per displayed channel
    "verts_bars_4096_stereo_caps": dict(cfg=dict(fft_size=4096, stereo=1, slope=1.0, bars=1, interp_mode=1, rounded_caps=1, channel_spacing=6, vertices=1),
                                        steps=_steps(4), record=1),
    "verts_bars_2048_mono_plain": dict(cfg=dict(fft_size=2048, stereo=0, bars=1, interp_mode=2, bar_width=10, bar_gap=3, min_bar_height=3, vertices=1),
                                       steps=_steps(3), record=1),
    "verts_bars_1024_mono_caps_mirror": dict(cfg=dict(fft_size=1024, stereo=0, capture_channels=1, bars=1, interp_mode=0, rounded_caps=1, bar_width=12,
                                                      bar_gap=2, mirror_freq_axis=1, vertices=1), steps=_steps(3), record=1),
    # radial layout: the cap fans are full circles (the polar transform itself is the plugin's shader)
    "verts_bars_2048_stereo_caps_radial": dict(cfg=dict(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=1, rounded_caps=1, channel_spacing=6,
                                                        height=180, radial=1, vertices=1), steps=_steps(3), record=1),
    "verts_bars_1024_mono_caps_radial": dict(cfg=dict(fft_size=1024, stereo=0, bars=1, interp_mode=2, rounded_caps=1, bar_width=14, bar_gap=4,
                                                      height=150, radial=1, vertices=1), steps=_steps(3), record=1),
    "verts_curve_2048_stereo_solid": dict(cfg=dict(fft_size=2048, stereo=1, curve=1, interp_mode=2, channel_spacing=8, width=640, vertices=1),
                                          steps=_steps(3), record=1),
    "verts_curve_1024_line": dict(cfg=dict(fft_size=1024, stereo=0, curve=1, interp_mode=1, width=500, filter_mode=1, filter_radius=1.5, vertices=2),
                                  steps=_steps(3), record=1),
    # stepped bars: the number of vertices follows the signal; narrow steps, stereo with spacing / mono
    "verts_stepped_2048_stereo": dict(cfg=dict(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=1, channel_spacing=6, vertices=3),
                                      steps=_steps(4), record=2),
    "verts_stepped_1024_mono_fine": dict(cfg=dict(fft_size=1024, stereo=0, bars=1, interp_mode=2, bar_width=6, bar_gap=1, step_width=3, step_gap=1,
                                                  min_bar_height=2, vertices=3), steps=_steps(3) + [("silence", 1200), ("tick",)], record=2),
    # the reference's whole slider range (width <= 3840, filter radius <= 32, src/source.cpp:287, :409) at small fft sizes: the
    # points + the filter's staging do not fit the tick kernel's exchange buffer; big_outputs_kernel derives the display from
    # the stored rows (formerly WF_HIP_ERR_UNSUPPORTED)
    "curve_1024_3840_gauss32": dict(cfg=dict(fft_size=1024, stereo=1, slope=1.0, curve=1, interp_mode=2, width=3840, filter_mode=1, filter_radius=32.0),
                                    steps=_steps(4), record=2),
    "bars_512_1px_gauss12_mono": dict(cfg=dict(fft_size=512, stereo=0, bars=1, interp_mode=1, width=3840, bar_width=1, bar_gap=0, filter_mode=1,
                                               filter_radius=12.5, log_scale=0), steps=_steps(3) + [("silence", 1400), ("tick",), ("noise", 800), ("tick",)],
                                      record=3),
    "curve_2048_lanczos_2560_gauss20_hide": dict(cfg=dict(fft_size=2048, stereo=1, capture_channels=1, curve=1, interp_mode=1, width=2560, filter_mode=1,
                                                          filter_radius=20.0, mirror_freq_axis=1),
                                                 steps=_steps(3) + [("hide",), ("noise", 800), ("tick",), ("show",)] + _steps(2), record=3),
    "bars_gauss_4096": dict(cfg=dict(fft_size=4096, stereo=1, bars=1, interp_mode=1, filter_mode=1, filter_radius=0.8), steps=_steps(4), record=1),
    # ragged packets: 441-frame hops (window start not 16-byte aligned), then a 1024 packet
    "ragged_hops": dict(cfg=dict(fft_size=2048, stereo=1),
                        steps=[("noise", 441), ("tick",), ("noise", 441), ("tick",), ("noise", 1024), ("tick",), ("noise", 3), ("tick",),
                               ("tick",)], record="all"),
    # silence state machine (src/source_generic.cpp:63-95,138-139): noise, then digital silence until the
    # display decays below floor-10 and the source goes silent, then noise again
    "silence_cycle": dict(cfg=dict(fft_size=1024, stereo=1, gravity=0.2),
                          steps=_steps(3) + [("silence", 1200), ("tick",)] + [("silence", 800), ("tick",)] * 14 + _steps(2), record="all"),
    "silence_cycle_mono": dict(cfg=dict(fft_size=1024, stereo=0, gravity=0.2),
                               steps=_steps(2) + [("silence", 1200), ("tick",)] + [("silence", 800), ("tick",)] * 14 + _steps(2), record="all"),
    # one channel silent, the other live (Appendix C.3 quirk: the skipped channel is re-dBFS'ed)
    "half_silent_stereo": dict(cfg=dict(fft_size=1024, stereo=1, gravity=0.2),
                               steps=_steps(2) + [("noise_ch0_only", 1200), ("tick",)] + [("noise_ch0_only", 800), ("tick",)] * 12, record="all"),
    # hidden / capture timeout branch (src/source_generic.cpp:34-48)
    "hide_show": dict(cfg=dict(fft_size=1024, stereo=1), steps=_steps(3) + [("hide",), ("noise", 800), ("tick",), ("noise", 800), ("tick",),
                                                                           ("show",)] + _steps(3), record="all"),
    # muted packets are pushed as zeros (src/source.cpp:1879-1880)
    "muted_packets": dict(cfg=dict(fft_size=1024, stereo=1, tsmoothing=0), steps=_steps(2) + [("mute", 800), ("tick",), ("noise", 800), ("tick",)],
                          record="all"),
    # ---- FFT sizes below the smallest geometry run zero-padded on the 1024-point one (512, 256, 128 = the slider's minimum)
    "small_512_stereo_bars": dict(cfg=dict(fft_size=512, stereo=1, slope=1.0, bars=1, interp_mode=1), steps=_steps(6), record=2),
    "small_256_mono_mix_tv": dict(cfg=dict(fft_size=256, stereo=0, tsmoothing=2, fast_peaks=1, window=3, rolloff_q=1.5, rolloff_rate=12.0),
                                  steps=[("noise", 441), ("tick",)] * 4 + [("noise", 1024), ("tick",), ("noise", 3), ("tick",)], record=3),
    "small_128_single_dup_curve": dict(cfg=dict(fft_size=128, stereo=1, capture_channels=1, curve=1, interp_mode=2, width=300, tsmoothing=0),
                                       steps=_steps(4), record=2),
    "small_512_silence_cycle": dict(cfg=dict(fft_size=512, stereo=1, gravity=0.2),
                                    steps=_steps(3) + [("silence", 800), ("tick",)] * 15 + _steps(2) + [("hide",), ("noise", 800), ("tick",), ("show",)]
                                    + _steps(2), record="all"),
    # ---- the largest geometry runs the channels of a stereo stream in different workgroups (split mode): everything that
    # couples the channels, at N = 16384
    "split_16384_silence_cycle": dict(cfg=dict(fft_size=16384, stereo=1, gravity=0.2),
                                      steps=_steps(2) + [("silence", 17000), ("tick",)] + [("silence", 800), ("tick",)] * 14 + _steps(2), record=2),
    "split_16384_half_silent": dict(cfg=dict(fft_size=16384, stereo=1, gravity=0.2, bars=1, interp_mode=1),
                                    steps=_steps(2) + [("noise_ch0_only", 17000), ("tick",)] + [("noise_ch0_only", 800), ("tick",)] * 12
                                    + [("silence", 17000), ("tick",)] + [("silence", 800), ("tick",)] * 13 + [("noise_ch1_only", 800), ("tick",)] * 2,
                                    record=3),
    "split_8192_half_silent": dict(cfg=dict(fft_size=8192, stereo=1, gravity=0.2, curve=1, interp_mode=1, filter_mode=1, filter_radius=2.0),
                                   steps=_steps(2) + [("noise_ch1_only", 9000), ("tick",)] + [("noise_ch1_only", 800), ("tick",)] * 10
                                   + [("silence", 9000), ("tick",)] + [("silence", 800), ("tick",)] * 12 + [("noise_ch0_only", 800), ("tick",)] * 2
                                   + [("hide",), ("noise", 800), ("tick",), ("show",)] + _steps(2), record=3),
    "split_16384_hide_timeout": dict(cfg=dict(fft_size=16384, stereo=1, slope=1.0),
                                     steps=_steps(2) + [("hide",), ("noise", 800), ("tick",), ("tick",), ("show",)] + _steps(2)
                                     + [("timeout",), ("tick",), ("tick",)] + _steps(2), record=2),
    # ---- FFT sizes that are not powers of two (every other position of the reference's slider, and its "auto" size 800):
    # Bluestein over the complex FFT core
    "any_800_mono_mix_bars": dict(cfg=dict(fft_size=800, stereo=0, slope=1.0, bars=1, interp_mode=1), steps=_steps(6), record=2),
    "any_1536_stereo_curve": dict(cfg=dict(fft_size=1536, stereo=1, tsmoothing=2, fast_peaks=1, curve=1, interp_mode=2, width=600, window=3),
                                  steps=[("noise", 441), ("tick",)] * 4 + [("noise", 1024), ("tick",), ("noise", 3), ("tick",)], record=2),
    "any_4160_stereo_silence": dict(cfg=dict(fft_size=4160, stereo=1, gravity=0.2, rolloff_q=1.5, rolloff_rate=12.0),
                                    steps=_steps(3) + [("noise_ch0_only", 4400), ("tick",)] + [("noise_ch0_only", 800), ("tick",)] * 5
                                    + [("silence", 4400), ("tick",)] + [("silence", 800), ("tick",)] * 12 + _steps(2)
                                    + [("hide",), ("noise", 800), ("tick",), ("show",)] + _steps(2), record=3),
    "any_8000_single_nosmooth": dict(cfg=dict(fft_size=8000, stereo=0, capture_channels=1, tsmoothing=0, window=1),
                                     steps=_steps(3), record=1),
    # ---- 32768 points ("large FFT"): one spectrum per workgroup, both radix-32 passes shared by thread pairs; stereo runs split
    "large_32768_stereo_bars": dict(cfg=dict(fft_size=32768, stereo=1, slope=1.0, bars=1, interp_mode=1), steps=_steps(4), record=1),
    "large_32768_single_tv": dict(cfg=dict(fft_size=32768, stereo=0, capture_channels=1, tsmoothing=2, fast_peaks=1, window=4),
                                  steps=[("noise", 441), ("tick",)] * 3 + [("noise", 1024), ("tick",)], record=1),
    "large_32768_half_silent": dict(cfg=dict(fft_size=32768, stereo=1, gravity=0.2),
                                    steps=_steps(2) + [("noise_ch0_only", 33000), ("tick",)] + [("noise_ch0_only", 800), ("tick",)] * 6
                                    + [("silence", 33000), ("tick",)] + [("silence", 800), ("tick",)] * 10 + [("noise_ch1_only", 800), ("tick",)] * 2,
                                    record=2),
    # ---- transforms beyond a CU's LDS (wf_big.hpp): 65536 samples = 2 x 16384 complex points through device memory, and the
    # Bluestein sizes above 10912 (L = 32768 / 65536 / 131072 complex points); rows of up to 32768 bins finished in two parts
    "huge_65536_stereo_bars": dict(cfg=dict(fft_size=65536, stereo=1, slope=1.0, bars=1, interp_mode=1), steps=_steps(3), record=1),
    "huge_65536_mono_mix_tv_curve": dict(cfg=dict(fft_size=65536, stereo=0, tsmoothing=2, fast_peaks=1, window=4, curve=1, interp_mode=2,
                                                  filter_mode=1, filter_radius=1.5),
                                         steps=[("noise", 441), ("tick",)] * 3 + [("noise", 1024), ("tick",)], record=2),
    "huge_65536_half_silent": dict(cfg=dict(fft_size=65536, stereo=1, gravity=0.2, rolloff_q=1.5, rolloff_rate=12.0),
                                   steps=_steps(2) + [("noise_ch0_only", 66000), ("tick",)] + [("noise_ch0_only", 800), ("tick",)] * 6
                                   + [("silence", 66000), ("tick",)] + [("silence", 800), ("tick",)] * 10 + [("noise_ch1_only", 800), ("tick",)] * 2
                                   + [("hide",), ("noise", 800), ("tick",), ("show",)] + _steps(2), record=3),
    "any_16400_stereo_curve": dict(cfg=dict(fft_size=16400, stereo=1, slope=0.5, curve=1, interp_mode=1), steps=_steps(3), record=1),
    "any_30000_mono_mix_bars": dict(cfg=dict(fft_size=30000, stereo=0, tsmoothing=2, bars=1, interp_mode=2, filter_mode=1, filter_radius=0.8),
                                    steps=_steps(3), record=2),
    "any_48000_single_dup": dict(cfg=dict(fft_size=48000, stereo=1, capture_channels=1, window=3, gravity=0.3),
                                 steps=_steps(3) + [("silence", 49000), ("tick",)] + [("silence", 800), ("tick",)] * 8 + _steps(2), record=2),
    # ---- volume normalisation with its producer (capture_audio's RMS part + update_input_rms, src/source.cpp:1842-1871,
    # :810-835, src/source_generic.cpp:392-403): every backend derives m_input_rms from the audio itself; records add
    #   rms  float32 scalar  m_input_rms after the tick
    "normalize_4096_stereo": dict(cfg=dict(fft_size=4096, stereo=1, normalize_volume=1),
                                  steps=[("noise_amp", 800, 0.05), ("tick",)] * 10, record=2),
    # muted packets still feed the RMS (not the rings); ragged hops; mono mixdown; gain limited by max_gain at first
    "normalize_mono_muted_ragged": dict(cfg=dict(fft_size=1024, stereo=0, normalize_volume=1, volume_target=-12.0, max_gain=20.0),
                                        steps=[("noise_amp", 441, 0.02), ("tick",)] * 4 + [("mute_noise", 441), ("tick",)] * 3
                                        + [("noise", 1024), ("tick",), ("noise", 3), ("tick",), ("tick",)], record="all"),
    # more than one second of audio: the one-second RMS window slides (m_input_rms_buf wraps), loud then quiet.  Packets stay
    # <= AUDIO_OUTPUT_FRAMES (1024) as OBS delivers them: capture_audio's RMS loop re-reads the start of a longer packet
    # for its second chunk (src/source.cpp:1848-1861 never advances `data`), which no OBS packet can trigger.
    "normalize_long_1024": dict(cfg=dict(fft_size=1024, stereo=1, normalize_volume=1, tsmoothing=0),
                                steps=[("noise", 1000), ("noise", 600), ("tick",)] * 25
                                + [("noise_amp", 1000, 0.1), ("noise_amp", 600, 0.1), ("tick",)] * 20, record=2),
    # ticks that find fewer samples than window + A/V-sync delay (:55-61): every channel is skipped, but the end-of-tick pass
    # still runs over the rows as they are -- with volume normalisation they read DB_MIN + gain until the first window is in
    "underflow_normalize_sync_2048": dict(cfg=dict(fft_size=2048, stereo=1, normalize_volume=1, volume_target=-12, max_gain=21), sync_ms=20,
                                          steps=[("noise_amp", 441, 0.3), ("tick",), ("noise_amp", 37, 0.3), ("tick",), ("noise_amp", 441, 0.3), ("tick",)]
                                          + [("noise_amp", 800, 0.3), ("tick",)] * 6, record="all"),
    # ---- waveform display (tick_waveform, src/source_generic.cpp:271-390) ------------------------------------------------
    "wave_stereo_800": dict(cfg=dict(waveform=1, stereo=1), steps=_steps(10), record=3),
    # mono mixdown (row 1 keeps raw samples), ragged packets, shorter history
    "wave_mono_mix_ragged": dict(cfg=dict(waveform=1, stereo=0, width=500, meter_ms=100),
                                 steps=[("noise", 441), ("tick",)] * 5 + [("noise", 1024), ("tick",), ("noise", 37), ("tick",), ("tick",)]
                                 + [("noise_amp", 800, 0.05), ("tick",)] * 3, record=4),
    # one captured channel shown as two rows; ticks without new audio, then a burst longer than the history
    "wave_single_dup_stall": dict(cfg=dict(waveform=1, stereo=1, capture_channels=1, width=640),
                                  steps=_steps(3) + [("tick",)] * 3 + [("noise", 1024)] * 9 + [("tick",)] + _steps(2), record=3),
    # hide / show / capture timeout, with a 10 ms audio sync offset (A/V-sync reserve of 480 frames)
    # volume normalisation in waveform mode, m_input_rms from its producer (quiet audio: the gain is not clipped by max_gain)
    "wave_normalize": dict(cfg=dict(waveform=1, stereo=1, width=400, normalize_volume=1),
                           steps=[("noise_amp", 800, 0.05), ("tick",)] * 6 + [("mute_noise", 800), ("tick",)] * 2 + [("noise_amp", 800, 0.5), ("tick",)] * 3,
                           record=3),
    # an A/V-sync reserve (20 ms = 960 frames) together with stalls and a burst longer than the history: the reference's ring
    # is at its cap, dropped from the front by capture_audio and trimmed again by the tick (:303-304) while a reserve stays
    "wave_sync_burst": dict(cfg=dict(waveform=1, stereo=1, width=480), sync_ms=20,
                            steps=_steps(4) + [("tick",)] * 2 + [("noise", 1024)] * 9 + [("tick",)] + _steps(2) + [("noise", 1024)] * 3 + [("tick",)]
                            + _steps(2), record="all"),
    # the first tick comes before any audio has a timestamp (the reference trims its ring, then gives up: "timestamp
    # rollover", :303-317); small packets, 20 ms of history over 1024 points (fewer samples than points), one captured channel
    # shown as two rows, a 5 ms reserve -- fuzz seed 7060, found by an extended sweep at the end of round 2
    "wave_tick_before_audio_sync": dict(cfg=dict(waveform=1, capture_channels=1, stereo=1, width=1024, meter_ms=20), sync_ms=5,
                                        steps=[("tick",), ("noise_amp", 37, 1.0), ("tick",), ("noise_amp", 37, 0.2), ("noise_amp", 37, 1.0), ("tick",),
                                               ("tick",), ("tick",), ("noise_amp", 441, 0.2), ("tick",), ("noise_amp", 441, 0.2),
                                               ("noise_amp", 800, 0.2), ("tick",), ("hide",), ("noise", 800), ("tick",), ("tick",), ("show",),
                                               ("noise", 800), ("tick",)], record="all"),
    "wave_hide_timeout_sync": dict(cfg=dict(waveform=1, stereo=1, width=333), sync_ms=10,
                                   steps=_steps(4) + [("hide",), ("noise", 800), ("tick",), ("tick",), ("show",)] + _steps(3)
                                   + [("timeout",), ("tick",), ("tick",)] + _steps(3), record="all"),
    # ---- level meter (tick_meter, src/source_generic.cpp:182-269) -------------------------------------------------------
    # defaults: RMS over 150 ms (7200 samples), EMA g = 0.65, two captured channels; m_meter_buf starts at DB_MIN (quirk)
    "meter_rms_stereo": dict(cfg=dict(meter=1), steps=_steps(30), record="all"),
    # peak mode, one captured channel, TV-EMA + fast peaks, 100 ms buffer, quieter audio in the middle (decay)
    "meter_peak_mono_tv_fastpeaks": dict(cfg=dict(meter=1, meter_rms=0, capture_channels=1, tsmoothing=2, fast_peaks=1, meter_ms=100),
                                         steps=_steps(6) + [("noise_amp", 800, 0.05), ("tick",)] * 8 + _steps(3), record="all"),
    # no smoothing, ragged 441-frame packets (window start not 16-byte aligned), a packet longer than the meter buffer
    "meter_nosmooth_ragged": dict(cfg=dict(meter=1, tsmoothing=0, meter_ms=50),
                                  steps=[("noise", 441), ("tick",)] * 5 + [("noise", 5000), ("tick",), ("noise", 3), ("tick",), ("tick",)],
                                  record="all"),
    # noise, then digital silence until both levels fall below floor-10 (m_last_silent), then noise again
    "meter_silence_cycle": dict(cfg=dict(meter=1, gravity=0.2, meter_ms=50),
                                steps=_steps(4) + [("silence", 800), ("tick",)] * 16 + _steps(3), record="all"),
    # one channel live, the other silent: never m_last_silent
    "meter_half_silent": dict(cfg=dict(meter=1, gravity=0.2, meter_ms=50),
                              steps=_steps(3) + [("noise_ch0_only", 800), ("tick",)] * 12, record="all"),
    # hide (audio still consumed, state reset) / show; capture timeout (meter buffer cleared) / resume; rounded caps geometry
    # (channel_mode stereo keeps m_channel_spacing through get_settings, src/source.cpp:579; update() then clears m_stereo)
    "meter_hide_show_timeout": dict(cfg=dict(meter=1, stereo=1, rounded_caps=1, channel_spacing=6, min_bar_height=3),
                                    steps=_steps(4) + [("hide",), ("noise", 800), ("tick",), ("noise", 800), ("tick",), ("show",)] + _steps(3)
                                    + [("timeout",), ("tick",), ("tick",)] + _steps(4), record="all"),
    # spectrum with an audio sync offset: the window ends sync_ms before the newest sample (dtaudio > 0,
    # src/source_generic.cpp:50-59, src/source.hpp:279-285); the first ticks underflow (fewer samples than window + reserve)
    "sync_spectrum_2048": dict(cfg=dict(fft_size=2048, stereo=1, slope=1.0), sync_ms=25,
                               steps=[("noise", 441), ("tick",)] * 6 + _steps(4) + [("timeout",), ("tick",)] + _steps(3), record="all"),
    "sync_spectrum_4096_normalize_mono": dict(cfg=dict(fft_size=4096, stereo=0, normalize_volume=1, volume_target=-12.0, max_gain=20.0), sync_ms=10,
                                              steps=[("noise_amp", 800, 0.05), ("tick",)] * 8 + [("hide",), ("noise", 800), ("tick",), ("show",)]
                                              + _steps(3), record=4),
    # spectrum: capture timeout takes the same reset branch as hide (src/source_generic.cpp:34)
    "timeout_spectrum": dict(cfg=dict(fft_size=1024, stereo=1), steps=_steps(3) + [("timeout",), ("tick",), ("tick",)] + _steps(3), record="all"),
}


def make_config(overrides: dict):
    """wf_config ctypes struct from defaults + overrides (works without a GPU)."""
    import waveform_amd as wf
    return wf.Config.defaults(**overrides)


class _Feeder:
    """turns ('noise', n) etc. into sample blocks; the stream index in the hash is always 0"""

    def __init__(self, channels):
        self.channels = channels
        self.pos = 0

    def block(self, kind, frames, amp=1.0):
        a = synth.block(SEED, 0, 1, 2, self.pos, frames)[0]
        self.pos += frames
        if kind == "silence" or kind == "mute":
            a[:] = 0.0
        elif kind == "noise_ch0_only":
            a[1] = 0.0
        elif kind == "noise_ch1_only":
            a[0] = 0.0
        elif kind == "noise_amp":
            a *= np.float32(amp)
        # "mute_noise": a muted packet that carries samples
        return a[: self.channels]


def sync_reserve_frames(audio_ts_ns: int, sync_ns: int, tick_ts_ns: int, sample_rate: int = 48000) -> int:
    """What a host hands over as the A/V-sync delay of a tick or packet: get_audio_sync(ts) = m_audio_ts + m_ts_offset - ts
    (src/source.hpp:279-285) in frames, ns_to_audio_frames' integer arithmetic; 0 when the audio is not ahead of the video."""
    if audio_ts_ns == 0:
        return 0  # no packet yet (get_audio_sync: m_audio_ts == 0 -> 0)
    dt = audio_ts_ns + sync_ns - tick_ts_ns
    return (dt * sample_rate) // 1_000_000_000 if dt > 0 else 0


def no_vertex_buffer(cfg) -> bool:
    """stepped bars whose step is taller than the channel: create_vbuf computes max_steps == 0 (src/source.cpp:988-1000), logs
    "Tried to allocate vbuf of size: 0" and leaves m_vbuf null -- render() then returns before render_bars (:1351), so the
    reference computes no bars at all.  Every backend records bars=None for such a configuration."""
    if getattr(cfg, "vertices", 0) != 3 or not cfg.bars:
        return False
    stride = int(cfg.step_width) + int(cfg.step_gap)
    cpos = np.float32(cfg.height) / np.float32(2) if cfg.stereo else np.float32(cfg.height)
    off = np.float32(cfg.channel_spacing) * np.float32(0.5)
    steps = int((cpos - off) / np.float32(stride))
    if (int(cpos) - steps * stride - int(off)) > int(cfg.step_width):
        steps += 1
    return steps == 0


def play(backend, scenario: dict):
    """returns list of per-tick records: dict(db=..., bars=... | None, silent=bool)"""
    feeder = _Feeder(backend.capture_channels)
    records = []
    if scenario.get("sync_ms"):
        backend.set_sync_ms(int(scenario["sync_ms"]))
    for step in scenario["steps"]:
        op = step[0]
        if op in ("noise", "silence", "noise_ch0_only", "noise_ch1_only"):
            backend.push(feeder.block(op, step[1]), muted=False)
        elif op == "noise_amp":
            backend.push(feeder.block(op, step[1], step[2]), muted=False)
        elif op == "timeout":
            backend.timeout()  # no packet for more than CAPTURE_TIMEOUT (500 ms); the next packet ends it
        elif op in ("mute", "mute_noise"):
            backend.push(feeder.block(op, step[1]), muted=True)
        elif op == "tick":
            seconds = step[1] if len(step) > 1 else 1.0 / 60.0
            backend.tick(seconds)
            records.append(backend.observe())
        elif op == "hide":
            backend.set_hidden(True)
        elif op == "show":
            backend.set_hidden(False)
        else:
            raise ValueError(op)
    return records


def recorded(records, record):
    if record == "all":
        return list(enumerate(records))
    return list(enumerate(records))[-int(record):]


# ---- backends --------------------------------------------------------------------------------------
class RefBackend:
    def __init__(self, cfg, isa="generic", extra_settings=None):
        from helpers import ref_settings
        from oracle import wfref
        self.cfg = cfg
        # update() stamps m_capture_ts with the clock (src/source.cpp:1242): the source is created "now", so that a tick
        # before the first packet is not a capture timeout of the harness's making
        wfref.lib().wfref_set_clock_ns(1_000_000_000)
        self.sr = int(cfg.sample_rate)
        self.src = wfref.RefSource(ref_settings(cfg, **(extra_settings or {})), isa=isa, sample_rate=self.sr, channels=int(cfg.capture_channels))
        assert self.src.capture_channels == cfg.capture_channels
        self.capture_channels = int(cfg.capture_channels)
        self.disp = 2 if cfg.stereo else 1
        self.now = 1_000_000_000
        assert self.src.meter_mode == bool(cfg.meter)

    def push(self, audio, muted):
        # OBS delivers packets of at most AUDIO_OUTPUT_FRAMES (1024) frames; capture_audio's RMS loop re-reads the start of a
        # longer packet for its second chunk (src/source.cpp:1848-1861 never advances `data`), which no OBS packet can
        # trigger -- a longer script packet reaches the reference as OBS would have cut it
        if audio.shape[1] > 1024:
            for i in range(0, audio.shape[1], 1024):
                self.push(audio[:, i:i + 1024], muted)
            return
        import ctypes as C
        n = audio.shape[1]
        self.now += n * 1_000_000_000 // self.sr + 1
        L = self.src.L
        L.wfref_set_clock_ns(self.now)
        a = np.ascontiguousarray(audio, np.float32)
        fp = C.POINTER(C.c_float)
        p0 = a[0].ctypes.data_as(fp)
        p1 = a[1].ctypes.data_as(fp) if a.shape[0] > 1 else fp()
        # end-of-audio timestamp == now  ->  get_audio_sync() == 0 at the following tick
        length = n * 1_000_000_000 // self.sr
        L.wfref_push_audio(self.src.h, p0, p1, n, self.now - length, 1 if muted else 0)

    def tick(self, seconds):
        self.src.L.wfref_set_clock_ns(self.now)
        self.src.L.wfref_tick(self.src.h, seconds)

    def timeout(self):
        self.now += 600_000_000  # > CAPTURE_TIMEOUT (500 ms) since the last packet

    def set_sync_ms(self, ms):
        self.src.L.wfref_set_clock_ns(self.now)
        self.src.update(dict(audio_sync_offset=ms))  # re-runs update(): only valid before the first packet

    def set_hidden(self, hidden):
        self.src.show(not hidden)

    def observe(self):
        if self.cfg.meter:
            levels = np.array([[self.src.meter_val(c) for c in range(self.capture_channels)]], np.float32)
            self.src.render()
            return dict(db=levels, bars=self.src.bars(0)[None, : self.capture_channels], silent=self.src.last_silent)
        db = np.stack([self.src.decibels(c) for c in range(self.disp)])
        bars = None
        if (self.cfg.bars or self.cfg.curve) and not self.cfg.waveform:
            self.src.render()
            rows = [self.src.bars(c) for c in range(self.disp)]
            # a display narrower than one bar has m_num_bars == 0: render_bars draws nothing (src/source.cpp:988-1000 logs
            # "Tried to allocate vbuf of size: 0"); every backend then records no bars
            bars = None if any(r is None or len(r) == 0 for r in rows) else np.stack(rows)
            if no_vertex_buffer(self.cfg):
                assert not self.src.draws(), "a draw call without a vertex buffer?"
                bars = None
        rec = dict(db=db, bars=bars, silent=self.src.last_silent)
        if self.cfg.vertices and bars is not None:
            # per displayed channel one flush (mode -1) and then a gs_draw with the vertex buffer as it was at that call --
            # unless the channel has no vertices (stepped bars at zero height): then nothing is drawn
            verts = []
            for mode, v in self.src.draws():
                if mode < 0:
                    verts.append(np.zeros((0, 4), np.float32))
                else:
                    verts[-1] = v
            assert len(verts) == self.disp, f"{len(verts)} channels flushed for {self.disp} displayed channels"
            rec["verts"] = verts  # [n, 4] each; n varies with the signal for stepped bars
        if self.cfg.normalize_volume:
            rec["rms"] = np.float32(self.src.input_rms)
        return rec


class OracleBackend:
    """input_rms=None: m_input_rms comes from the restated producer (update_input_rms before every tick, as
    WAVSource::tick does); a number: the host's value, fixed"""

    def __init__(self, cfg, input_rms=None, exact=False):
        from oracle import restate
        self.cfg = cfg
        self.hidden = False
        self.auto_rms = bool(cfg.normalize_volume) and input_rms is None
        self.now = 1_000_000_000  # the same clock model as RefBackend: packets end "now", ticks happen "now"
        self.sync_ns = 0
        self.audio_ts = 0
        if cfg.meter:
            self.src = restate.OracleMeter(cfg)
            if exact:  # level meter only: sums of squares in double (tests/helpers.py::assert_levels_close)
                self.src.set_exact(True)
        elif cfg.waveform:
            self.src = restate.OracleWave(cfg)
            self.src.set_input_rms(input_rms or 0.0)
        else:
            self.src = restate.OracleSource(cfg)
            self.src.set_input_rms(input_rms or 0.0)
        self.capture_channels = self.src.capture_channels

    def set_sync_ms(self, ms):
        self.sync_ns = ms * 1_000_000

    def _sync(self):
        """the A/V-sync reserve as of `now`, handed to the restatement the way a host computes it"""
        reserve = sync_reserve_frames(self.audio_ts, self.sync_ns, self.now, int(self.cfg.sample_rate))
        if self.cfg.waveform:
            self.src.set_time(self.audio_ts, reserve)
        else:
            self.src.set_sync_delay(reserve)

    def _state(self, timed_out=False):
        # the restatement takes the tick's gate as given: 0 shown, 1 !m_show, 2 capture timed out
        if self.cfg.meter:
            self.src.set_state(2 if timed_out else (1 if self.hidden else 0))
        else:
            self.src.set_hidden(self.hidden or timed_out)

    def push(self, audio, muted):
        self._state()  # a packet ends a capture timeout
        self.now += audio.shape[1] * 1_000_000_000 // int(self.cfg.sample_rate) + 1
        self.audio_ts = self.now  # m_audio_ts = end of this packet
        self._sync()
        self.src.push_audio(audio, muted=muted)

    def tick(self, seconds):
        self._sync()
        if self.cfg.waveform:
            if self.auto_rms:
                self.rms = self.src.update_input_rms()
            self.src.tick()
            return
        if self.auto_rms:
            self.rms = self.src.update_input_rms()
        self.src.tick(seconds)

    def timeout(self):
        self.now += 600_000_000  # as RefBackend: no packet for 600 ms
        self._state(timed_out=True)

    def set_hidden(self, hidden):
        self.hidden = hidden
        self._state()

    def observe(self):
        if self.cfg.meter:
            return dict(db=self.src.levels()[None], bars=self.src.bars()[None], silent=self.src.last_silent)
        if self.cfg.waveform:
            rec = dict(db=self.src.rows(), bars=None, silent=self.src.last_silent, wts=self.src.waveform_ts)
            if self.auto_rms:
                rec["rms"] = np.float32(self.rms)
            return rec
        bars = None
        if self.cfg.bars or self.cfg.curve:
            self.src.render_bars()
            bars = self.src.bars()
            if bars is None or bars.shape[-1] == 0 or no_vertex_buffer(self.cfg):
                bars = None
        rec = dict(db=self.src.decibels(), bars=bars, silent=self.src.last_silent)
        if self.cfg.vertices and bars is not None:
            rec["verts"] = [self.src.vertices(c, line=self.cfg.vertices == 2) for c in range(bars.shape[0])]
        if self.auto_rms:
            rec["rms"] = np.float32(self.rms)
        return rec


class HipBackend:
    """`streams` identical copies of the scenario run in one batch (they must all agree);
    observe() returns stream `probe`.  input_rms=None: the device producer (wf_hip_enable_input_rms) for
    normalize_volume configurations; a number: the host's m_input_rms, fixed."""

    def __init__(self, cfg, streams=3, probe=1, input_rms=None):
        import waveform_amd as wf
        self.cfg = cfg
        self.auto_rms = bool(cfg.normalize_volume) and not cfg.meter and input_rms is None
        self.input_rms = input_rms or 0.0  # m_input_rms (the host's update_input_rms), for normalize_volume configurations
        self.batch = wf.SpectrumBatch(cfg, streams)
        if self.auto_rms:
            self.batch.enable_input_rms()
        self.capture_channels = self.batch.capture_channels
        self.streams = streams
        self.probe = probe
        self.disp = self.batch.display_channels
        self.hidden = False
        self.now = 1_000_000_000  # the same clock model as RefBackend: packets end "now", ticks happen "now"
        self.audio_ts = 0         # m_audio_ts: 0 until the first packet (release_audio_capture, src/source.cpp:747)
        self.sync_ns = 0

    def set_sync_ms(self, ms):
        self.sync_ns = ms * 1_000_000

    def _state(self, timed_out=False):
        self.batch.set_hidden(np.full(self.streams, 2 if timed_out else (1 if self.hidden else 0), np.uint8))

    def timeout(self):
        self.now += 600_000_000
        self._state(timed_out=True)
        self.timed_out = True

    def push(self, audio, muted):
        self.now += audio.shape[1] * 1_000_000_000 // int(self.cfg.sample_rate) + 1
        self.audio_ts = self.now
        if getattr(self, "timed_out", False):
            self.timed_out = False
            self._state()  # a packet ends a capture timeout
        if muted and self.auto_rms:
            self.batch.push_audio_muted(np.broadcast_to(audio[None], (self.streams,) + audio.shape))
        elif muted:
            self.batch.push_silence(audio.shape[1])
        else:
            self.batch.push_audio(np.broadcast_to(audio[None], (self.streams,) + audio.shape))

    def tick(self, seconds):
        reserve = sync_reserve_frames(self.audio_ts, self.sync_ns, self.now, int(self.cfg.sample_rate))  # what WAVSourceHIP derives from get_audio_sync
        self.batch.tick(seconds=seconds, input_rms=self.input_rms, delay_frames=reserve, audio_ts_ns=self.audio_ts)

    def set_hidden(self, hidden):
        self.hidden = hidden
        self._state()

    def observe(self):
        if self.cfg.meter:
            lv, bars, silent = self.batch.meter(), self.batch.bars(), self.batch.last_silent()
            assert all(np.array_equal(lv[0], lv[i]) for i in range(1, self.streams)), "streams of one batch disagree"
            return dict(db=lv[self.probe][None], bars=bars[self.probe], silent=bool(silent[self.probe]))
        db = self.batch.decibels()
        if self.cfg.waveform:
            assert all(np.array_equal(db[0], db[i]) for i in range(1, self.streams)), "streams of one batch disagree"
            rec = dict(db=db[self.probe][: self.disp], bars=None, silent=bool(self.batch.last_silent()[self.probe]))
            wts = self.batch.waveform_ts()
            assert (wts == wts[0]).all(), "streams of one batch disagree"
            rec["wts"] = int(wts[self.probe])
            if self.auto_rms:
                rec["rms"] = self.batch.input_rms()[self.probe]
            return rec
        bars = self.batch.bars() if ((self.cfg.bars or self.cfg.curve) and self.batch.num_bars > 0 and not no_vertex_buffer(self.cfg)) else None
        silent = self.batch.last_silent()
        # every copy of the scenario must produce the same bits
        assert all(np.array_equal(db[0], db[i]) for i in range(1, self.streams)), "streams of one batch disagree"
        rec = dict(db=db[self.probe][: self.disp], bars=None if bars is None else bars[self.probe], silent=bool(silent[self.probe]))
        if self.cfg.vertices and bars is not None:
            v, n = self.batch.vertices()[self.probe], self.batch.vertex_counts()[self.probe]
            rec["verts"] = [v[c, : int(n[c])] for c in range(v.shape[0])]  # what the draw call of that channel uses
        if self.auto_rms:
            rec["rms"] = self.batch.input_rms()[self.probe]
        return rec

    def close(self):
        import os
        if os.environ.get("WF_HIP_CANARY"):  # the guard bytes behind every device block are compared in wf_hip_sync: a kernel that wrote past a buffer fails the case here
            self.batch.sync()
        self.batch.close()
