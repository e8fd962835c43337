/*
 * wf_config.h -- plain-C description of one WAVSource configuration.
 *
 * These are the WAVSource members the spectrum hot path reads
 * (reference src/source.hpp:101-247), with the values WAVSource::get_settings
 * leaves in them (src/source.cpp:501-674; defaults src/source.cpp:119-174).
 * A host that embeds the library (the OBS plugin) fills one
 * of these in WAVSource::update() right after get_settings().
 */
#ifndef WF_CONFIG_H
#define WF_CONFIG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum FFTWindow, src/source.hpp:33-41 */
typedef enum wf_window {
    WF_WINDOW_NONE = 0, WF_WINDOW_HANN, WF_WINDOW_HAMMING, WF_WINDOW_BLACKMAN,
    WF_WINDOW_BLACKMAN_HARRIS, WF_WINDOW_POWER_OF_SINE
} wf_window;
/* enum TSmoothingMode, src/source.hpp:56-61 */
typedef enum wf_tsmoothing { WF_TSMOOTH_NONE = 0, WF_TSMOOTH_EXPONENTIAL, WF_TSMOOTH_TVEXPONENTIAL } wf_tsmoothing;
/* enum InterpMode, src/source.hpp:43-48 */
typedef enum wf_interp { WF_INTERP_POINT = 0, WF_INTERP_LANCZOS, WF_INTERP_CATROM } wf_interp;
/* enum FilterMode, src/source.hpp */
typedef enum wf_filter { WF_FILTER_NONE = 0, WF_FILTER_GAUSS } wf_filter;

typedef struct wf_config {
    uint32_t fft_size;          /* m_fft_size: a multiple of 16 in [128, 65536] (src/source.cpp:562-565) */
    uint32_t sample_rate;       /* m_audio_info.samples_per_sec */
    uint32_t capture_channels;  /* m_capture_channels: 1 or 2 */
    uint32_t stereo;            /* m_stereo (channel_mode == "stereo") */
    int32_t window;             /* m_window_func (wf_window) */
    int32_t sine_exponent;      /* m_sine_exponent */
    int32_t tsmoothing;         /* m_tsmoothing (wf_tsmoothing) */
    float gravity;              /* m_gravity */
    uint32_t fast_peaks;        /* m_fast_peaks */
    float slope;                /* m_slope */
    float rolloff_q;            /* m_rolloff_q */
    float rolloff_rate;         /* m_rolloff_rate */
    int32_t cutoff_low;         /* m_cutoff_low  (Hz) */
    int32_t cutoff_high;        /* m_cutoff_high (Hz) */
    int32_t floor_db;           /* m_floor */
    int32_t ceiling_db;         /* m_ceiling */
    uint32_t normalize_volume;  /* m_normalize_volume */
    float volume_target;        /* m_volume_target */
    float max_gain;             /* m_max_gain */
    /* bar display (render_bars, src/source.cpp:1473-1567); bars are produced only when bars != 0 */
    uint32_t bars;              /* display_mode is BAR or STEPPED_BAR */
    int32_t interp_mode;        /* m_interp_mode (wf_interp) */
    uint32_t log_scale;         /* m_log_scale */
    uint32_t mirror_freq_axis;  /* m_mirror_freq_axis */
    uint32_t width;             /* m_width */
    uint32_t height;            /* m_height */
    int32_t bar_width;          /* m_bar_width */
    int32_t bar_gap;            /* m_bar_gap */
    int32_t channel_spacing;    /* m_channel_spacing */
    int32_t min_bar_height;     /* m_min_bar_height */
    uint32_t rounded_caps;      /* m_rounded_caps (changes border_top / border_bottom) */
    /* curve display (render_curve, src/source.cpp:1360-1425): `width` interpolated points per displayed channel instead
     * of bars; produced only when bars == 0 && curve != 0 */
    uint32_t curve;             /* display_mode is CURVE */
    /* smoothing filter across the bars / curve points before the dB -> pixel mapping (src/source.cpp:1396-1405, 1535-1545) */
    int32_t filter_mode;        /* m_filter_mode (wf_filter) */
    float filter_radius;        /* m_filter_radius: sigma of the Gaussian (src/source.cpp:527, 1279-1280) */
    /* level meter (display_mode METER / STEPPED_METER -> m_meter_mode, src/source.cpp:651-656; tick_meter,
     * src/source_generic.cpp:182-269).  With meter != 0 the handle is a meter batch: update()'s overrides apply
     * (window NONE, stereo off, slope 0, no normalisation / mirror / filter, interp POINT, src/source.cpp:1106-1128),
     * fft_size is ignored on input and becomes the meter buffer length sample_rate * (meter_ms / 1000.0) & -16 (:1121),
     * and the outputs are one level per captured channel (m_meter_val) plus its bar (render_bars, :1505-1509). */
    uint32_t meter;             /* m_meter_mode */
    uint32_t meter_rms;         /* m_meter_rms: RMS (1) or peak (0) */
    int32_t meter_ms;           /* m_meter_ms: milliseconds of audio the level is taken over (meter) / shown (waveform) */
    /* waveform display (display_mode WAVEFORM; tick_waveform, src/source_generic.cpp:271-390): a history of `width` points
     * per channel, one every meter_ms / width, each the dBFS of |sample| at that time.  With waveform != 0 the handle is a
     * waveform batch: fft_size is ignored on input and becomes `width` (m_fft_size = m_width, src/source.cpp:1140), the rings
     * hold m_waveform_samples = sample_rate * (meter_ms / 1000.0) samples (+ the A/V-sync reserve), window / slope / mirror /
     * log scale are off (:1133-1137). */
    uint32_t waveform;          /* m_display_mode == DisplayMode::WAVEFORM */
    /* vertex fill (the loops of render_bars / render_curve that write the vertex buffer, src/source.cpp:1576-1659, :1436-1461):
     * with vertices != 0 (and bars or curve) every tick also leaves, per displayed channel, the vertices the reference
     * hands to gs_draw -- WF_HIP_OUT_VERTICES.  1: filled geometry (bars: two triangles per bar, plus the cap fans with
     * rounded_caps; curve: a triangle strip of 2 * width vertices, RenderMode SOLID / GRADIENT / ...); 2: the curve as a
     * line strip of width vertices (RenderMode::LINE); 3: stepped bars (display_mode STEPPED_BAR, :1583-1607): per bar as
     * many step quads of step_width pixels, step_width + step_gap apart, as fit under its height -- the number of
     * vertices then changes from tick to tick (WF_HIP_OUT_VERTEX_COUNTS). */
    uint32_t vertices;
    int32_t step_width;         /* m_step_width (vertices == 3) */
    int32_t step_gap;           /* m_step_gap */
    /* radial layout (m_radial, src/source.cpp:508, :658-666).  The polar transform itself is the plugin's vertex shader
     * (:1745-1762); what changes on the CPU side -- and here -- are the cap fans of rounded bars, which become full circles
     * "to avoid distortion issues" (:1296, :1632-1633, :1646-1647).  `height` is m_height as get_settings leaves it (halved,
     * minus the dead zone). */
    uint32_t radial;
} wf_config;

/* get_defaults (src/source.cpp:119-174) + what update() derives for 48 kHz stereo OBS audio,
 * in bar display mode off.  Callers then override what the configuration under test needs. */
void wf_config_defaults(wf_config *cfg);

#ifdef __cplusplus
}
#endif
#endif
