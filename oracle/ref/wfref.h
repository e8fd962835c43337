/*
 * wfref.h -- C API of oracle/_ref/libwfref.so (TEST INFRASTRUCTURE).
 *
 * libwfref.so IS the reference: phandasm/waveform v1.9.1's own translation
 * units (src/source.cpp, source_generic.cpp, source_avx.cpp, source_avx2.cpp,
 * filter_fma3.cpp, module.cpp) compiled verbatim from /root/reference together
 * with the vendored FFTW 3.3.11, linked against the headless fake libobs in
 * fake_obs/.  This API drives a WAVSource the way OBS does (create -> update ->
 * audio callback -> video_tick -> video_render) and exposes the protected
 * members the hot path reads and writes (src/source.hpp:101-247) so tests can
 * compare them with the oracle restatement and with the HIP path.
 *
 * Only tests/, tools/make_golden.py, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product never does.
 */
#ifndef WFREF_H
#define WFREF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wfref wfref_t;

/* isa: "hip" -> WAVSourceHIP (host/wav_source_hip.cpp: the reference-side binding of libwaveform_hip.so, loaded with
 *      dlopen from $WF_HIP_LIBRARY); wfref_using_hip() tells whether the device path is really active.
 * isa: "generic" | "avx" | "avx2" -> WAVSourceGeneric / WAVSourceAVX / WAVSourceAVX2
 *      (the classes callbacks::create picks from, src/source.cpp:87-102).
 * settings: "key=value;key=value" over the plugin's own setting keys
 *      (src/settings.hpp), applied on top of get_defaults (src/source.cpp:119-174).
 * channels: OBS audio channel count (1 = mono, 2 = stereo, ...).
 * Returns NULL on failure. */
wfref_t *wfref_create(const char *isa, const char *settings, uint32_t sample_rate, int channels,
                      uint32_t fps_num, uint32_t fps_den);
void wfref_destroy(wfref_t *h);
/* re-run WAVSource::update() with new overrides (keeps earlier overrides) */
void wfref_update(wfref_t *h, const char *settings);

/* fake clock (thread-local): what os_gettime_ns() returns */
void wfref_set_clock_ns(uint64_t ns);
uint64_t wfref_clock_ns(void);

/* deliver one audio packet through the registered capture callback
 * (src/source.cpp:1817-1888). ch1 may be NULL. timestamp_ns = audio_data.timestamp. */
void wfref_push_audio(wfref_t *h, const float *ch0, const float *ch1, uint32_t frames, uint64_t timestamp_ns, int muted);
/* convenience used by every test: advance the clock to `now_ns`, push `frames`
 * samples whose end-of-audio timestamp equals now_ns (so get_audio_sync() == 0),
 * then call video_tick(seconds). */
void wfref_feed_and_tick(wfref_t *h, const float *ch0, const float *ch1, uint32_t frames, uint64_t now_ns, float seconds);
void wfref_tick(wfref_t *h, float seconds);
void wfref_render(wfref_t *h);
/* what the last wfref_render() handed to gs_draw (src/source.cpp:1463-1465, :1661-1664): one call per displayed channel;
 * returns the number of vertices drawn, *points = 4 floats (x, y, z, w) per vertex of the vertex buffer at that moment */
int wfref_draw_count(wfref_t *h);
int wfref_shader_value(wfref_t *h, const char *name, float out[4]); /* the last value set_shader_vars gave the parameter; 0: never set */
size_t wfref_draw(wfref_t *h, int i, int *mode, const float **points);
void wfref_show(wfref_t *h, int show);

/* ---- state of the object (valid until the next update/destroy) ---- */
size_t wfref_fft_size(wfref_t *h);
uint32_t wfref_capture_channels(wfref_t *h);
uint32_t wfref_output_channels(wfref_t *h);
int wfref_stereo(wfref_t *h);
int wfref_last_silent(wfref_t *h);
size_t wfref_ring_bytes(wfref_t *h, int ch);
int wfref_using_hip(wfref_t *h);
/* the class instantiated for this source: "hip", "avx2", "avx" or "generic" (isa "create": whatever the plugin's own
 * obs_source_info::create -- callbacks::create, src/source.cpp:87-102 -- chose) */
const char *wfref_class_name(wfref_t *h);
/* process-wide: ticks a WAVSourceHIP had to hand to the reference's CPU class since the library was loaded (0 for a
 * device path that never fell back; ticks skipped for lack of audio are not fallbacks) */
uint64_t wfref_hip_fallback_ticks(void);
/* update_input_rms calls of WAVSourceHIP sources that ran on the host (the batched mode feeds the device instead) */
uint64_t wfref_hip_host_rms_updates(void);
/* render() calls of WAVSourceHIP spectrum sources that drew from the device's vertices / that ran the reference's own render
 * (apply_interp_filter*, apply_filter*, the vertex loops) on the host */
uint64_t wfref_hip_device_renders(void);
uint64_t wfref_hip_host_renders(void);
float wfref_gravity(wfref_t *h, float seconds);    /* get_gravity(), src/source.hpp:301-312 */
float wfref_db_min(void);
const float *wfref_decibels(wfref_t *h, int ch);   /* m_decibels[ch], fft_size/2 floats */
const float *wfref_tsmooth(wfref_t *h, int ch);    /* m_tsmooth_buf[ch] or NULL */
const float *wfref_window(wfref_t *h);             /* m_window_coefficients or NULL */
float wfref_window_sum(wfref_t *h);
const float *wfref_slope(wfref_t *h);              /* m_slope_modifiers or NULL */
const float *wfref_rolloff(wfref_t *h);            /* m_rolloff_modifiers or NULL */
int wfref_num_bars(wfref_t *h);
size_t wfref_interp_indices(wfref_t *h, const float **out); /* m_interp_indices */
size_t wfref_band_widths(wfref_t *h, const int **out);      /* m_band_widths */
size_t wfref_interp_kernel(wfref_t *h, const float **out, int *radius, int *size); /* m_interp_kernel.weights */
size_t wfref_bars(wfref_t *h, int ch, const float **out);   /* m_interp_bufs[ch] after render(): pixel y */

/* level meter / volume normalisation / waveform state */
int wfref_meter_mode(wfref_t *h);                  /* m_meter_mode */
float wfref_meter_val(wfref_t *h, int ch);         /* m_meter_val[ch] (dBFS) */
float wfref_meter_buf(wfref_t *h, int ch);         /* m_meter_buf[ch] (EMA state) */
float wfref_input_rms(wfref_t *h);                 /* m_input_rms after tick()'s update_input_rms() */
size_t wfref_decibels_size(wfref_t *h);            /* floats in m_decibels[ch]: fft_size/2 (spectrum), fft_size (meter, waveform) */

/* ---- CPU baseline ----
 * n_streams WAVSource objects split statically over n_threads std::threads; each tick
 * every stream receives `hop` new frames/channel of counter-hash white noise through
 * the capture callback and then video_tick(seconds).  Returns spectra per second over
 * the timed ticks (1 spectrum = 1 channel of 1 stream-tick). */
double wfref_bench(const char *isa, const char *settings, uint32_t sample_rate, int channels,
                   int n_streams, int n_threads, int warmup_ticks, int timed_ticks, int hop,
                   uint64_t seed, double *elapsed_s);

/* The source under OBS' threads (src/source.hpp:98-101): one audio thread per source pushing packets through the capture callback,
 * a video thread ticking and rendering every source, a UI thread calling update() / show / hide / destroy + create on random
 * ones, for `seconds`; then every source against a fresh one of the same class on a serial script.  Returns the number of
 * sources that differ (0 = pass), -1 if a source could not be created.  stats[6]: ticks, packets, updates, re-creations,
 * renders, sources compared. */
int wfref_thread_stress(const char *isa, const char *settings, int n_sources, double seconds, uint64_t seed, uint64_t *stats);

/* != 0: every frame of wfref_bench also renders every source (video_render behind the frame's ticks, as OBS does) */
void wfref_bench_set_render(int on);

/* counter-hash white noise shared by oracle, harness and device generator */
float wfref_noise(uint64_t seed, uint32_t stream, uint32_t channel, uint64_t index);

#ifdef __cplusplus
}
#endif
#endif
