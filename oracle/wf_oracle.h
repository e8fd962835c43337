/*
 * wf_oracle.h -- CPU restatement of the reference's spectrum hot path.  TEST INFRASTRUCTURE.
 *
 * NOT part of the product: only tests/, tools/make_golden.py, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; libwaveform_hip.so never does.
 *
 * What it restates (plain C, one source = one WAVSource):
 *   setup       WAVSource::update()            src/source.cpp:1169-1290, init_interp :837-896,
 *                                              init_rolloff :898-918
 *   ingest      WAVSource::capture_audio()     src/source.cpp:1873-1886 (+ CircularBuffer)
 *   tick        WAVSourceGeneric::tick_spectrum  src/source_generic.cpp:26-180, including the
 *                                              hidden/timeout reset (:34-48) and the silence
 *                                              state machine (:63-95, :138-139)
 *   bars        render_bars' interpolation + dB->pixel mapping, src/source.cpp:1500-1557 with
 *               src/filter.hpp:160-211 (scalar apply_interp_filter, bar version)
 *
 * The FFT (FFTW 3.3.11 r2c in the reference, src/source.cpp:1187) is restated as the
 * mathematical DFT it computes (deps/fftw-3.3.11/doc/reference.texi:1926-1939: forward, sign -1,
 * unnormalised), evaluated in double precision and rounded once to float; FFTW's own float
 * result differs from that by float round-off (L2 rel. ~3e-8).
 *
 * PARITY PIN: this restatement is checked against the reference itself -- oracle/_ref
 * (libwfref.so: the reference's own TUs + vendored FFTW, built here from /root/reference) --
 * live in tests/test_oracle_vs_ref.py, and against golden vectors generated from it
 * (tests/golden/, tools/make_golden.py) which travel to the GPU box.
 */
#ifndef WF_ORACLE_H
#define WF_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/wf_config.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wfo_source wfo_source;

wfo_source *wfo_create(const wf_config *cfg);
void wfo_destroy(wfo_source *s);

/* capture_audio: append frames per captured channel (ch1 may be NULL for 1 channel), then trim the
 * ring to sync_delay + fft_size samples */
void wfo_push_audio(wfo_source *s, const float *ch0, const float *ch1, uint32_t frames, int muted);
/* frames of audio that lie after the tick time (dtaudio > 0, src/source_generic.cpp:50-51) */
void wfo_set_sync_delay(wfo_source *s, uint32_t frames);
/* m_show / capture timeout: hidden != 0 takes the reset branch (src/source_generic.cpp:34-48) */
void wfo_set_hidden(wfo_source *s, int hidden);
void wfo_set_input_rms(wfo_source *s, float rms);
/* update_input_rms() from what wfo_push_audio collected (cfg.normalize_volume): what WAVSource::tick does first,
 * src/source.cpp:1330-1331; overwrites the value of wfo_set_input_rms */
void wfo_update_input_rms(wfo_source *s);
float wfo_input_rms(const wfo_source *s);
void wfo_tick(wfo_source *s, float seconds);
void wfo_render_bars(wfo_source *s);
/* the vertex-buffer loops of render_bars (plain bars incl. rounded caps, src/source.cpp:1609-1657) and render_curve
 * (:1436-1461) for one displayed channel, after wfo_render_bars: writes x, y, z, w per vertex (as libobs' vec3) and
 * returns the number of vertices the reference passes to gs_draw; line != 0: RenderMode::LINE (curve only) */
size_t wfo_fill_vertices(const wfo_source *s, int channel, int line, float *out, size_t cap);

uint32_t wfo_fft_size(const wfo_source *s);
uint32_t wfo_output_channels(const wfo_source *s);
int wfo_last_silent(const wfo_source *s);
size_t wfo_ring_samples(const wfo_source *s, int ch);
float wfo_gravity(const wfo_source *s, float seconds);
float wfo_db_min(void);
const float *wfo_decibels(const wfo_source *s, int ch);
const float *wfo_tsmooth(const wfo_source *s, int ch);
float *wfo_tsmooth_mut(wfo_source *s, int ch);
const float *wfo_window(const wfo_source *s, float *sum);
const float *wfo_slope(const wfo_source *s);
const float *wfo_rolloff(const wfo_source *s);
int wfo_num_bars(const wfo_source *s);
size_t wfo_interp_indices(const wfo_source *s, const float **out);
size_t wfo_band_widths(const wfo_source *s, const int **out);
size_t wfo_interp_weights(const wfo_source *s, const float **out, int *radius, int *taps);
const float *wfo_bars(const wfo_source *s, int ch); /* pixel y per bar after wfo_render_bars */

/* ---- level meter (wf_oracle_meter.c): WAVSourceGeneric::tick_meter, src/source_generic.cpp:182-269 ------------- */
typedef struct wfo_meter wfo_meter;
wfo_meter *wfo_meter_create(const wf_config *cfg); /* cfg->meter must be set */
void wfo_meter_destroy(wfo_meter *m);
void wfo_meter_push_audio(wfo_meter *m, const float *ch0, const float *ch1, uint32_t frames, int muted);
void wfo_meter_set_sync_delay(wfo_meter *m, uint32_t frames);
void wfo_meter_set_state(wfo_meter *m, int state); /* 0 shown, 1 !m_show, 2 capture timed out */
/* exact != 0: the RMS sum of squares is accumulated in double instead of the reference's sequential float sum (:236-241) --
 * the value that sum approximates; used by the tests to tell "differs from the reference" from "is less accurate than it" */
void wfo_meter_set_exact(wfo_meter *m, int exact);
void wfo_meter_tick(wfo_meter *m, float seconds);
void wfo_meter_render(wfo_meter *m);                /* render_bars' mapping of the levels */
uint32_t wfo_meter_size(const wfo_meter *m);        /* m_fft_size (meter buffer length) */
int wfo_meter_last_silent(const wfo_meter *m);
float wfo_meter_val(const wfo_meter *m, int ch);    /* m_meter_val */
float wfo_meter_ema(const wfo_meter *m, int ch);    /* m_meter_buf */
float wfo_meter_bar(const wfo_meter *m, int ch);    /* m_interp_bufs[0][ch] after wfo_meter_render */

/* ---- waveform display (wf_oracle_wave.c): WAVSourceGeneric::tick_waveform, src/source_generic.cpp:271-390 -------- */
typedef struct wfo_wave wfo_wave;
wfo_wave *wfo_wave_create(const wf_config *cfg); /* cfg->waveform must be set */
void wfo_wave_destroy(wfo_wave *w);
void wfo_wave_push_audio(wfo_wave *w, const float *ch0, const float *ch1, uint32_t frames, int muted);
/* m_audio_ts (end-of-audio timestamp of the newest captured sample) and the A/V-sync reserve in frames, as of the next
 * push / tick */
void wfo_wave_set_time(wfo_wave *w, uint64_t audio_ts_ns, uint32_t reserve_frames);
void wfo_wave_set_hidden(wfo_wave *w, int hidden); /* !m_show || capture timed out */
void wfo_wave_set_input_rms(wfo_wave *w, float rms);
float wfo_wave_update_input_rms(wfo_wave *w);  /* update_input_rms() from the pushed audio (cfg.normalize_volume); returns m_input_rms */
void wfo_wave_tick(wfo_wave *w);
uint32_t wfo_wave_points(const wfo_wave *w);        /* m_fft_size = m_width */
uint32_t wfo_wave_output_channels(const wfo_wave *w);
int wfo_wave_last_silent(const wfo_wave *w);
const float *wfo_wave_row(const wfo_wave *w, int ch); /* m_decibels[ch], wfo_wave_points() floats */
uint64_t wfo_wave_ts(const wfo_wave *w);            /* m_waveform_ts */

/* the bare DFT stage, for FFT-only tests: out[k] = (re, im) of bin k, k < n/2 */
void wfo_r2c(const float *in, uint32_t n, float *out_interleaved);

#ifdef __cplusplus
}
#endif
#endif
