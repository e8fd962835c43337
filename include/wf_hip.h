/*
 * wf_hip.h -- C ABI of libwaveform_hip.so, the MI355X (gfx950) implementation of
 * phandasm/waveform's per-tick DSP: the spectrum path and, behind the same interface, the
 * level meter, the waveform display and the volume-normalisation RMS.
 *
 * Boundary.  The reference selects its DSP kernels through four virtuals on
 * WAVSource (src/source.hpp:273-277: update_input_rms, tick_spectrum, tick_meter,
 * tick_waveform); callbacks::create (src/source.cpp:87-102) instantiates WAVSourceAVX2 /
 * WAVSourceAVX / WAVSourceGeneric.  This library is what a fourth subclass, WAVSourceHIP,
 * calls from its tick_spectrum() / tick_meter() / tick_waveform() overrides
 * (host/wav_source_hip.hpp; the binding a maintainer adds is in INTEGRATION.md).
 * One wf_hip handle serves a *batch* of independent sources ("streams") that share
 * one configuration, because a GPU only pays off batched.
 *
 * Mapping of entry points to the reference code they replace:
 *   wf_hip_create        WAVSource::update(): buffers, FFTW plan, window/slope/rolloff/
 *                        interp tables (src/source.cpp:1169-1290, :837-918)
 *   wf_hip_destroy       WAVSource::free_bufs() (src/source.cpp:782-808)
 *   wf_hip_reset         state init in update(): m_tsmooth_buf = 0, m_decibels = DB_MIN,
 *                        rings pre-filled with N zero samples (:1170-1182, :1243-1248)
 *   wf_hip_push_audio*   WAVSource::capture_audio(): CircularBuffer::push_back per channel
 *                        (src/source.cpp:1873-1886, src/circular_buffer.hpp:42-63)
 *   wf_hip_tick          WAVSource*::tick_spectrum(seconds) for every stream of the batch
 *                        (src/source_generic.cpp:26-180 -- the parity target; AVX variants
 *                        src/source_avx.cpp:29-200, src/source_avx2.cpp:24-209), plus, when
 *                        the configuration displays bars, the bar reduction of render_bars
 *                        (src/source.cpp:1500-1557; src/filter.hpp:160-211)
 *   wf_hip_read[_async]  reading m_decibels / m_interp_bufs / m_tsmooth_buf / m_meter_val ... (wf_hip_output)
 *   wf_hip_enable_input_rms
 *                        capture_audio's RMS part + sync_rms_buffer + update_input_rms
 *                        (src/source.cpp:1842-1871, :810-835; src/source_generic.cpp:392-403)
 * FFT sizes: every multiple of 16 from 128 to 65536, the reference's own range with "enable large FFT" (src/source.cpp:349,
 * :359-363, :562-565).  Powers of two up to 32768 and the other sizes up to 16384 -- as a mixed-radix transform where the
 * size has small prime factors and at most one prime factor of up to 127 (the automatic sizes, 114 of the slider's 120
 * positions that are not powers of two), by Bluestein's algorithm otherwise -- run inside one fused kernel; 65536 runs in one
 * kernel of its own; the other sizes above 16384 as rows of n/2 = C R points with one complex scratch buffer in device memory
 * (wf_big.hpp: rows of a mixed-radix transform -- two rows: in one kernel without scratch --, or rows by Bluestein inside LDS; 12-28 % of
 * the HBM roofline).
 * Anything else -> WF_HIP_ERR_UNSUPPORTED.
 *
 * Waveform display.  A handle created from a configuration with cfg.waveform != 0 is a *waveform batch*: wf_hip_tick runs
 * WAVSource*::tick_waveform (src/source_generic.cpp:271-390) for every stream -- a history of cfg.width dBFS points per
 * channel, extended by one point per meter_ms / width of newly consumed audio -- and WF_HIP_OUT_DECIBELS holds the rows
 * ([count][output_channels][width]; wf_hip_fft_size() == width as m_fft_size does in this mode).  The tick needs
 * wf_hip_tick_params::audio_ts_ns.  Widths up to 8192 points.
 *
 * Level meter.  A handle created from a configuration with cfg.meter != 0 is a *meter batch*: wf_hip_tick runs
 * WAVSource*::tick_meter (src/source_generic.cpp:182-269; AVX src/source_avx.cpp:202-322) for every stream -- the
 * meter buffer is the last wf_hip_fft_size() samples consumed from the device ring, RMS or peak over it, temporal
 * smoothing, dBFS, m_last_silent -- and WF_HIP_OUT_METER / WF_HIP_OUT_BARS hold m_meter_val and the bars
 * render_bars draws from it (src/source.cpp:1505-1509, :1548-1557).  Spectrum-only entry points
 * (WF_HIP_OUT_DECIBELS, WF_HIP_OUT_TSMOOTH / wf_hip_write_tsmooth, wf_hip_table) fail with WF_HIP_ERR_INVALID on a meter batch.
 *
 * Conventions: plain C types only; every function returns WF_HIP_OK (0) or a negative
 * wf_hip_status, never throws, never aborts; wf_hip_last_error() gives the text.  A handle
 * is used by one thread at a time (the reference holds m_mtx around tick/update,
 * src/source.cpp:1326,1079); different handles may be used concurrently.  Host buffers
 * passed in are borrowed for the duration of the call.  All work of a handle is issued on
 * its own HIP stream; functions that return data to the host synchronise that stream.
 *
 * There is NO CPU fallback in this library: without a usable gfx950 device
 * wf_hip_create fails with WF_HIP_ERR_NO_DEVICE and the caller keeps using its
 * CPU class (WAVSourceGeneric) -- the reference's own failure mode for a missing
 * FFTW plan is to skip the channel (src/source_generic.cpp:105-108).
 */
#ifndef WF_HIP_H
#define WF_HIP_H
#include <stddef.h>
#include <stdint.h>
#include "wf_config.h"

#ifdef __cplusplus
extern "C" {
#endif

#define WF_HIP_ABI_VERSION 13

typedef enum wf_hip_status {
    WF_HIP_OK = 0,
    WF_HIP_ERR_INVALID = -1,     /* bad argument / configuration */
    WF_HIP_ERR_UNSUPPORTED = -2, /* legal for the reference, not implemented here (e.g. a waveform display wider than 8192 points) */
    WF_HIP_ERR_NO_DEVICE = -3,   /* no usable HIP device */
    WF_HIP_ERR_RUNTIME = -4,     /* a HIP call failed; see wf_hip_last_error */
    WF_HIP_ERR_NOMEM = -5
} wf_hip_status;

typedef struct wf_hip wf_hip; /* opaque */

/* ---- library ---------------------------------------------------------------------- */
int wf_hip_abi_version(void);
/* number of usable HIP devices (0 when there is none; never fails) */
int wf_hip_device_count(void);
/* text of the last error on this handle (or of the last failed create when h == NULL) */
const char *wf_hip_last_error(const wf_hip *h);

/* ---- lifetime ----------------------------------------------------------------------- */
/* max_streams: batch size (independent WAVSource instances sharing cfg).
 * ring_frames: capacity of each per-channel device ring in samples; 0 = default
 *              (smallest power of two >= max(2 * fft_size, 4096)). Rounded up to a power of two.  A packet longer than
 *              the ring keeps its newest ring_frames samples. */
int wf_hip_create(const wf_config *cfg, int device, uint32_t max_streams, uint32_t ring_frames, wf_hip **out);
void wf_hip_destroy(wf_hip *h);
/* re-initialise streams [first, first+count): smoothing state 0, decibels DB_MIN, rings = N zeros */
int wf_hip_reset(wf_hip *h, uint32_t first, uint32_t count);

/* ---- geometry of the batch ---------------------------------------------------------- */
uint32_t wf_hip_fft_size(const wf_hip *h);
uint32_t wf_hip_num_streams(const wf_hip *h);
uint32_t wf_hip_capture_channels(const wf_hip *h);
uint32_t wf_hip_output_channels(const wf_hip *h);  /* m_output_channels */
uint32_t wf_hip_display_channels(const wf_hip *h); /* m_stereo ? 2 : 1 */
uint32_t wf_hip_num_bars(const wf_hip *h);         /* outputs per displayed row: m_num_bars with cfg.bars, m_width with cfg.curve
                                                      (render_curve, src/source.cpp:1360-1425), 0 with neither */
uint32_t wf_hip_ring_frames(const wf_hip *h);

/* ---- audio ingest ------------------------------------------------------------------- */
/* Append `frames` samples per channel to streams [first, first+count).
 * Layout of `samples` (host memory): [count][capture_channels][frames], planar float32 --
 * what capture_audio receives per source in audio_data::data[] (src/source.cpp:1873-1882). */
int wf_hip_push_audio(wf_hip *h, uint32_t first, uint32_t count, const float *samples, uint32_t frames);
/* Pipelined ingest: page-locked host buffers (wf_hip_host_alloc) are copied by DMA without the runtime's bounce buffer, on a
 * second HIP stream of the handle, so that the copy of the next packet runs under the tick of the previous one; the ring
 * append is ordered behind the copy, the next tick behind the append.  The call does not wait: `samples` must stay
 * untouched until wf_hip_ingest_done(slot) -- two buffers used alternately (slot 0 / 1) keep a 60 fps loop from ever waiting.
 * Layout as wf_hip_push_audio.  Measured (4096 stereo streams, one 800-frame hop each per step). */
int wf_hip_push_audio_async(wf_hip *h, uint32_t first, uint32_t count, const float *pinned_samples, uint32_t frames, uint32_t slot);
/* blocks until the copy issued with `slot` (0 or 1) has left the host buffer */
int wf_hip_ingest_done(wf_hip *h, uint32_t slot);
/* Hops of different lengths (a plugin's sources tick with whatever their capture buffers gained): stream first+i appends
 * frames[i] <= max_frames frames from pinned_samples[i][channel][0 .. max_frames); frames[i] == 0 leaves it alone.
 * `frames` is copied before the call returns; `pinned_samples` follows the rules of wf_hip_push_audio_async (slot 0 / 1,
 * wf_hip_ingest_done).  Not available while the device RMS producer is enabled. */
int wf_hip_push_audio_ragged_async(wf_hip *h, uint32_t first, uint32_t count, const float *pinned_samples, const uint32_t *frames,
                                   uint32_t max_frames, uint32_t slot);

/* page-locked host memory for wf_hip_push_audio_async (hipHostMalloc); NULL on failure */
void *wf_hip_host_alloc(size_t bytes);
void wf_hip_host_free(void *p);
/* Same, from a device pointer on the handle's device (no PCIe crossing). */
int wf_hip_push_audio_device(wf_hip *h, uint32_t first, uint32_t count, const float *d_samples, uint32_t frames);
/* Same, but the samples are generated on the device by the counter hash of wf_synth.h:
 * stream s, channel c receives wf_synth_noise(seed, stream_id0 + s, c, index0 + i), i < frames. */
int wf_hip_push_synth(wf_hip *h, uint32_t first, uint32_t count, uint64_t seed, uint32_t stream_id0,
                      uint64_t index0, uint32_t frames);
/* muted / silent packet: the rings receive `frames` zeros, CircularBuffer::push_back_zero (src/source.cpp:1879-1880).  `samples`
 * (layout of wf_hip_push_audio) may be NULL: a packet without data.  A muted packet that still carries samples (muted &&
 * !m_ignore_mute) passes them: capture_audio takes the volume-normalisation RMS from the packet itself
 * (src/source.cpp:1842-1871), so the device RMS producer (wf_hip_enable_input_rms) receives them; without the producer they are
 * not read. */
int wf_hip_push_audio_muted(wf_hip *h, uint32_t first, uint32_t count, const float *samples, uint32_t frames);

/* ---- the tick ------------------------------------------------------------------------- */
typedef struct wf_hip_tick_params {
    float seconds;          /* tick_spectrum(seconds): only TVEXPONENTIAL smoothing uses it */
    uint32_t delay_frames;  /* audio already captured beyond the tick time: the window is the fft_size
                               samples ending delay_frames before the newest one
                               (dtaudio > 0 in src/source_generic.cpp:50-51) */
    float input_rms;        /* m_input_rms, only read when cfg.normalize_volume; the same value for every stream unless
                               wf_hip_set_input_rms has given the streams their own */
    uint32_t flags;         /* WF_HIP_TICK_* */
    uint64_t audio_ts_ns;   /* m_audio_ts: end-of-audio timestamp of the newest pushed sample (src/source.cpp:1829-1832), in ns;
                               only waveform batches read it (tick_waveform places its points in time with it) */
} wf_hip_tick_params;
/* bars/curve-only batch mode: skip the m_decibels store (cfg.bars or cfg.curve must be set).  The silence state machine
 * (src/source_generic.cpp:74-95) keeps working: the kernel leaves a one-word verdict per row ("a value > floor - 10") for the
 * next tick's test instead.  After the first such tick, WF_HIP_OUT_DECIBELS holds rows only as fresh as the last tick
 * without the flag that rewrote them.  Mono mixdown of two captured channels stores its (single) row regardless, and so do
 * the batches whose outputs are derived from the stored rows by a kernel of their own (fft sizes beyond a CU's LDS; displays
 * whose Gaussian-filter staging does not fit the tick kernel's on-chip buffer): there the flag is accepted and changes nothing. */
#define WF_HIP_TICK_NO_DECIBELS 1u

/* Asynchronous: enqueues the fused kernel for all streams on the handle's stream. */
int wf_hip_tick(wf_hip *h, const wf_hip_tick_params *p);
/* show()/hide()/capture-timeout per stream: hidden streams take the reset branch of
 * tick_spectrum (src/source_generic.cpp:34-48).  mask[i] != 0 -> hidden.  tick_meter tells the two causes apart
 * (capture timeout: the meter buffer is cleared and nothing is consumed, src/source_generic.cpp:184-199; !m_show: the
 * audio is consumed, then the state is reset, :222-230), so a host passes WF_HIP_HIDDEN_TIMEOUT for the former. */
#define WF_HIP_SHOWN 0
#define WF_HIP_HIDDEN 1          /* !m_show */
#define WF_HIP_HIDDEN_TIMEOUT 2  /* m_tick_ts - m_capture_ts > CAPTURE_TIMEOUT */
#define WF_HIP_PAUSED 3          /* the source was not ticked in this video frame (OBS ticks only active sources; a waveform source whose buffers
                                    hold no more than the A/V-sync reserve returns before it touches anything, src/source_generic.cpp:293-295) --
                                    the next wf_hip_tick leaves the stream exactly as it is; cleared by any other value */
#define WF_HIP_STARVED 4         /* spectrum batches whose host keeps the sources' own buffers (the plugin binding): the source holds fewer
                                    samples than window + A/V-sync delay (src/source_generic.cpp:55-61: every channel is skipped) -- the
                                    next wf_hip_tick processes no channel of the stream but still runs the reference's end-of-tick pass
                                    over the rows as they are (:138-179: dbfs of a stale dB value is DB_MIN; volume normalisation and
                                    roll-off on top), as the kernel does for a stream whose device ring is that short; stays until
                                    another value is set */
int wf_hip_set_hidden(wf_hip *h, uint32_t first, uint32_t count, const uint8_t *mask);
/* A/V-sync delay per stream, in frames (dtaudio > 0 of each source, src/source_generic.cpp:50-51), for batches whose
 * sources run on their own audio timestamps: stream first+i analyses the window ending delay[i] + the tick's common
 * delay_frames before its newest sample; stays in force until the next call for that stream.  Each delay[i] + fft_size
 * must fit the ring (and so must their sum with any wf_hip_tick_params::delay_frames used later). */
int wf_hip_set_stream_delay(wf_hip *h, uint32_t first, uint32_t count, const uint32_t *delay_frames);
/* Waveform batches whose sources run on their own audio timestamps (the plugin's batched mode): m_audio_ts of stream
 * first+i (src/source.hpp; what tick_waveform measures every point's time against, src/source_generic.cpp:306-331), in
 * nanoseconds.  Stays in force until the next call for that stream; once any stream has been given one,
 * wf_hip_tick_params::audio_ts_ns is ignored (streams never set read 0: "no audio has a timestamp yet").
 * WF_HIP_ERR_INVALID for batches that are not waveform displays. */
int wf_hip_set_stream_audio_ts(wf_hip *h, uint32_t first, uint32_t count, const uint64_t *audio_ts_ns);
/* m_input_rms per stream (what update_input_rms leaves, src/source_generic.cpp:392-403), for batches whose streams are
 * normalised independently (cfg.normalize_volume): rms[i] belongs to stream first+i and stays in force until the next
 * call for that stream.  Once any stream has been given a value, wf_hip_tick_params::input_rms is ignored (streams never
 * set read 0, i.e. the full max_gain, as a source that has not seen audio yet does).  The volume compensation
 * min(volume_target - dbfs(rms), max_gain) (src/source_generic.cpp:163) is evaluated here, on the host, in float. */
int wf_hip_set_input_rms(wf_hip *h, uint32_t first, uint32_t count, const float *rms);
/* Volume normalisation produced on the device: update_input_rms for every stream (src/source_generic.cpp:392-403 with
 * sync_rms_buffer, src/source.cpp:810-835, fed by the RMS part of capture_audio, :1842-1871).  feed == 0: after this call every
 * wf_hip_push_* also appends the squared per-frame peak of the captured channels to a per-stream RMS ring, and every
 * wf_hip_tick first recomputes m_input_rms over the m_input_rms_size (= sample_rate & -16) frames that end at the
 * A/V-sync point, as WAVSource::tick does (src/source.cpp:1330-1331); wf_hip_tick_params::input_rms is then ignored and
 * wf_hip_set_input_rms fails.  Needs cfg.normalize_volume (spectrum or waveform batches).  Audio pushed before the call
 * counts as silence.
 * feed != 0: the same producer for hosts that already hold capture_audio's per-frame squared peaks -- the plugin binding: the
 * reference's own capture_audio fills m_rms_sync_buf (src/source.cpp:1842-1871, from the packet even when it is muted),
 * and WAVSourceHIP::update_input_rms (the override of src/source.hpp:273, src/source_generic.cpp:392-403) hands what
 * sync_rms_buffer would move into m_input_rms_buf this tick (src/source.cpp:810-835) to
 * wf_hip_push_rms_ragged_async instead of adding up 48000 floats per source and frame on the host.  The squared-peak
 * ring is then independent of the audio rings' positions; wf_hip_tick recomputes every stream's m_input_rms as above.
 * The two forms are mutually exclusive on a handle. */
int wf_hip_enable_input_rms(wf_hip *h, int feed);
/* sq: page-locked [count][max_frames] squared peaks, oldest first; frames[count] values are valid per stream (0: that
 * stream's sync_rms_buffer had nothing to consume).  max_frames <= sample_rate & -16.  Shares the ingest slots of
 * wf_hip_push_audio*_async: wf_hip_ingest_done(slot) says when `sq` may be written again. */
int wf_hip_push_rms_ragged_async(wf_hip *h, uint32_t first, uint32_t count, const float *pinned_sq, const uint32_t *frames,
                                 uint32_t max_frames, uint32_t slot);
/* waits for everything the handle has issued.  With WF_HIP_CANARY=1 in the environment of wf_hip_create every device block of the
 * handle ends in guard bytes, which this call then reads back: a kernel that wrote past a buffer makes it return
 * WF_HIP_ERR_RUNTIME with the block named in wf_hip_last_error (a debugging aid: one small copy per block and sync) */
int wf_hip_sync(wf_hip *h);

/* ---- results ----------------------------------------------------------------------------- */
/* What a tick leaves per stream.  One synchronous reader and one pipelined one serve all of them (ABI 13; ABI 12 had a function
 * per output and per form).  Shapes per stream, in the order the reference's members hold them: */
typedef enum wf_hip_output {
    WF_HIP_OUT_DECIBELS = 0,   /* float [output_channels][fft_size/2]        m_decibels (dBFS); not on meter batches */
    WF_HIP_OUT_BARS,           /* float [display_channels][num_bars]         bar tops / curve points in pixels: m_interp_bufs after the optional
                                  Gaussian filter, the dB->y mapping and the mirror of render_bars (src/source.cpp:1535-1564) or
                                  render_curve (:1396-1424) */
    WF_HIP_OUT_PREMIRROR,      /* float [display_channels]                   cfg.mirror_freq_axis displays: render_bars / render_curve take the
                                  row's smallest y for the shader (miny / minpos; gradient and pulse render modes) BEFORE the outputs
                                  above the middle are replaced by images of the lower ones (src/source.cpp:1548-1567, :1411-1424).
                                  Without the Gaussian filter every output above the middle sits on the clamped top position and has one
                                  and the same value: that value (output num_bars / 2 + 1 before the mirror).  With it and the BARS rows
                                  the host finds the reference's miny / minpos without interpolating the row itself.  (With the filter on
                                  the outputs next to the middle are blends and the value is not enough: the plugin binding keeps the
                                  reference's own loops for that combination.) */
    WF_HIP_OUT_VERTICES,       /* float [display_channels][num_vertices][4]  cfg.vertices: x, y, z, w as libobs' vec3 holds them (z = w = 0), the
                                  vertices render_bars / render_curve write into their vertex buffer (src/source.cpp:1576-1659,
                                  :1436-1461), produced by every tick from the bars / curve points of that tick */
    WF_HIP_OUT_VERTEX_COUNTS,  /* uint32 [display_channels]                  stepped bars (cfg.vertices == 3): num_vertices is the buffer's capacity
                                  per channel (num_bars * 6 * max_steps, create_vbuf src/source.cpp:988-1000); how many of them a tick's
                                  draw call uses -- gs_draw(GS_TRIS, 0, vertpos), :1663; vertices beyond it are whatever earlier ticks
                                  left, as in the reference's buffer */
    WF_HIP_OUT_LAST_SILENT,    /* uint8                                      m_last_silent */
    WF_HIP_OUT_TSMOOTH,        /* float [capture_channels][fft_size/2]       m_tsmooth_buf (spectrum batches) */
    WF_HIP_OUT_METER,          /* float [capture_channels]                   meter batches: m_meter_val (dBFS) */
    WF_HIP_OUT_INPUT_RMS,      /* float                                      m_input_rms as of the last tick (wf_hip_enable_input_rms) */
    WF_HIP_OUT_WAVEFORM_TS     /* uint64                                     waveform batches: m_waveform_ts (src/source.hpp:135, the timestamp of
                                  the next point the sweep will draw, ns) -- what a source needs to continue the sweep on the host
                                  (src/source_generic.cpp:318-353) when it leaves a batch */
} wf_hip_output;
/* bytes per stream of an output of this batch (0: the batch has no such output) */
size_t wf_hip_output_bytes(const wf_hip *h, wf_hip_output what);
/* `what` of streams [first, first+count) as the ticks issued so far leave it, into `out` ([count] x the shape above); waits for
 * those ticks.  WF_HIP_ERR_INVALID when the batch has no such output (wf_hip_last_error says why). */
int wf_hip_read(wf_hip *h, wf_hip_output what, uint32_t first, uint32_t count, void *out);
/* m_tsmooth_buf written back (state restore; layout of WF_HIP_OUT_TSMOOTH) */
int wf_hip_write_tsmooth(wf_hip *h, uint32_t first, uint32_t count, const float *in);
/* Pipelined readback: what the ticks issued so far leave for streams [first, first+count) is copied into page-locked memory
 * (wf_hip_host_alloc) on the handle's readback stream, without waiting; the following ticks run meanwhile.  `slot` (0 or 1) names
 * the copy for wf_hip_readback_done, which blocks until everything of it has landed.  Every destination is [count] x the shape
 * of the output of that name; a NULL pointer leaves that output out.  Three combinations:
 *  - rows (+ last_silent, both required) and any of bars / premirror / vertices / vertex_counts / input_rms behind them: the
 *    plugin binding's frame (its render() override draws from them one frame later).  The copies read the handle's own
 *    buffers: the next wf_hip_tick waits (on the device, not the host) for a copy still in flight before it overwrites them.
 *  - bars alone: the device keeps one snapshot per slot (a device copy of a few MB at most behind the ticks), so a later tick
 *    neither waits for nor disturbs a copy in flight (bench.py's host-fed leg).
 *  - meter + last_silent (meter batches; both required): snapshots as well -- the plugin's batched mode reads every source's
 *    level one video frame late. */
typedef struct wf_hip_readback {
    float *rows;             /* WF_HIP_OUT_DECIBELS */
    uint8_t *last_silent;    /* WF_HIP_OUT_LAST_SILENT */
    float *bars;             /* WF_HIP_OUT_BARS */
    float *premirror;        /* WF_HIP_OUT_PREMIRROR */
    float *vertices;         /* WF_HIP_OUT_VERTICES */
    uint32_t *vertex_counts; /* WF_HIP_OUT_VERTEX_COUNTS */
    float *input_rms;        /* WF_HIP_OUT_INPUT_RMS */
    float *meter;            /* WF_HIP_OUT_METER */
} wf_hip_readback;
int wf_hip_read_async(wf_hip *h, uint32_t first, uint32_t count, const wf_hip_readback *dst, uint32_t slot);
int wf_hip_readback_done(wf_hip *h, uint32_t slot);
/* The bars copied device-to-device into `d_out` (a buffer on the handle's device, e.g. the send buffer of an RCCL all-gather;
 * [count][display_channels][num_bars]) without waiting: the copy is enqueued behind the ticks issued so far (every lane of a large batch copies the bars
 * of its own slice: the tick's concurrent launches are not joined) and `consumer_stream` (a hipStream_t of the caller, e.g.
 * the stream its RCCL all-gather runs on) is made to wait for it; the handle goes on with the next tick meanwhile.  The
 * caller keeps `d_out` untouched by anything else until its consumer has run (wf_hip_wait_event orders a reuse). */
int wf_hip_copy_bars_device_async(wf_hip *h, uint32_t first, uint32_t count, void *d_out, void *consumer_stream);
/* Zero-copy form of the above (ABI 13): from the next tick on, every tick ALSO leaves the bars of the whole batch -- the ones it
 * finishes and, copied over, the ones it leaves as they are (paused, hidden, silent streams) -- in caller-owned device memory of
 * the same shape ([num_streams][display_channels][num_bars] floats): the send buffer of an all-gather is written by the tick
 * kernel itself, nothing is enqueued behind the tick.  Two SETS of n <= 8 buffers each; every tick writes every buffer of the
 * current write set.  The buffers may be memory of peer devices this device can address (hipDeviceEnablePeerAccess must have
 * SUCCEEDED for the pair: a kernel store to an unmapped peer address faults): a shard then leaves its slice in every device's
 * gathered result itself, and the exchange of BASELINE configs[4] needs no copy and no collective kernel at all (the C ABI's
 * multi-device group, peer transport).  n = 0 turns it off.  The call waits for the ticks in flight (they write the old
 * buffers) before it replaces the sets: the caller may free the old buffers when it returns.
 * Power-of-two fft sizes up to 32768 whose display the tick kernel finishes itself ONLY: WF_HIP_ERR_UNSUPPORTED for the sizes that
 * are not powers of two (Bluestein / mixed-radix instantiations run at their register caps), for the sizes beyond a CU's LDS and
 * for filtered displays that do not fit the tick kernel's staging -- those keep wf_hip_copy_bars_device_async (wf_hip_multi_* and
 * waveform_amd.dist.BarsGather pick the path per handle).  WF_HIP_ERR_INVALID for level-meter and waveform batches. */
int wf_hip_set_bars_mirrors(wf_hip *h, uint32_t n, void *const *set0, void *const *set1);
/* Hand-over: `consumer_stream` is made to wait for the newest tick (every lane); *d_out = buffer 0 of the set that tick wrote,
 * and the OTHER set becomes the write set -- ticks issued after this call leave the handed-over set alone until the next
 * hand-over makes it the write set again.  (The write set changes here and only here: ticks between two hand-overs rewrite the
 * same set, a tick that fails changes nothing.)  If no tick has written the set since it became the write set, the call fills it
 * from the handle's own bars first (a device copy per buffer, behind the ticks issued so far).  Nothing waits on the host.
 * The caller's side of the protocol: whatever still reads the set that now becomes the write set (the consumer of the hand-over
 * before last) must have run before the next tick is issued -- an event of its own behind that consumer, a tick old by then:
 * hipEventSynchronize returns at once (a device-side wait in front of every tick instead cost 4 % of the tick rate). */
int wf_hip_bars_mirror_ready(wf_hip *h, void *consumer_stream, void **d_out);
/* everything the handle issues after this call (on all of its internal streams) waits, on the device, for `event` (a
 * hipEvent_t of the caller, recorded before the call) -- e.g. "the gather that read the buffer the next
 * wf_hip_copy_bars_device_async overwrites has run".  Does not wait on the host. */
int wf_hip_wait_event(wf_hip *h, void *event);
/* device pointers for zero-copy consumers on the same device (e.g. an RCCL all-gather of
 * the bars, or a renderer): valid until wf_hip_destroy */
float *wf_hip_decibels_device(wf_hip *h);
float *wf_hip_bars_device(wf_hip *h);
void *wf_hip_stream(wf_hip *h); /* hipStream_t */

/* ---- host tables (what update() precomputes), for tests and for hosts that render themselves */
typedef enum wf_hip_table_id {
    WF_HIP_TABLE_WINDOW = 0,     /* float [fft_size]   m_window_coefficients (src/source.cpp:1190-1226) */
    WF_HIP_TABLE_WINDOW_SUM,     /* float [1]          m_window_sum (:1228-1234) */
    WF_HIP_TABLE_SLOPE,          /* float [fft_size/2] m_slope_modifiers (:1283-1290); empty when cfg.slope <= 0 */
    WF_HIP_TABLE_ROLLOFF,        /* float [fft_size/2] m_rolloff_modifiers (:898-918) */
    WF_HIP_TABLE_INTERP_INDICES, /* float []           m_interp_indices (init_interp, :837-896) */
    WF_HIP_TABLE_BAND_WIDTHS,    /* int   [num_bars]   m_band_widths */
    WF_HIP_TABLE_INTERP_WEIGHTS, /* float [][taps]     m_interp_kernel's weights, one row per sample position */
    WF_HIP_TABLE_INTERP_SHAPE    /* int   [2]          {radius, taps} of the interpolation kernel */
} wf_hip_table_id;
/* number of elements; *out (may be NULL) = the table, owned by the handle, NULL when empty */
size_t wf_hip_table(const wf_hip *h, wf_hip_table_id which, const void **out);
float wf_hip_gravity(const wf_hip *h, float seconds); /* get_gravity(), src/source.hpp:301-312 */
/* vertex fill (cfg.vertices; WF_HIP_OUT_VERTICES / WF_HIP_OUT_VERTEX_COUNTS): vertices per displayed channel, 0 when cfg.vertices is off */
uint32_t wf_hip_num_vertices(const wf_hip *h);
const float *wf_hip_vertices_device(wf_hip *h); /* [n_streams][display_channels][num_vertices][4], device pointer */

float wf_hip_db_min(void);                            /* DB_MIN, src/source.cpp:43 */

/* ---- measurement ---------------------------------------------------------------------------- */
/* Runs `ticks` ticks back to back, each `hop` frames further into audio that must already be in the rings
 * (delay_frames = first_delay - i*hop; a walk that has reached the newest sample starts over at first_delay: the same work
 * per tick on the same resident audio, with no host synchronisation in between), and returns the average device time per
 * tick in milliseconds, measured with hipEvents on the handle's stream around all of the ticks' launches. */
int wf_hip_time_ticks(wf_hip *h, const wf_hip_tick_params *p, uint32_t ticks, uint32_t hop, float *avg_kernel_ms);
/* The same measurement around calls of the host's choosing: wf_hip_time_begin records a hipEvent on the handle's stream,
 * wf_hip_time_end joins everything issued since (ticks on every lane, copies), records the second event, waits for it and
 * returns the elapsed device time in milliseconds. */
int wf_hip_time_begin(wf_hip *h);
int wf_hip_time_end(wf_hip *h, float *elapsed_ms);
const char *wf_hip_kernel_name(const wf_hip *h);
/* launches of the fused kernel one wf_hip_tick issues (lanes: slices of the batch on their own HIP streams, running
 * concurrently; a profiler's per-launch average is then not the time a tick takes) */
uint32_t wf_hip_launches_per_tick(const wf_hip *h);
/* algorithmic HBM bytes one tick moves: per spectrum 4N in + state r/w + dB out */
uint64_t wf_hip_algorithmic_bytes_per_tick(const wf_hip *h, uint32_t flags);


/* ---- one batch over several devices (BASELINE.json configs[4]) ----------------------------------
 * The reference has nothing distributed: its sources share nothing (one WAVSource per OBS source, src/source.cpp:87-102), so
 * a batch shards embarrassingly.  A wf_hip_multi is ONE batch of `streams_total` streams split into contiguous shards over
 * n devices of one node -- shard i = streams [first_i, first_i + count_i), sizes differing by at most one, the first
 * (streams_total % n) shards holding one more -- each shard a plain wf_hip handle on its device, driven by its own host
 * thread (single process; the calls below fan out to the threads and return when every device has *enqueued* its part, as
 * wf_hip_tick does).  There is no collective on the data path.  The one exchange is the result the north star asks for:
 * wf_hip_multi_allgather_bars leaves the bar heights of ALL streams on EVERY device ([streams_total][display_channels]
 * [num_bars] floats, global stream order) for a combined render -- by ncclAllGather of a dlopen()ed librccl.so over xGMI
 * (transport "rccl"; shards of unequal size are padded to the largest and compacted on the device), or, where RCCL is
 * absent or refuses the device list (e.g. the same device named twice, which this library allows), by direct peer copies
 * (hipMemcpyPeerAsync, transport "peer"); WF_HIP_MULTI_TRANSPORT=rccl|peer forces one.  The gather is asynchronous and
 * double-buffered: it is enqueued on a side stream of every device behind the ticks issued so far, the next tick runs
 * meanwhile, and a result stays valid until the FIRST TICK AFTER THE NEXT GATHER (where the tick kernels write the send
 * buffers -- or, one device / peer access everywhere, the results -- themselves, that tick's kernels rewrite the buffer; the
 * copying paths keep it until the second-next gather, but no caller should count on which path a size takes).  Ticks between
 * two gathers, on the group or on a shard handle, and ticks that fail on one shard do not move the buffers: the pair is
 * switched by the gather, on every shard together.  wf_hip_set_bars_mirrors / wf_hip_bars_mirror_ready must not be called on a
 * shard handle (the group owns its shards' mirror buffers; a gather that finds the shards disagreeing fails).
 * A gather that fails on one device (the others may have their half in flight) takes the exchange out of service for the
 * group: the failing call returns the error, the RCCL communicators are aborted (ncclCommAbort -- no device keeps waiting for
 * a rank that never joined), every later gather returns WF_HIP_ERR_RUNTIME, and everything else -- ticks, reads,
 * wf_hip_multi_sync, wf_hip_multi_destroy -- keeps working.
 * A wf_hip_multi is used by one thread at a time, like a handle.  Shard handles may be used directly (shard-local stream
 * indices) between multi calls -- every wf_hip_* entry point works on them. */
typedef struct wf_hip_multi wf_hip_multi;
int wf_hip_multi_create(const wf_config *cfg, const int *devices, uint32_t n_devices, uint32_t streams_total, uint32_t ring_frames,
                        wf_hip_multi **out);
void wf_hip_multi_destroy(wf_hip_multi *m);
/* text of the last error on this group (or of the last failed wf_hip_multi_create when m == NULL).  Right after a successful
 * wf_hip_multi_create: why the group did not get the transport it would have picked by itself ("ncclCommInitAll failed: ...",
 * "peer access is not enabled between every pair of devices ..."), or "" */
const char *wf_hip_multi_last_error(const wf_hip_multi *m);
uint32_t wf_hip_multi_num_devices(const wf_hip_multi *m);
uint32_t wf_hip_multi_num_streams(const wf_hip_multi *m);
/* "rccl", "peer" or "local" (one shard: the gather is a device copy) */
const char *wf_hip_multi_transport(const wf_hip_multi *m);
/* shard i: its handle, its HIP device, its first global stream and its stream count (any out pointer may be NULL) */
wf_hip *wf_hip_multi_shard(wf_hip_multi *m, uint32_t i, int *device, uint32_t *first, uint32_t *count);
/* wf_hip_push_audio / wf_hip_push_synth / wf_hip_set_hidden / wf_hip_reset with global stream indices (a range may span
 * shards; stream s of wf_hip_multi_push_synth receives wf_synth_noise(seed, stream_id0 + s, ..): the same audio whatever
 * the number of devices) */
int wf_hip_multi_push_audio(wf_hip_multi *m, uint32_t first, uint32_t count, const float *samples, uint32_t frames);
int wf_hip_multi_push_synth(wf_hip_multi *m, uint32_t first, uint32_t count, uint64_t seed, uint32_t stream_id0, uint64_t index0, uint32_t frames);
int wf_hip_multi_set_hidden(wf_hip_multi *m, uint32_t first, uint32_t count, const uint8_t *mask);
int wf_hip_multi_reset(wf_hip_multi *m, uint32_t first, uint32_t count);
/* WAVSource*::tick_spectrum for every stream of every shard (wf_hip_tick on each device, issued concurrently by the
 * devices' host threads) */
int wf_hip_multi_tick(wf_hip_multi *m, const wf_hip_tick_params *p);
int wf_hip_multi_sync(wf_hip_multi *m); /* every device's streams, the gather streams included */
/* results with global stream indices, as wf_hip_read (any output of the batch) */
int wf_hip_multi_read(wf_hip_multi *m, wf_hip_output what, uint32_t first, uint32_t count, void *out);
/* the exchange (see above); needs cfg.bars or cfg.curve */
int wf_hip_multi_allgather_bars(wf_hip_multi *m);
/* device i's copy of the newest gathered result: a pointer on that device (valid until the first tick after the next gather; ordered behind
 * the gather on wf_hip_multi_gather_stream(m, i)), or copied to the host after waiting for it */
const float *wf_hip_multi_gathered_device(wf_hip_multi *m, uint32_t i);
void *wf_hip_multi_gather_stream(wf_hip_multi *m, uint32_t i); /* hipStream_t */
int wf_hip_multi_read_gathered(wf_hip_multi *m, uint32_t i, float *out);
/* measurement: `ticks` ticks back to back on every device as wf_hip_time_ticks does (each device's host thread runs its own
 * loop), with an all-gather of the bars behind every tick when gather != 0; returns the largest per-device average device
 * time per tick (ms) and, in per_device_ms[n_devices] when not NULL, each device's own */
int wf_hip_multi_time_ticks(wf_hip_multi *m, const wf_hip_tick_params *p, uint32_t ticks, uint32_t hop, int gather, float *avg_ms,
                            float *per_device_ms);


/* ---- test aids (development builds only: -DWF_DEV_BUILD, libwaveform_hip_dev.so; the release library exports neither) ---- */
#ifdef WF_DEV_BUILD
/* Moves the 32-bit sample counters of streams [first, first+count) on by `frames` (a multiple of the ring capacity), as
 * if that much audio had been captured before what the rings hold: tests reach the 2^32-sample wrap-around (a day at
 * 48 kHz; the reference's deques have no such counter) without feeding a day of audio. */
int wf_hip_debug_age(wf_hip *h, uint32_t first, uint32_t count, uint32_t frames);
/* Shard `shard` of the group reports a failure in its half of the next gather (wf_hip_multi_allgather_bars or a gathering
 * wf_hip_multi_time_ticks) before it enqueues anything -- what a failed wait or copy on one device looks like to the others,
 * which have their collective / copies in flight by then.  Tests use it to check that the group survives: the call returns
 * the error, the communicators are aborted, later gathers are refused, ticks, reads, sync and destroy go on working. */
int wf_hip_multi_debug_fail_next_gather(wf_hip_multi *m, uint32_t shard);
#endif

#ifdef __cplusplus
}
#endif
#endif
