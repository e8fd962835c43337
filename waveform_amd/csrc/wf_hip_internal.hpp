// wf_hip_internal.hpp -- what the translation units of libwaveform_hip.so share: the handle, the error helpers and the
// functions by which the plan (wf_hip_plan.hip), the entry points (wf_hip.hip) and the kernel dispatch (wf_tick_geom.hip, one
// object per FFT geometry; wf_big_dispatch.hip for the transforms beyond a CU's LDS) call each other.  Not installed: the
// library's interface is include/wf_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "wf_hip.h"
#include "wf_dev_guard.hpp"
#include "wf_host_tables.hpp"
#include "wf_tick_phases.hpp" // TickArgs, BarsOnlyState (plain structs: no kernel is instantiated by including it)

struct wf_hip {
    wf_config cfg{};
    wf::HostTables tab;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_bars_lane[4] = {nullptr, nullptr, nullptr, nullptr}; // wf_hip_copy_bars_device_async: a lane's part of the copy has been made
    // Lanes: a large batch is ticked as n_lanes slices of streams, slice 0 on `stream`, the others on their own HIP streams.
    // Consecutive ticks of a slice are ordered by its stream; slices share nothing, so while no other call intervenes the
    // tail of one slice's launch overlaps the head of another's (a lone launch leaves the chip draining for a workgroup's
    // lifetime at both ends).  Every other entry point first makes `stream` wait for the lanes (join_lanes) and the next
    // tick makes the lanes wait for `stream`: outside wf_hip_tick the handle behaves as if it had the one stream.
    static constexpr int MAX_LANES = 4;
    int n_lanes = 1;
    uint32_t wg_lds = 0, wg_threads = 0; // dynamic LDS and threads of one workgroup of the tick kernel (how many fit a CU)
    hipStream_t lane_stream[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_lane[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr}, ev_fork = nullptr;
    bool lanes_pending = false; // a lane holds work `stream` has not waited for
    bool main_dirty = true;     // `stream` holds work the lanes have not waited for
    // pipelined ingest (wf_hip_push_audio_async): a copy stream, per-slot staging blocks and events
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr};   // the H2D copy of the slot has finished (host buffer free, staging full)
    hipEvent_t ev_consumed[2] = {nullptr, nullptr}; // the ring append that read the slot's staging block has finished
    float *d_stage_async[2] = {nullptr, nullptr};
    size_t stage_async_floats[2] = {0, 0};
    bool slot_used[2] = {false, false};
    // pipelined readback (wf_hip_read_bars_async): a stream for the D2H copies, a device snapshot and two events per slot
    hipStream_t read_stream = nullptr;
    hipEvent_t ev_snap[2] = {nullptr, nullptr}, ev_read[2] = {nullptr, nullptr};
    float *d_snap[2] = {nullptr, nullptr};
    size_t snap_floats[2] = {0, 0};
    bool read_used[2] = {false, false};
    bool rows_in_flight[2] = {false, false}; // wf_hip_read_rows_async copies straight from m_decibels: the next tick waits for them
    uint32_t *d_frames_async[2] = {nullptr, nullptr}; // ragged ingest: per-stream frame counts of the slot
    uint32_t *h_frames_async[2] = {nullptr, nullptr}; // (page-locked host copy)
    size_t frames_async_cap[2] = {0, 0};
    uint8_t *d_silent_bytes[2] = {nullptr, nullptr};  // rows readback: m_last_silent as bytes
    size_t silent_bytes_cap[2] = {0, 0};
    uint32_t n_streams = 0;
    uint32_t ring_cap = 0;
    uint32_t ring_stride = 0;        // floats between consecutive rings: ring_cap + padding (see wf_hip_create)
    uint32_t N = 0, M = 0;
    uint32_t cap_ch = 1, out_ch = 1, disp_ch = 1;
    uint32_t num_bars = 0;
    bool all_aligned = true; // every push so far was a multiple of 4 frames
    // device memory
    float *d_ring = nullptr;
    uint32_t *d_wpos = nullptr;
    float *d_tsmooth = nullptr;
    float *d_decibels = nullptr;
    uint32_t *d_flags = nullptr;     // [flag_bufs][n_streams]; the buffer flag_cur holds the current m_last_silent / hidden bits
    uint32_t *d_verdict = nullptr;   // split mode: [3][n_streams * cap_ch] "row has a value > floor - 10" (TickArgs::verdict_*)
    uint32_t flag_bufs = 1, flag_cur = 0;
    // bars-only ticks on a batch that does not run split: per-wavefront row verdicts (TickArgs::row_verdict), allocated by the
    // first tick that carries WF_HIP_TICK_NO_DECIBELS; from the tick after it the silence test reads them instead of the rows
    uint32_t *d_row_verdict = nullptr;
    float *d_stale_row = nullptr;    // [M] of DB_MIN (BarsOnlyState::stale_row), allocated with the first bars-only tick
    wf::BarsOnlyState *d_bars_only = nullptr; // the kernel's view of the three fields above
    uint32_t waves_per_spectrum = 1;
    bool verdict_tracking = false;
    bool split = false;              // the channels of a stream run in different workgroups (spectrum_tick_kernel<.., SPLIT>)
    bool split_mono = false;         // ... and, for mono mixdown, in different launches (TickArgs::split_ch)
    // FFT sizes that are not powers of two: Bluestein over the geometry of geom_n = 2 * L points (spectrum_tick_kernel<.., BLU>)
    bool blu = false;
    int mr_half = 0, mr_s3 = 0, mr_lds_cf = 0; // MrPlan::half / s3 / lds_cf: the spectrum's exchange buffer sized by the transform
    int mr_passes = 0;               // > 0: fft_size = 2^a 3^b 5^c, the transform runs as mixed-radix passes inside the Bluestein instantiation (wf_mixed.hpp)
    int mr_radix[4] = {0, 0, 0, 0}, mr_tw_off[4] = {0, 0, 0, 0};
    wf::cf *d_mr_tw = nullptr;       // the passes' twiddle tables (wf::build_mixed_radix_tables)
    wf::cf *d_mr_wp = nullptr;       // W_p^m of a prime first pass (wf::build_prime_twiddles)
    uint32_t geom_n = 0;             // the fft size whose geometry runs the batch (N itself for the power-of-two sizes >= 1024)
    wf::cf *d_blu_a = nullptr, *d_blu_b = nullptr, *d_blu_q = nullptr, *d_blu_qr = nullptr, *d_blu_w = nullptr;
    // transforms beyond a CU's LDS (wf_big.hpp): big_l = big_rows * 16384 complex points in two steps through device memory
    uint32_t big_l = 0, big_rows = 0;
    bool big_mr = false;             // fft sizes above 16384 with small prime factors: big_rows rows of a mixed-radix transform (big_mr_rows_kernel)
    bool big_mrw = false;            // ... two rows on 512 threads: both rows and the end of the tick in one kernel (big_mr_whole_kernel), no scratch
    bool big_br = false;             // fft sizes above 16384 with a prime factor no plan takes: big_rows (= 8) rows, each by Bluestein inside LDS (big_br_rows_kernel)
    uint32_t br_rs = 0;              // ... a row's stride in the scratch buffer: M / big_rows rounded up to even
    uint32_t br_l = 0;               // ... over br_l complex points (4096 / 8192: the 8192- / 16384-sample geometry as container)
    wf::cf *d_br_tw1 = nullptr, *d_br_tw2 = nullptr; // the container's pass-1 / pass-2 twiddles
    wf::cf *d_br_rowtw = nullptr, *d_br_bhat = nullptr, *d_br_q = nullptr; // build_bluestein_rows
    wf::cf *d_big_wc = nullptr;      // [8][8] W_big_rows^(c k1)
    bool big_whole = false;          // fft_size 65536: both rows plus the end of the tick in ONE kernel (big_whole_kernel), nothing through device memory
    wf::cf *d_big_v = nullptr, *d_big_z = nullptr, *d_big_tw = nullptr, *d_big_tws = nullptr;
    uint32_t *d_big_nz = nullptr;
    size_t big_out_lds = 0;          // dynamic LDS of big_outputs_kernel
    int *d_big_task = nullptr, *d_big_bar_task = nullptr; // BarArgs::big_task / big_bar_task
    int big_num_tasks = 0;
    int mr_plan_id = 0;              // spectrum_tick_kernel's PLAN: the compile-time mixed-radix plan of this size (0: the run-time plan)
    int interp_shape[2] = {0, 0};    // {tab.interp_radius, tab.interp_taps} (WF_HIP_TABLE_INTERP_SHAPE)
    float *d_bars = nullptr;
    float *d_bars_pre = nullptr;    // BarArgs::pre_out: [n_streams][disp_ch], mirrored displays only
    // wf_hip_set_bars_mirrors: the caller-owned buffers the ticks also write their bars into -- the mirror_n buffers of the write
    // set; wf_hip_bars_mirror_ready hands the write set over and makes the other one the write set
    float *bars_mirror[2][8] = {};
    uint32_t mirror_n = 0;
    uint32_t mirror_next = 0;       // the set the ticks write
    bool mirror_fresh = false;      // a tick has written set mirror_next since it became the write set
    wf::VertexTables vtab;           // cfg.vertices: the vertex fill behind every tick
    wf::f4 *d_verts = nullptr;
    uint32_t *d_vert_counts = nullptr; // [n_streams][disp_ch] vertices of each row's draw call
    float *d_cap_xy = nullptr;
    float *d_window = nullptr, *d_slope = nullptr, *d_rolloff = nullptr;
    wf::cf *d_tw1 = nullptr, *d_tw2 = nullptr, *d_tws = nullptr;
    float *d_bar_coef = nullptr;
    int *d_bar_bin = nullptr, *d_bar_off = nullptr, *d_band_widths = nullptr, *d_bar_chunk = nullptr;
    int bar_chunks = 0, bar_lpb = 1, bar_segs = 0;
    int bar_blocks = 0;
    int bar_stage_off = 0;           // BarArgs::stage_off
    uint32_t *d_delay = nullptr;     // [n_streams] A/V-sync delay per stream (wf_hip_set_stream_delay), or nullptr
    uint32_t max_stream_delay = 0;   // largest value ever set (ring-capacity check of the tick)
    unsigned long long *d_audio_ts = nullptr; // [n_streams] m_audio_ts per stream of a waveform batch (wf_hip_set_stream_audio_ts), or nullptr
    bool stream_delays_aligned = true; // all of them multiples of 4 frames (vector fetch without straddling)
    float *d_vol_comp = nullptr;     // [n_streams] volume compensation per stream (wf_hip_set_input_rms), or nullptr
    // volume-normalisation producer on the device (wf_hip_enable_input_rms): update_input_rms per stream and tick
    float *d_rms_ring = nullptr;     // [n_streams][rms_cap] squared peaks (capture_audio's m_rms_sync_buf)
    float *d_rms_bsum = nullptr;     // [n_streams][rms_cap / RMS_BLOCK]
    uint32_t *d_rend = nullptr;      // [n_streams] consumption point of sync_rms_buffer
    bool rms_feed = false;           // the squared peaks come from the host (wf_hip_push_rms_ragged_async), not from the pushed audio
    float *d_sq_stage[2] = {nullptr, nullptr};      // feed staging per ingest slot: [count][max_frames] squared peaks ...
    size_t sq_stage_floats[2] = {0, 0};
    uint32_t *d_sq_frames[2] = {nullptr, nullptr};  // ... and their counts
    uint32_t *h_sq_frames[2] = {nullptr, nullptr};  // page-locked copy the H2D reads from
    size_t sq_frames_cap[2] = {0, 0};
    hipEvent_t ev_sq_consumed[2] = {nullptr, nullptr};
    bool sq_slot_used[2] = {false, false};
    float *d_input_rms = nullptr;    // [n_streams] m_input_rms
    uint32_t rms_cap = 0, rms_size = 0;
    // waveform batches (cfg.waveform): N = M = width (points per row), there is no FFT state
    bool wave = false;
    uint32_t wave_samples = 0;       // m_waveform_samples
    uint32_t *d_cend = nullptr;      // [n_streams] samples consumed so far
    unsigned long long *d_wts = nullptr; // [n_streams] m_waveform_ts
    // level-meter batches (cfg.meter): N is the meter buffer length, there is no FFT state
    bool meter = false;
    uint32_t *d_mend = nullptr;      // [n_streams] consumption point of tick_meter
    float *d_meter_buf = nullptr;    // [n_streams * cap_ch] m_meter_buf
    float *d_meter_val = nullptr;    // [n_streams * cap_ch] m_meter_val
    // The device copies of the window (and, for Bluestein, chirped-window) tables carry a power-of-two factor and the magnitude
    // coefficient its inverse: scaling by 2^k is exact, the transform is linear, and |X|^2 = re^2 + im^2 -- the one place where
    // the path squares -- then stays representable down to |X| ~ 1e-31 instead of ~1e-19 (hypotf in the reference answers for
    // the whole float range: the first ticks behind a reset through a narrow window, a few samples under sin^16 tails, give
    // |X| ~ 1e-26).  Headroom: N * amplitude * in_scale squared must stay below FLT_MAX; the factor is 2^40 up to 4096 samples and
    // halves with every doubling beyond (wf_hip_create), which keeps the overflow point at an amplitude of 4096 (+72 dBFS) from
    // 4096 samples up (the reference's hypotf overflows far later still: a stated deviation).  Bluestein
    // through device memory squares values that still carry its factor L: 2^24 there.
    float in_scale = 1.0f;
    bool ext_outputs = false;        // the outputs are derived from the stored rows by big_outputs_kernel behind the tick kernel
                                     // (displays whose staging does not fit the tick kernel's exchange buffer)
    bool curve = false;              // the outputs are curve points (render_curve), not bars
    bool curve_both = false;         // ... finished by the threads of both spectra of a workgroup (mono mixdown)
    bool curve_catrom = false;       // ... Catmull-Rom: positions only, weights on the device (BarArgs::cur_x)
    bool stream_steps = false;       // ... more points per thread than OutVals holds (BarArgs::stream_steps)
    float *d_cur_x = nullptr;
    int out_steps = 0;               // outputs finished per thread (curve: ceil(width / T); bars in segment form: 1)
    float *d_cur_coef = nullptr, *d_gauss = nullptr, *d_gauss_wsum = nullptr;
    int *d_cur_base = nullptr;
    float *d_lane_coef = nullptr;
    int *d_lane_base = nullptr, *d_bar_seg = nullptr, *d_seg_group = nullptr, *d_lead_bar = nullptr, *d_lead_end = nullptr;
    bool bar_wave_local = false;
    bool bar_piece_mode = false;     // BarArgs::piece_mode (d_seg_group holds BarPieceTables::info, d_bar_seg its bar_piece)
    float *d_ps_tab = nullptr;       // BarArgs::ps_tab (BarPsTables::tab), prefix-sum layout of the bar reduction
    int bar_ps_lanes = 0;            // BarArgs::ps_lanes; 0: the layout is not used
    unsigned long long *d_phase_clock = nullptr; // only allocated by WF_PHASE_TIMING builds
    uint8_t *d_mask = nullptr;
    size_t mask_bytes = 0;
    float *d_stage = nullptr;
    size_t stage_floats = 0;
    // per-stream words the host sets every video frame (wf_hip_set_stream_delay / _audio_ts): staged in page-locked memory of
    // the handle's own, two blocks used alternately, so that the calls copy and return instead of draining the stream
    void *h_words[2] = {nullptr, nullptr};
    size_t h_words_bytes[2] = {0, 0};
    hipEvent_t ev_words[2] = {nullptr, nullptr};
    uint32_t words_next = 0;
    std::vector<void *> allocs;
    bool canary = false;                                  // WF_HIP_CANARY=1 at create: guard bytes behind every block, checked by wf_hip_sync
    std::vector<std::pair<void *, size_t>> guards;        // (block, payload bytes) of every guarded block still alive
    std::string last_error;
    std::string kernel_name;
    // launch description, fixed at create
    void (*launch)(wf_hip *, const wf::TickArgs &, bool aligned) = nullptr;
    hipStream_t launch_stream = nullptr; // where `launch` enqueues (the lane's stream, set by wf_hip_tick)
    int launch_rc = 0;                   // status of the last `launch` that can fail before its kernels (the big path's memset)
};

namespace wf::host {

// text of the last failed wf_hip_create of this thread (wf_hip_last_error(NULL))
extern thread_local std::string g_create_error;
// records the message on the handle (or, h == nullptr, as the create error) and returns `code`
int fail(wf_hip *h, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

#define WF_TRY_RC(expr)                 \
    do {                                \
        const int rc_ = (expr);         \
        if(rc_ != WF_HIP_OK)            \
            return rc_;                 \
    } while(0)

#define WF_HIP_TRY(h, expr)                                                                                       \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if(e_ != hipSuccess)                                                                                      \
            return fail((h), WF_HIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                                \
    } while(0)

// Every device block carries 256 bytes of slack behind its payload (some kernels read -- never write -- a few words past a row).
// With WF_HIP_CANARY=1 in the environment of wf_hip_create the slack is a guard: filled with GUARD_BYTE when the block is made,
// checked by wf_hip_sync (check_canaries): a kernel that wrote past its buffer turns the next sync into WF_HIP_ERR_RUNTIME
// naming the block (there is no compute-sanitizer on this stack).
constexpr size_t GUARD_BYTES = 256;
constexpr int GUARD_BYTE = 0xA5;
int guard_block(wf_hip *h, void *p, size_t payload_bytes); // wf_hip_plan.hip
int check_canaries(wf_hip *h);                             // wf_hip_plan.hip

template<class T> int dev_alloc(wf_hip *h, T **out, size_t count)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T) + GUARD_BYTES);
    if(e != hipSuccess)
        return fail(h, WF_HIP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    h->allocs.push_back(p);
    *out = static_cast<T *>(p);
    if(h->canary)
        return guard_block(h, p, count * sizeof(T));
    return WF_HIP_OK;
}

template<class T> int upload(wf_hip *h, T **out, const std::vector<T> &v)
{
    *out = nullptr;
    if(v.empty())
        return WF_HIP_OK;
    int rc = dev_alloc(h, out, v.size());
    if(rc)
        return rc;
    WF_HIP_TRY(h, hipMemcpyAsync(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return WF_HIP_OK;
}

inline uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while(p < v)
        p <<= 1;
    return p;
}

// `bytes` of per-stream words from a borrowed host array to `d_dst`, without waiting for the stream: through one of the handle's
// two page-locked staging blocks (wf_hip.hip)
int upload_words(wf_hip *h, void *d_dst, const void *src, size_t bytes);

// ---- kernel dispatch -----------------------------------------------------------------------------------------------------
// One object file per geometry (wf_tick_geom.hip compiled with -DWF_TU_GEOM=<N>): picks the spectrum_tick_kernel instantiation
// of this handle's configuration (plain / staged tables / shared curve row / split / zero-padded / Bluestein / mixed radix),
// sets its dynamic-LDS attribute and leaves h->launch, h->wg_lds, h->wg_threads, h->split, h->flag_bufs, h->kernel_name.
int setup_tick_512(wf_hip *h, bool want_split);
int setup_tick_1024(wf_hip *h, bool want_split);
int setup_tick_2048(wf_hip *h, bool want_split);
int setup_tick_4096(wf_hip *h, bool want_split);
int setup_tick_8192(wf_hip *h, bool want_split);
int setup_tick_16384(wf_hip *h, bool want_split);
int setup_tick_32768(wf_hip *h, bool want_split);
// wf_big_dispatch.hip: fft sizes whose transform does not fit a CU's LDS, and big_outputs_kernel for the displays that are
// finished behind the tick kernel (h->ext_outputs)
int setup_launch_big(wf_hip *h);
int big_outputs_set_lds(wf_hip *h);
void big_outputs_launch(wf_hip *h, const wf::TickArgs &a, uint32_t rows, hipStream_t st);

} // namespace wf::host
