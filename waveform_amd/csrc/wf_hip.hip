// wf_hip.hip -- implementation of the C ABI in include/wf_hip.h (host side + kernel launches).
// gfx950 only.  There is no CPU fallback: every entry point either drives the device or fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "wf_hip.h"
#include "wf_host_tables.hpp"
#include "wf_kernels.hpp"
#include "wf_big.hpp"
#include "wf_meter.hpp"
#include "wf_rms.hpp"
#include "wf_wave.hpp"
#include "wf_vertex.hpp"

namespace {

thread_local std::string g_create_error;

} // namespace

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
    int mr_passes = 0;               // > 0: fft_size = 2^a 3^b 5^c, the transform runs as mixed-radix passes inside the Bluestein instantiation (wf_mixed.hpp)
    int mr_radix[4] = {0, 0, 0, 0}, mr_tw_off[4] = {0, 0, 0, 0};
    wf::cf *d_mr_tw = nullptr;       // the passes' twiddle tables (wf::build_mixed_radix_tables)
    wf::cf *d_mr_wp = nullptr;       // W_p^m of a prime first pass (wf::build_prime_twiddles)
    uint32_t geom_n = 0;             // the fft size whose geometry runs the batch (N itself for the power-of-two sizes >= 1024)
    wf::cf *d_blu_a = nullptr, *d_blu_b = nullptr, *d_blu_q = nullptr, *d_blu_qr = nullptr, *d_blu_w = nullptr;
    // transforms beyond a CU's LDS (wf_big.hpp): big_l = big_rows * 16384 complex points in two steps through device memory
    uint32_t big_l = 0, big_rows = 0;
    bool big_mr = false;             // fft sizes above 16384 with small prime factors: big_rows rows of a mixed-radix transform (big_mr_rows_kernel)
    wf::cf *d_big_wc = nullptr;      // [8][8] W_big_rows^(c k1)
    bool big_fused = false;          // fft_size 65536: column step and real split folded into the rows kernel (big_rows_fold_kernel)
    float *d_big_mag = nullptr;      // [n_spec][2][16384] its output: magnitudes by bin parity
    wf::cf *d_big_v = nullptr, *d_big_z = nullptr, *d_big_tw = nullptr, *d_big_tws = nullptr;
    uint32_t *d_big_nz = nullptr;
    size_t big_out_lds = 0;          // dynamic LDS of big_outputs_kernel
    int *d_big_task = nullptr, *d_big_bar_task = nullptr; // BarArgs::big_task / big_bar_task
    int big_num_tasks = 0;
    float *d_bars = nullptr;
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
    // |X| ~ 1e-26).  Headroom: N * amplitude * 2^40 squared must stay below FLT_MAX -- amplitude < 256 at N = 65536, < 4000 at
    // N = 4096 (+48 dBFS and more; the reference overflows 2^40 times later).  Bluestein through device memory squares values
    // that still carry its factor L: 2^24 there.
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
    unsigned long long *d_phase_clock = nullptr; // only allocated by WF_PHASE_TIMING builds
    uint8_t *d_mask = nullptr;
    size_t mask_bytes = 0;
    float *d_stage = nullptr;
    size_t stage_floats = 0;
    std::vector<void *> allocs;
    std::string last_error;
    std::string kernel_name;
    // launch description, fixed at create
    void (*launch)(wf_hip *, const wf::TickArgs &, bool aligned) = nullptr;
    hipStream_t launch_stream = nullptr; // where `launch` enqueues (the lane's stream, set by wf_hip_tick)
    int launch_rc = 0;                   // status of the last `launch` that can fail before its kernels (the big path's memset)
};

namespace {

int fail(wf_hip *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if(h)
        h->last_error = buf;
    else
        g_create_error = buf;
    return code;
}

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

template<class T> int dev_alloc(wf_hip *h, T **out, size_t count)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T) + 256);
    if(e != hipSuccess)
        return fail(h, WF_HIP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    h->allocs.push_back(p);
    *out = static_cast<T *>(p);
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

template<class G> void launch_tick_split(wf_hip *h, const wf::TickArgs &a0, bool aligned)
{
    const dim3 block(G::T);
    const size_t lds = wf::tick_lds_bytes<G, 1>();
    // mono mixdown: channel 1 of every stream, then channel 0 (TickArgs::split_ch); stereo pairs: everything at once
    for(int pass = 0; pass < (h->split_mono ? 2 : 1); ++pass) {
        wf::TickArgs a = a0;
        a.split_ch = h->split_mono ? (uint32_t)(1 - pass) : 0xffffffffu;
        const dim3 grid(h->split_mono ? a.stream_count : a.stream_count * a.cap_ch);
        if(aligned)
            hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, 1, true, true>), grid, block, lds, h->launch_stream, a);
        else
            hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, 1, false, true>), grid, block, lds, h->launch_stream, a);
    }
}

template<class G> int setup_launch_split(wf_hip *h)
{
    const int lds = (int)wf::tick_lds_bytes<G, 1>();
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, 1, true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, 1, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    h->launch = &launch_tick_split<G>;
    h->wg_lds = (uint32_t)lds;
    h->wg_threads = (uint32_t)G::T;
    h->split = true;
    h->flag_bufs = 3;
    char name[96];
    snprintf(name, sizeof(name), "spectrum_tick_kernel<N=%d,T=%d,R=%dx%dx%d,SPW=1,split>", G::N, G::T, G::R1, G::R2, G::R3);
    h->kernel_name = name;
    return WF_HIP_OK;
}

// FFT sizes 256 / 128 on the 512-point geometry, zero-padded (spectrum_tick_kernel<.., DEC>)
template<class G, int DEC> void launch_tick_dec(wf_hip *h, const wf::TickArgs &a, bool aligned)
{
    const uint32_t n_spec = a.stream_count * a.cap_ch;
    const dim3 grid((n_spec + 1) / 2), block(G::T * 2);
    const size_t lds = wf::tick_lds_bytes<G, 2>();
    if(aligned)
        hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, 2, true, false, DEC>), grid, block, lds, h->launch_stream, a);
    else
        hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, 2, false, false, DEC>), grid, block, lds, h->launch_stream, a);
}

template<class G, int DEC> int setup_launch_dec(wf_hip *h)
{
    const int lds = (int)wf::tick_lds_bytes<G, 2>();
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, 2, true, false, DEC>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, 2, false, false, DEC>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    h->launch = &launch_tick_dec<G, DEC>;
    h->wg_lds = (uint32_t)lds;
    h->wg_threads = (uint32_t)G::T * 2u;
    char name[96];
    snprintf(name, sizeof(name), "spectrum_tick_kernel<N=%d zero-padded to %d,T=%d,R=%dx%dx%d,SPW=2>", G::N >> DEC, G::N, G::T, G::R1, G::R2, G::R3);
    h->kernel_name = name;
    return WF_HIP_OK;
}

// Bluestein path (FFT sizes that are not powers of two): always the scalar fetch
template<class G, int SPW, bool SPLIT, bool MR = false> void launch_tick_blu(wf_hip *h, const wf::TickArgs &a0, bool)
{
    const uint32_t n_spec = a0.stream_count * a0.cap_ch;
    const dim3 block(G::T * SPW);
    const size_t lds = wf::tick_lds_bytes<G, SPW>();
    const bool two = SPLIT && h->split_mono; // mono mixdown in two launches (TickArgs::split_ch)
    for(int pass = 0; pass < (two ? 2 : 1); ++pass) {
        wf::TickArgs a = a0;
        a.split_ch = two ? (uint32_t)(1 - pass) : 0xffffffffu;
        const dim3 grid(two ? a.stream_count : (n_spec + SPW - 1) / SPW);
        hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, SPW, false, SPLIT, 0, false, true, false, MR>), grid, block, lds, h->launch_stream, a);
    }
}

template<class G, int SPW, bool SPLIT, bool MR = false> int setup_launch_blu(wf_hip *h)
{
    if constexpr(!MR && G::N >= 1024) { // (the smallest container a size that is not a power of two ever gets: wf::bluestein_length)
        // sizes with no prime factor above 5 take the same instantiation's fetch and epilogue around a direct transform
        const char *off = std::getenv("WF_HIP_NO_MIXED_RADIX"); // (development: A/B against Bluestein)
        if(!(off && off[0] == '1') && wf::plan_mixed_radix(h->N / 2, (uint32_t)G::T, h->mr_radix, (uint64_t)G::M) > 0) {
            h->mr_passes = wf::plan_mixed_radix(h->N / 2, (uint32_t)G::T, h->mr_radix, (uint64_t)G::M);
            if(const char *e = std::getenv("WF_HIP_MR_PLAN")) { // (development: "25,16" -- another order or split of the same product)
                int r[4] = {0, 0, 0, 0}, n = 0;
                uint64_t prod = 1;
                for(const char *q = e; *q && n < 4;) {
                    r[n] = std::atoi(q);
                    prod *= (uint64_t)std::max(r[n], 1);
                    ++n;
                    while(*q && *q != ',') ++q;
                    if(*q == ',') ++q;
                }
                bool ok = n >= 2 && prod == h->N / 2 && r[n - 1] <= 16 && (h->N / 2) / (uint32_t)r[n - 1] <= (uint32_t)G::T;
                for(int i = 0; i < n; ++i) {
                    const int v = r[i];
                    ok = ok && (v == 2 || v == 3 || v == 4 || v == 5 || v == 6 || v == 8 || v == 9 || v == 10 || v == 12 || v == 15 || v == 16 || v == 7 || v == 11 || v == 13 ||
                                (i == 0 && (v == 20 || v == 25 || v == 17 || v == 19 || v == 23 || v == h->mr_radix[0])));
                }
                if(ok) {
                    h->mr_passes = n;
                    for(int i = 0; i < 4; ++i)
                        h->mr_radix[i] = r[i];
                }
            }
            return setup_launch_blu<G, SPW, SPLIT, true>(h);
        }
    }
    const int lds = (int)wf::tick_lds_bytes<G, SPW>();
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, SPW, false, SPLIT, 0, false, true, false, MR>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    h->launch = &launch_tick_blu<G, SPW, SPLIT, MR>;
    h->wg_lds = (uint32_t)lds;
    h->wg_threads = (uint32_t)(G::T * SPW);
    h->split = SPLIT;
    char name[160];
    if(MR) {
        char rad[48];
        int o = 0;
        for(int i = 0; i < h->mr_passes; ++i)
            o += snprintf(rad + o, sizeof(rad) - (size_t)o, "%s%d", i ? "x" : "", h->mr_radix[i]);
        snprintf(name, sizeof(name), "spectrum_tick_kernel<N=%u: %u complex points as mixed radix %s,T=%d,SPW=%d%s>", h->N, h->N / 2, rad, G::T, SPW,
                 SPLIT ? ",split" : "");
    } else
        snprintf(name, sizeof(name), "spectrum_tick_kernel<N=%u by Bluestein over %d complex points,T=%d,R=%dx%dx%d,SPW=%d%s>", h->N, G::M, G::T,
                 G::R1, G::R2, G::R3, SPW, SPLIT ? ",split" : "");
    h->kernel_name = name;
    return WF_HIP_OK;
}

template<class G, int SPW, bool TLDS, bool BOTH = false> void launch_tick(wf_hip *h, const wf::TickArgs &a, bool aligned)
{
    const uint32_t n_spec = a.stream_count * a.cap_ch;
    const dim3 grid((n_spec + SPW - 1) / SPW), block(G::T * SPW);
    const size_t lds = wf::tick_lds_bytes<G, SPW>();
    if(aligned)
        hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, SPW, true, false, 0, TLDS, false, BOTH>), grid, block, lds, h->launch_stream, a);
    else
        hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, SPW, false, false, 0, TLDS, false, BOTH>), grid, block, lds, h->launch_stream, a);
}

template<class G, int SPW, bool TLDS, bool BOTH = false> int setup_launch_impl(wf_hip *h)
{
    const int lds = (int)wf::tick_lds_bytes<G, SPW>();
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, SPW, true, false, 0, TLDS, false, BOTH>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, SPW, false, false, 0, TLDS, false, BOTH>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    h->launch = &launch_tick<G, SPW, TLDS, BOTH>;
    h->wg_lds = (uint32_t)lds;
    h->wg_threads = (uint32_t)(G::T * SPW);
    char name[112];
    snprintf(name, sizeof(name), "spectrum_tick_kernel<N=%d,T=%d,R=%dx%dx%d,SPW=%d%s%s>", G::N, G::T, G::R1, G::R2, G::R3, SPW,
             TLDS ? ",tables via LDS" : "", BOTH ? ",curve row shared by both spectra" : "");
    h->kernel_name = name;
    return WF_HIP_OK;
}

// Workgroups of two spectra can stage the window / pass-1 twiddle tables in LDS once (spectrum_tick_kernel<.., TLDS>).
// Measured on MI355X (interleaved A/B): N = 1024 63.4 -> 68.7 % of the HBM peak (8-byte table loads, 23 per thread, become
// 8 DMA requests per wavefront), N = 2048 +-1 %, N = 4096 -1.5 % (the extra barrier costs what the halved table traffic
// saves), N = 8192 +1 %: on for the 8-point geometry only.  WF_HIP_TLDS=0/1 overrides (development aid).
template<class G, int SPW> int setup_launch(wf_hip *h)
{
    if constexpr(SPW == 2) {
        bool tlds = G::P <= 8;
        if(const char *e = std::getenv("WF_HIP_TLDS"))
            tlds = e[0] == '1';
        // mono mixdown with a curve display: the kernel whose two spectra share the row (a TLDS override keeps the plain one)
        if(h->curve_both && h->N == (uint32_t)G::N && tlds == (G::P <= 8)) {
            if constexpr(G::P <= 8)
                return setup_launch_impl<G, 2, true, true>(h);
            else
                return setup_launch_impl<G, 2, false, true>(h);
        }
        if(tlds)
            return setup_launch_impl<G, 2, true>(h);
    }
    return setup_launch_impl<G, SPW, false>(h);
}

// FFT sizes whose transform does not fit a CU's LDS (wf_big.hpp): columns -> rows (twice for Bluestein) -> epilogue -> outputs
// fft_size 65536: rows kernel with the column step and the real split folded in, then the epilogue on magnitudes (wf_big.hpp)
int launch_tick_big_fold(wf_hip *h, const wf::TickArgs &a0, bool aligned)
{
    const uint32_t n_spec = a0.stream_count * a0.cap_ch;
    hipStream_t st = h->launch_stream;
    const uint32_t spec_base = a0.stream_base * a0.cap_ch;
    WF_HIP_TRY(h, hipMemsetAsync(h->d_big_nz + spec_base, 0, (size_t)n_spec * sizeof(uint32_t), st));
    const dim3 grow(2u * ((n_spec + 7u) & ~7u)); // (row, spectrum) by XCD: see big_rows_fold_kernel
    const size_t rows_lds = wf::big_rows_lds_bytes<2>();
    if(aligned)
        hipLaunchKernelGGL(wf::big_rows_fold_kernel<true>, grow, dim3(wf::GFold::T), rows_lds, st, a0);
    else
        hipLaunchKernelGGL(wf::big_rows_fold_kernel<false>, grow, dim3(wf::GFold::T), rows_lds, st, a0);
    const uint32_t parts = (h->M + (uint32_t)wf::BIG_TP - 1u) / (uint32_t)wf::BIG_TP;
    // mono mixdown: channel 1 of every stream, then channel 0 (TickArgs::split_ch)
    for(int pass = 0; pass < (h->split_mono ? 2 : 1); ++pass) {
        wf::TickArgs a = a0;
        a.split_ch = h->split_mono ? (uint32_t)(1 - pass) : 0xffffffffu;
        const dim3 grid(parts, h->split_mono ? a.stream_count : n_spec);
        hipLaunchKernelGGL((wf::big_epilogue_kernel<3>), grid, dim3(wf::GBig::T), 0, st, a);
    }
    if(a0.bar.out != nullptr)
        hipLaunchKernelGGL(wf::big_outputs_kernel, dim3(a0.stream_count * a0.bar.disp_ch), dim3(wf::GBig::T), h->big_out_lds, st, a0);
    WF_HIP_TRY(h, hipGetLastError());
    return WF_HIP_OK;
}

template<int L1> int launch_tick_big_l(wf_hip *h, const wf::TickArgs &a0)
{
    const uint32_t n_spec = a0.stream_count * a0.cap_ch;
    hipStream_t st = h->launch_stream;
    wf::BigArgs b{};
    b.ring = a0.ring;
    b.wpos = a0.wpos;
    b.delay_stream = a0.delay_stream;
    b.ring_mask = a0.ring_mask;
    b.ring_stride = a0.ring_stride;
    b.delay = a0.delay;
    b.cap_ch = a0.cap_ch;
    b.n = h->N;
    b.L = h->big_l;
    b.window = a0.window;
    b.blu_a = h->d_blu_a;
    b.blu_b = h->d_blu_b;
    b.tw_big = h->d_big_tw;
    b.tw1 = a0.tw1;
    b.tw2 = a0.tw2;
    b.v = h->d_big_v;
    b.z = h->d_big_z;
    b.nz = h->d_big_nz;
    b.spec_base = a0.stream_base * a0.cap_ch;
    WF_HIP_TRY(h, hipMemsetAsync(h->d_big_nz + b.spec_base, 0, (size_t)n_spec * sizeof(uint32_t), st));
    const dim3 gcol(wf::BIG_L2 / 512u, n_spec), grow(L1, n_spec);
    const size_t rows_lds = wf::big_rows_lds_bytes<L1>();
    if(h->blu) {
        hipLaunchKernelGGL((wf::big_columns_kernel<L1, 1>), gcol, dim3(256), 0, st, b);
        hipLaunchKernelGGL((wf::big_rows_kernel<L1>), grow, dim3(wf::GBig::T), rows_lds, st, b);
        hipLaunchKernelGGL((wf::big_columns_kernel<L1, 2>), gcol, dim3(256), 0, st, b);
        hipLaunchKernelGGL((wf::big_rows_kernel<L1>), grow, dim3(wf::GBig::T), rows_lds, st, b);
    } else {
        hipLaunchKernelGGL((wf::big_columns_kernel<L1, 0>), gcol, dim3(256), 0, st, b);
        hipLaunchKernelGGL((wf::big_rows_kernel<L1>), grow, dim3(wf::GBig::T), rows_lds, st, b);
    }
    const uint32_t parts = (h->M + (uint32_t)wf::BIG_TP - 1u) / (uint32_t)wf::BIG_TP;
    // mono mixdown: channel 1 of every stream, then channel 0 (TickArgs::split_ch)
    for(int pass = 0; pass < (h->split_mono ? 2 : 1); ++pass) {
        wf::TickArgs a = a0;
        a.split_ch = h->split_mono ? (uint32_t)(1 - pass) : 0xffffffffu;
        const dim3 grid(parts, h->split_mono ? a.stream_count : n_spec);
        if(h->blu)
            hipLaunchKernelGGL((wf::big_epilogue_kernel<2>), grid, dim3(wf::GBig::T), 0, st, a);
        else
            hipLaunchKernelGGL((wf::big_epilogue_kernel<1>), grid, dim3(wf::GBig::T), 0, st, a);
    }
    if(a0.bar.out != nullptr)
        hipLaunchKernelGGL(wf::big_outputs_kernel, dim3(a0.stream_count * a0.bar.disp_ch), dim3(wf::GBig::T), h->big_out_lds, st, a0);
    WF_HIP_TRY(h, hipGetLastError());
    return WF_HIP_OK;
}

// fft sizes above 16384 with small prime factors: rows of a mixed-radix transform (column step folded into the fetch), then the
// epilogue of the packed real transform (wf_big.hpp)
int launch_tick_big_mr(wf_hip *h, const wf::TickArgs &a0)
{
    const uint32_t n_spec = a0.stream_count * a0.cap_ch;
    hipStream_t st = h->launch_stream;
    const uint32_t spec_base = a0.stream_base * a0.cap_ch;
    WF_HIP_TRY(h, hipMemsetAsync(h->d_big_nz + spec_base, 0, (size_t)n_spec * sizeof(uint32_t), st));
    const dim3 grow(h->big_rows * ((n_spec + 7u) & ~7u)); // (row, spectrum) by XCD: see big_mr_rows_kernel
    hipLaunchKernelGGL(wf::big_mr_rows_kernel, grow, dim3(wf::GBig::T), (size_t)(wf::GBig::LDS_CF + 128) * sizeof(wf::cf), st, a0);
    const uint32_t parts = (h->M + (uint32_t)wf::BIG_TP - 1u) / (uint32_t)wf::BIG_TP;
    for(int pass = 0; pass < (h->split_mono ? 2 : 1); ++pass) { // mono mixdown: channel 1 of every stream, then channel 0
        wf::TickArgs a = a0;
        a.split_ch = h->split_mono ? (uint32_t)(1 - pass) : 0xffffffffu;
        const dim3 grid(parts, h->split_mono ? a.stream_count : n_spec);
        hipLaunchKernelGGL((wf::big_epilogue_kernel<1>), grid, dim3(wf::GBig::T), 0, st, a);
    }
    if(a0.bar.out != nullptr)
        hipLaunchKernelGGL(wf::big_outputs_kernel, dim3(a0.stream_count * a0.bar.disp_ch), dim3(wf::GBig::T), h->big_out_lds, st, a0);
    WF_HIP_TRY(h, hipGetLastError());
    return WF_HIP_OK;
}

void launch_tick_big(wf_hip *h, const wf::TickArgs &a, bool aligned)
{
    // (a failure leaves its text in last_error and its HIP error sticky: wf_hip_tick's hipGetLastError() behind the launches
    // reports it; launch_rc carries the code for the errors that are not HIP's)
    // the kernels of this path index spectra with blockIdx.y (<= 65535): larger slices go out in parts
    const uint32_t part = 65535u / a.cap_ch;
    for(uint32_t off = 0; off < a.stream_count && h->launch_rc == WF_HIP_OK; off += part) {
        wf::TickArgs s = a;
        s.stream_base = a.stream_base + off;
        s.stream_count = std::min(part, a.stream_count - off);
        if(h->big_mr) {
            h->launch_rc = launch_tick_big_mr(h, s);
            continue;
        }
        if(h->big_fused) {
            h->launch_rc = launch_tick_big_fold(h, s, aligned);
            continue;
        }
        switch(h->big_rows) {
        case 2: h->launch_rc = launch_tick_big_l<2>(h, s); break;
        case 4: h->launch_rc = launch_tick_big_l<4>(h, s); break;
        default: h->launch_rc = launch_tick_big_l<8>(h, s); break;
        }
    }
}

template<int L1> int setup_big_rows(wf_hip *h)
{
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_rows_kernel<L1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)wf::big_rows_lds_bytes<L1>()));
    return WF_HIP_OK;
}

int setup_launch_big(wf_hip *h)
{
    int rc = WF_HIP_OK;
    if(h->big_mr)
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_mr_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)((size_t)(wf::GBig::LDS_CF + 128) * sizeof(wf::cf))));
    else
        rc = h->big_rows == 2 ? setup_big_rows<2>(h) : h->big_rows == 4 ? setup_big_rows<4>(h) : setup_big_rows<8>(h);
    if(rc)
        return rc;
    // fft_size 65536 (the one power of two up here): everything in one kernel.  WF_HIP_BIG_FUSED=0 keeps the three-kernel path
    // (development aid: A/B, and the path every Bluestein size above 16384 takes)
    h->big_fused = !h->blu && !h->big_mr && h->big_rows == 2;
    if(const char *e = std::getenv("WF_HIP_BIG_FUSED"))
        h->big_fused = h->big_fused && e[0] != '0';
    if(h->big_fused) {
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_rows_fold_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)wf::big_rows_lds_bytes<2>()));
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_rows_fold_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)wf::big_rows_lds_bytes<2>()));
    }
    if(h->big_out_lds)
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_outputs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)h->big_out_lds));
    h->launch = &launch_tick_big;
    h->split = true;
    h->flag_bufs = 3;
    char name[200];
    if(h->big_mr) {
        char rad[48];
        int o = 0;
        for(int i = 0; i < h->mr_passes; ++i)
            o += snprintf(rad + o, sizeof(rad) - (size_t)o, "%s%d", i ? "x" : "", h->mr_radix[i]);
        snprintf(name, sizeof(name), "big_mr_rows_kernel + big_epilogue_kernel<N=%u: %u rows of %u complex points as mixed radix %s, column step folded into the fetch>",
                 h->N, h->big_rows, h->M / h->big_rows, rad);
    } else if(h->big_fused)
        snprintf(name, sizeof(name), "big_rows_fold_kernel + big_epilogue_kernel<N=%u: two rows of 16384 complex points, column step and real split folded into the rows>", h->N);
    else if(h->blu)
        snprintf(name, sizeof(name), "big_{columns,rows,epilogue}_kernel<N=%u by Bluestein over %u = %u x 16384 complex points through device memory>",
                 h->N, h->big_l, h->big_rows);
    else
        snprintf(name, sizeof(name), "big_{columns,rows,epilogue}_kernel<N=%u: %u = %u x 16384 complex points through device memory>", h->N,
                 h->big_l, h->big_rows);
    h->kernel_name = name;
    return WF_HIP_OK;
}

uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while(p < v)
        p <<= 1;
    return p;
}

wf::TickArgs make_args(wf_hip *h, const wf_hip_tick_params *p)
{
    wf::TickArgs a{};
    a.ring = h->d_ring;
    a.wpos = h->d_wpos;
    a.ring_cap = h->ring_cap;
    a.ring_stride = h->ring_stride;
    a.ring_mask = h->ring_cap - 1;
    a.delay = p->delay_frames;
    a.delay_stream = h->d_delay;
    a.window = h->d_window;
    a.tw1 = h->d_tw1;
    a.tw2 = h->d_tw2;
    a.tws = h->d_tws;
    a.slope = h->d_slope;
    a.rolloff = h->d_rolloff;
    a.tsmooth = h->d_tsmooth;
    a.decibels = h->d_decibels;
    a.stream_flags = h->d_flags + (size_t)h->flag_cur * h->n_streams;
    if(h->split) {
        const uint32_t nxt = (h->flag_cur + 1) % 3, clr = (h->flag_cur + 2) % 3;
        const size_t n_spec = (size_t)h->n_streams * h->cap_ch;
        a.flags_out = h->d_flags + (size_t)nxt * h->n_streams;
        a.verdict_in = h->d_verdict + (size_t)h->flag_cur * n_spec;
        a.verdict_out = h->d_verdict + (size_t)nxt * n_spec;
        a.verdict_clear = h->d_verdict + (size_t)clr * n_spec;
    }
    // mono mixdown keeps storing its row: the silence quirk adds the stale row to the partner's magnitudes (wf_kernels.hpp)
    const bool mono_mix_rows = !h->cfg.stereo && h->cap_ch > 1;
    a.skip_decibels = ((p->flags & WF_HIP_TICK_NO_DECIBELS) && !mono_mix_rows) ? 1u : 0u;
    a.split_ch = 0xffffffffu;
    a.bars_only = h->d_bars_only;
    a.stale_row = h->d_stale_row;
    a.bar = wf::BarArgs{};
    if(h->d_bars) {
        a.bar.coef = h->d_bar_coef;
        a.bar.bin = h->d_bar_bin;
        a.bar.off = h->d_bar_off;
        a.bar.count = h->d_band_widths;
        a.bar.big_task = h->d_big_task;
        a.bar.big_bar_task = h->d_big_bar_task;
        a.bar.big_num_tasks = h->big_num_tasks;
        a.bar.chunk = h->d_bar_chunk;
        a.bar.num_chunks = h->bar_chunks;
        a.bar.lane_coef = h->d_lane_coef;
        a.bar.lane_base = h->d_lane_base;
        a.bar.bar_seg = h->d_bar_seg;
        a.bar.seg_group = h->d_seg_group;
        a.bar.lead_bar = h->d_lead_bar;
        a.bar.lead_end = h->d_lead_end;
        a.bar.wave_local = h->bar_wave_local ? 1 : 0;
        a.bar.num_segs = h->bar_segs;
        a.bar.lane_blocks = h->bar_blocks;
        a.bar.cur_coef = h->d_cur_coef;
        a.bar.cur_base = h->d_cur_base;
        a.bar.cur_x = h->d_cur_x;
        a.bar.curve = h->curve ? (h->curve_catrom ? 2 : 1) : 0;
        a.bar.stream_steps = h->stream_steps ? 1 : 0;
        a.bar.both_subs = h->curve_both ? 1 : 0;
        a.bar.out_steps = h->out_steps;
        a.bar.gauss = h->d_gauss;
        a.bar.gauss_wsum = h->d_gauss_wsum;
        a.bar.gauss_radius = h->tab.gauss_radius;
        a.bar.stage_off = h->bar_stage_off;
        a.bar.entries = (int)h->tab.bar_coef.size();
        a.bar.lanes_per_bar = h->bar_lpb;
        a.bar.out = h->d_bars;
        a.bar.num_bars = (int)h->num_bars;
        a.bar.mirror = h->cfg.mirror_freq_axis ? 1 : 0;
        a.bar.border_top = h->tab.border_top;
        a.bar.border_bottom = h->tab.border_bottom;
        a.bar.ceiling = (float)h->cfg.ceiling_db;
        a.bar.dbrange = (float)(h->cfg.ceiling_db - h->cfg.floor_db);
        a.bar.inv_dbrange = 1.0f / a.bar.dbrange;
        a.bar.lerp_mixed = ((a.bar.border_top <= 0 && a.bar.border_bottom >= 0) || (a.bar.border_top >= 0 && a.bar.border_bottom <= 0)) ? 1 : 0;
        a.bar.disp_ch = h->disp_ch;
    }
    a.half_coef = 0.5f * (2.0f / h->tab.window_sum); // mag_coefficient (reference src/source_generic.cpp:110), halved: the
                                                     // kernel produces 2X[k] from the real split
    a.row_bins = h->M;
    if(h->blu) {
        a.blu_a = h->d_blu_a;
        a.blu_b = h->d_blu_b;
        a.blu_n = h->N;
        a.blu_q = h->d_blu_q;
        a.blu_qr = h->d_blu_qr;
        a.blu_w = h->d_blu_w;
        a.mr.passes = h->mr_passes;
        for(int i = 0; i < 4; ++i) {
            a.mr.radix[i] = h->mr_radix[i];
            a.mr.tw_off[i] = h->mr_tw_off[i];
        }
        a.mr.tw = h->d_mr_tw;
        a.mr.wp = h->d_mr_wp;
        if(h->big_l) // direct form: |c_k| / L, times mag_coefficient (the packed form's tables carry the 1 / L, and its real split the 1 / 2)
            a.half_coef = (2.0f / h->tab.window_sum) / (float)h->big_l;
    }
    if(h->big_mr) {
        a.mr.passes = h->mr_passes;
        for(int i = 0; i < 4; ++i) {
            a.mr.radix[i] = h->mr_radix[i];
            a.mr.tw_off[i] = h->mr_tw_off[i];
        }
        a.mr.tw = h->d_mr_tw;
        a.mr.wp = h->d_mr_wp;
        a.big_c = h->big_rows;
        a.big_r = h->M / h->big_rows;
        a.big_wc = h->d_big_wc;
    }
    if(h->big_l) {
        a.big_z = h->d_big_z;
        a.big_tws = h->d_big_tws;
        a.big_tw = h->d_big_tw;
        a.big_mag = h->d_big_mag;
        a.big_nz_out = h->d_big_nz;
        a.big_nz = h->d_big_nz;
        a.big_m = h->blu ? 0u : h->N / 2;
        a.big_l = h->big_l;
        a.blu_n = h->N; // the window length the underflow test compares with
    }
    a.half_coef *= 1.0f / h->in_scale; // the window tables on the device carry in_scale
    a.g = wf::gravity_for(h->cfg, p->seconds);
    a.g2 = 1.0f - a.g;
    a.db_min = wf::db_min();
    a.silent_floor = (float)(h->cfg.floor_db - 10);
    a.vol_comp = 0.0f;
    a.n_streams = h->n_streams;
    a.stream_base = 0;
    a.stream_count = h->n_streams;
    a.cap_ch = h->cap_ch;
    a.out_ch = h->out_ch;
    uint32_t mode = 0;
    if(h->cfg.tsmoothing != WF_TSMOOTH_NONE) mode |= wf::WF_MODE_TSMOOTH;
    if(h->cfg.fast_peaks) mode |= wf::WF_MODE_FAST_PEAKS;
    if(h->cfg.stereo) mode |= wf::WF_MODE_STEREO;
    if(!h->tab.slope.empty()) mode |= wf::WF_MODE_SLOPE;
    if(h->d_rolloff) mode |= wf::WF_MODE_ROLLOFF;
    if(!h->tab.window.empty()) mode |= wf::WF_MODE_WINDOW;
    if(!h->cfg.stereo && h->cap_ch > 1) mode |= wf::WF_MODE_MONO_MIX;
    if(h->cfg.normalize_volume) {
        mode |= wf::WF_MODE_NORMALIZE;
        // volume_compensation, reference src/source_generic.cpp:163
        const float rms_db = (p->input_rms > 0.0f) ? 20.0f * std::log10(p->input_rms) : wf::db_min();
        a.vol_comp = std::min(h->cfg.volume_target - rms_db, h->cfg.max_gain);
        a.vol_comp_stream = h->d_vol_comp; // per-stream values once wf_hip_set_input_rms has been used
    }
    a.mode = mode;
    a.phase_clock = h->d_phase_clock;
    return a;
}

wf::MeterArgs make_meter_args(wf_hip *h, const wf_hip_tick_params *p)
{
    wf::MeterArgs m{};
    m.ring = h->d_ring;
    m.wpos = h->d_wpos;
    m.mend = h->d_mend;
    m.ring_cap = h->ring_cap;
    m.ring_stride = h->ring_stride;
    m.ring_mask = h->ring_cap - 1;
    m.delay = p->delay_frames;
    m.delay_stream = h->d_delay;
    m.size = h->N;
    m.meter_buf = h->d_meter_buf;
    m.meter_val = h->d_meter_val;
    m.stream_flags = h->d_flags;
    m.bars = h->d_bars;
    m.g = wf::gravity_for(h->cfg, p->seconds);
    m.g2 = 1.0f - m.g;
    m.db_min = wf::db_min();
    m.silent_floor = (float)(h->cfg.floor_db - 10);
    m.border_top = h->tab.border_top;
    m.border_bottom = h->tab.border_bottom;
    m.ceiling = (float)h->cfg.ceiling_db;
    m.dbrange = (float)(h->cfg.ceiling_db - h->cfg.floor_db);
    m.n_streams = h->n_streams;
    m.cap_ch = h->cap_ch;
    m.rms = h->cfg.meter_rms ? 1u : 0u;
    m.tsmooth = (h->cfg.tsmoothing != WF_TSMOOTH_NONE) ? 1u : 0u;
    m.fast_peaks = h->cfg.fast_peaks ? 1u : 0u;
    return m;
}

// update_input_rms of every stream (what WAVSource::tick does first, src/source.cpp:1330-1331); leaves the per-stream volume
// compensation where the tick kernels read it.  No-op unless wf_hip_enable_input_rms has been called.
void launch_input_rms(wf_hip *h, const wf_hip_tick_params *p)
{
    if(h->d_rms_ring == nullptr)
        return;
    wf::RmsArgs r{};
    r.rms_ring = h->d_rms_ring;
    r.bsum = h->d_rms_bsum;
    r.wpos = h->d_wpos;
    r.rend = h->d_rend;
    r.rms_cap = h->rms_cap;
    r.size = h->rms_size;
    r.delay = p->delay_frames;
    r.delay_stream = h->d_delay;
    r.input_rms = h->d_input_rms;
    r.vol_comp = h->d_vol_comp;
    r.volume_target = h->cfg.volume_target;
    r.max_gain = h->cfg.max_gain;
    r.db_min = wf::db_min();
    r.n_streams = h->n_streams;
    r.feed = h->rms_feed ? 1u : 0u;
    hipLaunchKernelGGL(wf::input_rms_kernel, dim3(h->n_streams), dim3(64), 0, h->stream, r);
}

// Every entry point other than wf_hip_tick: `stream` waits for what the lanes hold, and the next tick's lanes will wait for
// what this call enqueues on `stream`.
int join_lanes(wf_hip *h)
{
    if(h->lanes_pending) {
        WF_HIP_TRY(h, hipSetDevice(h->device));
        for(int l = 1; l < h->n_lanes; ++l)
            WF_HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_lane[l], 0));
        h->lanes_pending = false;
    }
    h->main_dirty = true;
    return WF_HIP_OK;
}

// wf_hip_read_rows_async copies straight out of m_decibels on the readback stream: whatever is about to overwrite rows (a
// tick of a spectrum or waveform batch, wf_hip_reset) first makes `stream` wait -- on the device -- for copies in flight
int wait_rows_in_flight(wf_hip *h)
{
    for(int i = 0; i < 2; ++i)
        if(h->rows_in_flight[i]) {
            WF_HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_read[i], 0));
            h->rows_in_flight[i] = false;
            h->main_dirty = true;
        }
    return WF_HIP_OK;
}

int check_range(wf_hip *h, uint32_t first, uint32_t count)
{
    if(h == nullptr)
        return WF_HIP_ERR_INVALID;
    if(count == 0 || first >= h->n_streams || count > h->n_streams - first)
        return fail(h, WF_HIP_ERR_INVALID, "stream range [%u, %u+%u) outside 0..%u", first, first, count, h->n_streams);
    return join_lanes(h);
}

// frees a block handed out by dev_alloc (the caller has made sure nothing enqueued still uses it)
void dev_release(wf_hip *h, void *p)
{
    if(p == nullptr)
        return;
    for(size_t i = 0; i < h->allocs.size(); ++i)
        if(h->allocs[i] == p) {
            h->allocs[i] = h->allocs.back();
            h->allocs.pop_back();
            break;
        }
    (void)hipFree(p);
}

// staging blocks grow geometrically and the outgrown block is released once the stream has drained it
size_t grown(size_t have, size_t need) { return std::max(need, have + have / 2); }

int ensure_stage(wf_hip *h, size_t floats)
{
    if(h->stage_floats >= floats)
        return WF_HIP_OK;
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream)); // the old block may still feed a ring append
    dev_release(h, h->d_stage);
    h->d_stage = nullptr;
    const size_t want = grown(h->stage_floats, floats);
    h->stage_floats = 0;
    float *p = nullptr;
    int rc = dev_alloc(h, &p, want);
    if(rc)
        return rc;
    h->d_stage = p;
    h->stage_floats = want;
    return WF_HIP_OK;
}

// the squared-peak ring follows the pushed audio (wf_hip_enable_input_rms) -- as opposed to being fed by the host
inline bool rms_follows_audio(const wf_hip *h) { return h->d_rms_ring != nullptr && !h->rms_feed; }

constexpr uint32_t PUSH_SLICE = 16384; // streams per launch of the ingest kernels (rows = streams * cap_ch <= 65535)

// the RMS ring follows every push (before wpos advances): squared peaks, then the sums of the blocks the push completed
void rms_after_push(wf_hip *h, uint32_t first, uint32_t count, uint32_t frames)
{
    hipLaunchKernelGGL(wf::rms_block_kernel, dim3(frames / wf::RMS_BLOCK + 1, count), dim3(64), 0, h->stream, h->d_rms_ring,
                       h->d_rms_bsum, h->d_wpos, h->rms_cap, first, frames);
}

// d_src feeds the audio rings (nullptr: zeros); d_rms_src feeds the squared-peak ring when the producer is enabled
// (capture_audio takes the RMS from the packet even when it is muted, src/source.cpp:1842-1871 vs :1879-1880)
int push_common(wf_hip *h, uint32_t first, uint32_t count, const float *d_src, const float *d_rms_src, uint32_t frames)
{
    if(frames == 0)
        return WF_HIP_OK;
    // a packet longer than the ring keeps its newest ring_cap frames, as CircularBuffer + capture_audio's trimming would
    if(rms_follows_audio(h) && frames > h->rms_cap)
        return fail(h, WF_HIP_ERR_INVALID, "push of %u frames exceeds the RMS ring capacity %u", frames, h->rms_cap);
    // the kernels index (stream, channel) rows by blockIdx.y (at most 65535): larger batches go in slices
    for(uint32_t off = 0; off < count; off += PUSH_SLICE) {
        const uint32_t cnt = std::min(PUSH_SLICE, count - off);
        const size_t skip = (size_t)off * h->cap_ch * frames;
        const dim3 grid((frames + 255) / 256 > 64 ? 64 : (frames + 255) / 256, cnt * h->cap_ch), block(256);
        hipLaunchKernelGGL(wf::ring_push_kernel, grid, block, 0, h->stream, h->d_ring, h->d_wpos, h->ring_cap, h->ring_stride, h->cap_ch,
                           first + off, d_src ? d_src + skip : nullptr, frames);
        if(rms_follows_audio(h)) {
            hipLaunchKernelGGL(wf::rms_push_kernel, dim3(grid.x, cnt), block, 0, h->stream, h->d_rms_ring, h->d_wpos, h->rms_cap,
                               h->cap_ch, first + off, d_rms_src ? d_rms_src + skip : nullptr, frames);
            rms_after_push(h, first + off, cnt, frames);
        }
    }
    hipLaunchKernelGGL(wf::wpos_advance_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->d_wpos,
                       h->d_flags + (size_t)h->flag_cur * h->n_streams, first, count, frames);
    WF_HIP_TRY(h, hipGetLastError());
    if(frames % 4u)
        h->all_aligned = false;
    return WF_HIP_OK;
}

} // namespace

extern "C" {

int wf_hip_abi_version(void) { return WF_HIP_ABI_VERSION; }

int wf_hip_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *wf_hip_last_error(const wf_hip *h) { return h ? h->last_error.c_str() : g_create_error.c_str(); }

int wf_hip_create(const wf_config *cfg, int device, uint32_t max_streams, uint32_t ring_frames, wf_hip **out)
{
    if(out == nullptr)
        return WF_HIP_ERR_INVALID;
    *out = nullptr;
    if(cfg == nullptr || max_streams == 0)
        return fail(nullptr, WF_HIP_ERR_INVALID, "cfg is NULL or max_streams is 0");
    wf::HostTables tab;
    wf_config cfg_eff = *cfg;
    uint32_t wave_samples = 0;
    wf::normalize_config(cfg_eff);
    if(cfg_eff.waveform)
        wave_samples = wf::waveform_config(cfg_eff); // update()'s overrides; fft_size becomes the row length (width)
    else if(cfg_eff.meter)
        wf::meter_config(cfg_eff); // update()'s overrides for the mode; fft_size becomes the meter buffer length
    cfg = &cfg_eff;
    int rc = wf::build_host_tables(*cfg, tab);
    if(rc == WF_HIP_ERR_UNSUPPORTED && cfg->waveform)
        return fail(nullptr, rc, "waveform display: width %u above 8192 points is not implemented", cfg->width);
    if(rc == WF_HIP_ERR_UNSUPPORTED)
        return fail(nullptr, rc, "fft_size %u: implemented is every multiple of 16 from 128 to 65536 (the reference's own range)", cfg->fft_size);
    if(rc)
        return fail(nullptr, rc, "invalid configuration");
    const int ndev = wf_hip_device_count();
    if(ndev <= 0)
        return fail(nullptr, WF_HIP_ERR_NO_DEVICE, "no HIP device available");
    if(device < 0 || device >= ndev)
        return fail(nullptr, WF_HIP_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);

    wf_hip *h = new(std::nothrow) wf_hip();
    if(h == nullptr)
        return fail(nullptr, WF_HIP_ERR_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->tab = std::move(tab);
    h->device = device;
    h->n_streams = max_streams;
    h->N = cfg->fft_size;
    h->M = cfg->fft_size / 2;
    h->cap_ch = cfg->capture_channels;
    h->out_ch = h->tab.output_channels;
    h->disp_ch = h->tab.display_channels;
    h->num_bars = (uint32_t)h->tab.num_bars;
    h->ring_cap = next_pow2(ring_frames ? std::max(ring_frames, h->N) : std::max(2 * h->N, 4096u));
    {
        const uint32_t L = (cfg->meter || cfg->waveform) ? 0u : wf::bluestein_length(cfg->fft_size);
        h->blu = L != 0;
        h->big_l = L > 16384u ? L : (!L && h->N == 65536u) ? 32768u : 0u;
        h->big_rows = h->big_l / 16384u;
        h->geom_n = h->big_l ? 32768u : L ? 2 * L : std::max(h->N, 512u); // big: the row transform's geometry
        // above 16384 samples and not a power of two: where n/2 = C R with R <= 8192 a length that has a mixed-radix plan, C <= 8
        // rows of that transform (big_mr_rows_kernel) instead of Bluestein through device memory
        const char *no_mr = std::getenv("WF_HIP_NO_MIXED_RADIX");
        if(h->blu && h->big_l && !(no_mr && no_mr[0] == '1')) {
            const uint32_t np = h->N / 2;
            for(uint32_t c = 2; c <= 8 && !h->big_mr; ++c) {
                if(np % c || np / c > 8192u)
                    continue;
                const int passes = wf::plan_mixed_radix(np / c, 1024u, h->mr_radix);
                if(passes > 0) {
                    h->big_mr = true;
                    h->mr_passes = passes;
                    h->blu = false;      // no chirp tables, no chirped window: the plain packed real transform
                    h->big_l = np;       // (complex points per spectrum in the scratch buffer)
                    h->big_rows = c;
                }
            }
        }
    }
    if(cfg->waveform) {
        // rows of `width` points; the ring holds the history the points are picked from (+ the width zeros of update())
        h->wave = true;
        h->wave_samples = wave_samples;
        h->M = h->N;
        h->ring_cap = next_pow2(std::max(ring_frames, 2 * (wave_samples + h->N)));
    }
    {
        // Deep rings (a window of fft_size samples somewhere in a row of >= 256 KB) with a power-of-two row stride put every
        // stream's window at the same offset modulo the stride; 64 KB + 256 B of padding per row spreads them over the memory
        // channels: +2.5-4 % on the 1 MB rows of bench.py (60.0-60.3 -> 61.7-63.2 % of peak, three interleaved runs), nothing
        // to gain on shallow rings.  WF_HIP_RING_PAD=<floats> overrides (development aid).
        uint32_t pad = h->ring_cap >= 65536u ? 16448u : 0u;
        if(const char *e = std::getenv("WF_HIP_RING_PAD"))
            pad = (uint32_t)std::strtoul(e, nullptr, 10) & ~3u;
        h->ring_stride = h->ring_cap + pad;
    }
    h->meter = cfg->meter != 0;

    auto bail = [&](int code) {
        g_create_error = h->last_error;
        wf_hip_destroy(h);
        return code;
    };
#define WF_CREATE_TRY(expr)                  \
    do {                                     \
        int rc_ = (expr);                    \
        if(rc_ != WF_HIP_OK)                 \
            return bail(rc_);                \
    } while(0)
#define WF_CREATE_HIP(expr)                                                                              \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if(e_ != hipSuccess) {                                                                           \
            fail(h, WF_HIP_ERR_RUNTIME, "%s failed: %s", #expr, hipGetErrorString(e_));                  \
            return bail(WF_HIP_ERR_RUNTIME);                                                             \
        }                                                                                                \
    } while(0)

    WF_CREATE_HIP(hipSetDevice(device));
    hipDeviceProp_t prop{};
    WF_CREATE_HIP(hipGetDeviceProperties(&prop, device));
    if(std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(h, WF_HIP_ERR_NO_DEVICE, "device %d is %s; this library contains gfx950 code only", device, prop.gcnArchName);
        return bail(WF_HIP_ERR_NO_DEVICE);
    }
    WF_CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    WF_CREATE_HIP(hipEventCreate(&h->ev0));
    WF_CREATE_HIP(hipEventCreate(&h->ev1));

    const size_t n_spec = (size_t)h->n_streams * h->cap_ch;
    WF_CREATE_TRY(dev_alloc(h, &h->d_ring, n_spec * h->ring_stride));
    WF_CREATE_TRY(dev_alloc(h, &h->d_wpos, (size_t)h->n_streams));
    if(h->wave) {
        WF_CREATE_TRY(dev_alloc(h, &h->d_cend, (size_t)h->n_streams));
        WF_CREATE_TRY(dev_alloc(h, &h->d_wts, (size_t)h->n_streams));
        WF_CREATE_TRY(dev_alloc(h, &h->d_decibels, (size_t)h->n_streams * h->out_ch * h->M));
        WF_CREATE_TRY(dev_alloc(h, &h->d_flags, (size_t)h->n_streams));
        h->kernel_name = "waveform_tick_kernel";
        WF_CREATE_TRY(wf_hip_reset(h, 0, h->n_streams));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
        *out = h;
        return WF_HIP_OK;
    }
    if(h->meter) {
        // level meter: rings, consumption points, two floats of state per channel, one bar per channel
        WF_CREATE_TRY(dev_alloc(h, &h->d_mend, (size_t)h->n_streams));
        WF_CREATE_TRY(dev_alloc(h, &h->d_meter_buf, n_spec));
        WF_CREATE_TRY(dev_alloc(h, &h->d_meter_val, n_spec));
        WF_CREATE_TRY(dev_alloc(h, &h->d_flags, (size_t)h->n_streams));
        WF_CREATE_TRY(dev_alloc(h, &h->d_bars, n_spec));
        h->kernel_name = "meter_tick_kernel";
        WF_CREATE_TRY(wf_hip_reset(h, 0, h->n_streams));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
        *out = h;
        return WF_HIP_OK;
    }
    WF_CREATE_TRY(dev_alloc(h, &h->d_tsmooth, n_spec * h->M));
    WF_CREATE_TRY(dev_alloc(h, &h->d_decibels, (size_t)h->n_streams * h->out_ch * h->M));
    // Split mode: the channels of a stereo pair in different workgroups.  Measured on MI355X: N = 16384 45 -> 52 % of the HBM
    // peak (two workgroups per CU instead of one), N = 8192 57.2 -> 58.5 % (four instead of two), N = 32768 cannot run a
    // pair any other way.  WF_HIP_SPLIT=0/1 overrides (development aid; mono mixdown and single-channel captures never split).
    bool want_split = h->geom_n >= 8192;
    if(const char *e = std::getenv("WF_HIP_SPLIT"))
        want_split = (e[0] == '1') && h->geom_n >= 8192;
    want_split = want_split && cfg->capture_channels == 2 && cfg->stereo;
    // mono mixdown needs both channels' magnitudes; where a workgroup holds one spectrum (132 KB of LDS) the pair runs split
    // as well, channel 1 a launch ahead of channel 0
    h->split_mono = h->geom_n >= 32768 && cfg->capture_channels == 2 && !cfg->stereo;
    want_split = want_split || h->split_mono;
    if(h->big_l) { // the epilogue couples the channels through the rotating verdict words, whatever the channel layout
        want_split = true;
    }
    h->flag_bufs = want_split ? 3 : 1;
    WF_CREATE_TRY(dev_alloc(h, &h->d_flags, (size_t)h->flag_bufs * h->n_streams));
    if(want_split)
        WF_CREATE_TRY(dev_alloc(h, &h->d_verdict, 3 * n_spec));
    if(h->num_bars)
        WF_CREATE_TRY(dev_alloc(h, &h->d_bars, (size_t)h->n_streams * h->disp_ch * h->num_bars));
    if(cfg->vertices) {
        if(cfg->vertices > 3u || (cfg->vertices == 3u && (!cfg->bars || cfg->step_width < 1 || cfg->step_gap < 0)) || (cfg->vertices == 2u && cfg->bars) ||
           (!cfg->bars && !cfg->curve))
            return bail(fail(h, WF_HIP_ERR_INVALID, "cfg.vertices: 1 needs bars or curve, 2 the curve, 3 bars with step_width >= 1 and step_gap >= 0"));
        // a display narrower than one bar (m_num_bars == 0), or steps taller than the channel: the reference allocates no vertex
        // buffer ("Tried to allocate vbuf of size: 0", src/source.cpp:1044) and draws nothing -- wf_hip_num_vertices() == 0
        if(h->num_bars != 0)
            wf::build_vertex_tables(*cfg, (int)h->num_bars, h->vtab);
        if(h->num_bars != 0 && h->vtab.per_row > 0) {
            WF_CREATE_TRY(dev_alloc(h, &h->d_vert_counts, (size_t)h->n_streams * h->disp_ch));
            WF_CREATE_HIP(hipMemsetAsync(h->d_vert_counts, 0, (size_t)h->n_streams * h->disp_ch * sizeof(uint32_t), h->stream));
            WF_CREATE_TRY(dev_alloc(h, &h->d_verts, (size_t)h->n_streams * h->disp_ch * h->vtab.per_row));
            WF_CREATE_HIP(hipMemsetAsync(h->d_verts, 0, (size_t)h->n_streams * h->disp_ch * h->vtab.per_row * sizeof(wf::f4), h->stream));
            WF_CREATE_TRY(upload(h, &h->d_cap_xy, h->vtab.cap_xy));
            WF_CREATE_HIP(hipStreamSynchronize(h->stream));
        }
    }

#ifdef WF_PHASE_TIMING
    WF_CREATE_TRY(dev_alloc(h, &h->d_phase_clock, n_spec * 16));
#endif
    // the kernel always multiplies by the window and slope tables; a disabled feature is a table of ones (x * 1.0f == x)
    {
        const std::vector<float> ones_m(h->M, 1.0f);
        h->in_scale = (h->big_l && h->blu) ? 0x1p24f : 0x1p40f;
        std::vector<float> win_dev(h->N, h->in_scale);
        for(size_t i = 0; i < h->tab.window.size() && i < win_dev.size(); ++i)
            win_dev[i] = h->tab.window[i] * h->in_scale; // (exact)
        WF_CREATE_TRY(upload(h, &h->d_window, win_dev));
        WF_CREATE_TRY(upload(h, &h->d_slope, h->tab.slope.empty() ? ones_m : h->tab.slope));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
    }
    WF_CREATE_TRY(upload(h, &h->d_rolloff, h->tab.rolloff));
    std::vector<int> chunks;
    // The display tables.  ext == false: the outputs are finished inside the tick kernel, from the dB row parked in the
    // spectrum's exchange buffer (or, beyond a CU's LDS, by big_outputs_kernel).  Where the row's points + the Gaussian
    // filter's staging do not fit that buffer -- wide filtered curves and many narrow filtered bars at small fft sizes: the
    // reference allows width <= 3840 and radius <= 32 at every size (src/source.cpp:287, :409) -- the plan is made again with
    // ext == true: the tick kernel stores its rows and big_outputs_kernel (one workgroup per displayed row, up to 160 KB of
    // LDS) derives the outputs from them through L2, as it does for the transforms beyond a CU's LDS.
#define WF_PLAN_TRY(expr)                    \
    do {                                     \
        int rc_ = (expr);                    \
        if(rc_ != WF_HIP_OK)                 \
            return rc_;                      \
    } while(0)
#define WF_PLAN_HIP(expr)                                                                                \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if(e_ != hipSuccess)                                                                             \
            return fail(h, WF_HIP_ERR_RUNTIME, "%s failed: %s", #expr, hipGetErrorString(e_));          \
    } while(0)
    auto plan_outputs = [&](bool ext) -> int {
        WF_PLAN_TRY(upload(h, &h->d_bar_coef, h->tab.bar_coef));
        WF_PLAN_TRY(upload(h, &h->d_bar_bin, h->tab.bar_bin));
        WF_PLAN_TRY(upload(h, &h->d_bar_off, h->tab.bar_off));
        WF_PLAN_TRY(upload(h, &h->d_band_widths, h->tab.band_widths));
        // LDS scratch for the products: what is left of a spectrum's exchange buffer behind the M dB values
        size_t lds_floats = 0;
        int threads = 64;
        const bool own_kernel = h->big_l != 0 || ext; // the outputs come from the stored rows, by big_outputs_kernel
        wf::dispatch_geometry(ext ? 32768u : h->geom_n, [&](auto g) {
            using G = decltype(g);
            lds_floats = (size_t)G::LDS_CF * 2;
            threads = G::T;
        });
        // the kernels that run on wf::GBig's 1024 threads of 16 points whatever the power-of-two kernel of that size does:
        // big_outputs_kernel, and the Bluestein / mixed-radix instantiations of the largest container (setup_launch_blu)
        const bool on_gbig = own_kernel || (h->blu && h->geom_n == 32768u);
        if(on_gbig) {
            lds_floats = (size_t)wf::GBig::LDS_CF * 2;
            threads = wf::GBig::T;
        }
        int lpb = 1;
        while(lpb < 64 && (uint32_t)(threads / (lpb * 2)) >= h->num_bars)
            lpb *= 2;
        h->bar_lpb = lpb;
        int points = 16;
        wf::dispatch_geometry(ext ? 32768u : h->geom_n, [&](auto g) { points = decltype(g)::P; });
        if(on_gbig)
            points = wf::GBig::P;
        const int kmax = threads <= 64 ? 16 : 8; // wf::OutVals<G>::KMAX
        h->curve = !cfg->bars && cfg->curve;
        if(h->curve) {
            // one curve point per thread and step; the filter stages the row's points in the spectrum's LDS
            wf::CurveLaneTables cl;
            // mono mixdown with both channels of a stream in one workgroup: the one displayed row is finished by the threads
            // of both spectra (spectrum_tick_kernel, BarArgs::both_subs)
            h->curve_both = !cfg->stereo && cfg->capture_channels == 2 && !own_kernel && !want_split && !h->blu && h->N >= 1024u &&
                            std::getenv("WF_HIP_TLDS") == nullptr; // (the kernels that exist with BOTH: setup_launch)
            if(const char *e = std::getenv("WF_HIP_CURVE_BOTH"))
                h->curve_both = h->curve_both && e[0] != '0';
            if(h->curve_both)
                threads *= 2;
            if(!wf::curve_lanes(h->tab, *cfg, threads, kmax, cl))
                return (fail(h, WF_HIP_ERR_INVALID, "curve display: no point table for width %u at fft_size %u", cfg->width, h->N));
            h->out_steps = cl.steps;
            h->curve_catrom = !cl.x.empty();
            h->stream_steps = cl.steps > kmax || own_kernel; // wider than a thread's registers hold (always on the large-transform path, whose outputs have a kernel of their own): points are finished as they are produced
            WF_PLAN_TRY(upload(h, &h->d_cur_coef, cl.coef));
            WF_PLAN_TRY(upload(h, &h->d_cur_base, cl.base));
            WF_PLAN_TRY(upload(h, &h->d_cur_x, cl.x));
            WF_PLAN_HIP(hipStreamSynchronize(h->stream)); // the staging vectors die here
        } else if(!own_kernel) { // (big_outputs_kernel reduces its bars from the flat tables, one wavefront per bar)
            wf::BarLaneTables lanes;
            // (wave-local layout: no workgroup barrier inside the reduction; not with the filter, whose inputs are staged by
            // bar index behind a barrier anyway.  WF_HIP_BARS_WAVE_LOCAL=0: the plain layout, development aid)
            bool local = h->tab.gauss_radius == 0;
            if(const char *e = std::getenv("WF_HIP_BARS_WAVE_LOCAL"))
                local = local && e[0] != '0';
            if(wf::bar_segments(h->tab, threads, points / 4 + 2, lanes, local)) {
                h->bar_wave_local = lanes.wave_local;
                h->bar_segs = lanes.num_segs;
                h->bar_blocks = lanes.blocks;
                h->out_steps = 1;
                WF_PLAN_TRY(upload(h, &h->d_lane_coef, lanes.coef));
                WF_PLAN_TRY(upload(h, &h->d_lane_base, lanes.base));
                WF_PLAN_TRY(upload(h, &h->d_bar_seg, lanes.bar_seg));
                WF_PLAN_TRY(upload(h, &h->d_seg_group, lanes.seg_group));
                WF_PLAN_TRY(upload(h, &h->d_lead_bar, lanes.lead_bar));
                WF_PLAN_TRY(upload(h, &h->d_lead_end, lanes.lead_end));
                WF_PLAN_HIP(hipStreamSynchronize(h->stream)); // the staging vectors die here
            }
        }
        size_t chunk_cap = lds_floats > h->M ? lds_floats - h->M : 0; // LDS scratch for the products: what is left behind the dB row
        if(own_kernel) {
            // big_outputs_kernel: the whole row in LDS, two guard zeros, then the filter's staging
            const size_t staged = h->tab.gauss_radius > 0 ? (size_t)h->num_bars + 2 * (size_t)(h->tab.gauss_radius - 1) + h->tab.gauss.size() : 0;
            // (bars read their bins from the row in device memory: only the staging lives in LDS; a curve parks the row first)
            const size_t parked = h->curve ? (size_t)h->M + 2 : 0;
            h->bar_stage_off = (int)parked;
            // bars: the entries in tasks of at most 2048 (a multiple of 64), one wavefront each; their sums meet in LDS
            h->big_num_tasks = 0;
            if(!h->curve && !h->tab.bar_off.empty()) {
                std::vector<int> task, bar_task(h->tab.bar_off.size(), 0);
                for(size_t bq = 0; bq + 1 < h->tab.bar_off.size(); ++bq) {
                    bar_task[bq] = (int)(task.size() / 3);
                    const int e0 = h->tab.bar_off[bq], e1 = h->tab.bar_off[bq + 1];
                    const int parts = std::max(1, (e1 - e0 + 2047) / 2048);
                    const int per = (((e1 - e0 + parts - 1) / parts) + 63) & ~63;
                    for(int q = 0; q < parts; ++q) {
                        const int lo = std::min(e0 + q * per, e1), hi = std::min(lo + per, e1);
                        if(q == 0 || lo < hi) {
                            task.push_back((int)bq);
                            task.push_back(lo);
                            task.push_back(hi);
                        }
                    }
                }
                bar_task.back() = (int)(task.size() / 3);
                h->big_num_tasks = (int)(task.size() / 3);
                WF_PLAN_TRY(upload(h, &h->d_big_task, task));
                WF_PLAN_TRY(upload(h, &h->d_big_bar_task, bar_task));
                WF_PLAN_HIP(hipStreamSynchronize(h->stream));
            }
            h->big_out_lds = std::max<size_t>(((parked + staged + (size_t)h->big_num_tasks) * sizeof(float) + 15) & ~(size_t)15, 16);
            if(h->big_out_lds > 160u * 1024u)
                return (fail(h, WF_HIP_ERR_UNSUPPORTED, "fft_size %u with filter_mode gauss over %u outputs: row + staging exceed a CU's LDS", h->N,
                                 h->num_bars));
            if(h->tab.gauss_radius > 0) {
                WF_PLAN_TRY(upload(h, &h->d_gauss, h->tab.gauss));
                WF_PLAN_TRY(upload(h, &h->d_gauss_wsum, h->tab.gauss_wsum));
                WF_PLAN_HIP(hipStreamSynchronize(h->stream));
            }
        } else if(h->tab.gauss_radius > 0) {
            // staged in the spectrum's LDS: the row with radius-1 zeros on either side, then the weights
            const size_t staged = (size_t)h->num_bars + 2 * (size_t)(h->tab.gauss_radius - 1) + h->tab.gauss.size();
            if(h->stream_steps) {
                // wide curve: the points are staged behind the dB row (and the two guard zeros of the Catmull-Rom taps)
                if(h->M + 2 + staged > lds_floats) {
                    return (fail(h, WF_HIP_ERR_UNSUPPORTED,
                                     "filter_mode gauss: %u curve points + the filter's staging do not fit behind the row in this configuration's on-chip buffer (%zu floats)",
                                     h->num_bars, lds_floats));
                }
                h->bar_stage_off = (int)h->M + 2;
            } else if(h->out_steps == 0) {
                // bars in chunked form (more bars than threads): the staging area sits at the end of the buffer, the product
                // scratch shrinks by it and must still hold the longest bar
                int longest = 0;
                for(uint32_t b = 0; b < h->num_bars; ++b)
                    longest = std::max(longest, h->tab.bar_off[(size_t)b + 1] - h->tab.bar_off[(size_t)b]);
                if(staged + (size_t)longest + h->M > lds_floats) {
                    return (fail(h, WF_HIP_ERR_UNSUPPORTED,
                                     "filter_mode gauss: %u bars + the filter's staging do not fit this configuration's on-chip buffer (%zu floats)",
                                     h->num_bars, lds_floats));
                }
                chunk_cap -= staged;
                h->bar_stage_off = (int)(lds_floats - staged);
            } else if(staged > lds_floats) {
                return (fail(h, WF_HIP_ERR_UNSUPPORTED,
                                 "filter_mode gauss: %u outputs per row do not fit this configuration's on-chip staging (%zu floats)",
                                 h->num_bars, lds_floats));
            }
            WF_PLAN_TRY(upload(h, &h->d_gauss, h->tab.gauss));
            WF_PLAN_TRY(upload(h, &h->d_gauss_wsum, h->tab.gauss_wsum));
            WF_PLAN_HIP(hipStreamSynchronize(h->stream));
        }
        if(h->bar_segs == 0 && !h->curve && !own_kernel) { // chunked form: a chunk holds at least one whole bar
            int longest = 0;
            for(uint32_t b = 0; b < h->num_bars; ++b)
                longest = std::max(longest, h->tab.bar_off[(size_t)b + 1] - h->tab.bar_off[(size_t)b]);
            if((size_t)longest > chunk_cap)
                return (fail(h, WF_HIP_ERR_UNSUPPORTED, "bars: the widest band (%d bins and taps) does not fit the on-chip scratch (%zu floats)",
                                 longest, chunk_cap));
        }
        chunks = wf::bar_chunks(h->tab, chunk_cap);
        WF_PLAN_TRY(upload(h, &h->d_bar_chunk, chunks));
        WF_PLAN_HIP(hipStreamSynchronize(h->stream));
        h->bar_chunks = (int)chunks.size() - 1;
        return WF_HIP_OK;
    };
#undef WF_PLAN_TRY
#undef WF_PLAN_HIP
    if(h->num_bars) {
        const size_t mark = h->allocs.size();
        int orc = plan_outputs(false);
        if(orc == WF_HIP_ERR_UNSUPPORTED && h->big_l == 0) {
            // give back what the first plan uploaded, forget what it decided, plan again for big_outputs_kernel
            WF_CREATE_HIP(hipStreamSynchronize(h->stream));
            while(h->allocs.size() > mark) {
                (void)hipFree(h->allocs.back());
                h->allocs.pop_back();
            }
            h->d_bar_coef = nullptr; h->d_bar_bin = nullptr; h->d_bar_off = nullptr; h->d_band_widths = nullptr; h->d_bar_chunk = nullptr;
            h->d_cur_coef = nullptr; h->d_cur_base = nullptr; h->d_cur_x = nullptr; h->d_gauss = nullptr; h->d_gauss_wsum = nullptr;
            h->d_lane_coef = nullptr; h->d_lane_base = nullptr; h->d_bar_seg = nullptr; h->d_seg_group = nullptr;
            h->d_lead_bar = nullptr; h->d_lead_end = nullptr;
            h->curve = h->curve_both = h->curve_catrom = h->stream_steps = h->bar_wave_local = false;
            h->out_steps = h->bar_segs = h->bar_blocks = h->bar_chunks = h->bar_stage_off = 0;
            h->bar_lpb = 1;
            chunks.clear();
            h->ext_outputs = true;
            orc = plan_outputs(true);
        }
        if(orc)
            return bail(orc);
        if(h->ext_outputs && h->big_out_lds)
            WF_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_outputs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)h->big_out_lds));
    }

    // FFT plan: twiddle tables for the geometry of this fft_size + the kernel instantiation
    int setup_rc = WF_HIP_ERR_UNSUPPORTED;
    std::vector<wf::cfloat> tw1, tw2, tws;
    wf::dispatch_geometry(h->geom_n, [&](auto g) {
        using G = decltype(g);
        wf::build_twiddles(G::M, G::R1, G::R2, G::R3, tw1, tw2, tws);
        h->waves_per_spectrum = G::T / 64;
        // the channels of a stream share a workgroup (silence state machine, mono mixdown)
        if(h->big_l) {
            if constexpr(G::N == 32768)
                setup_rc = setup_launch_big(h);
        } else
#ifdef WF_GEOM_ONLY // development builds: no Bluestein instantiations
        if(h->blu) {
            setup_rc = fail(h, WF_HIP_ERR_UNSUPPORTED, "development build without the Bluestein kernels");
        } else
#endif
        if(h->blu) {
            if constexpr(G::N >= 32768) {
                // (the Bluestein and mixed-radix instantiations of this container keep 1024 threads of 16 points: a mixed-radix
                // plan's last pass has one butterfly per thread at most, and 39 sizes have no plan on 512 threads)
                using GB = wf::GBig;
                h->waves_per_spectrum = GB::T / 64;
                if(want_split)
                    setup_rc = setup_launch_blu<GB, 1, true>(h);
                else if(cfg->capture_channels == 1)
                    setup_rc = setup_launch_blu<GB, 1, false>(h);
                else
                    setup_rc = fail(h, WF_HIP_ERR_RUNTIME, "fft_size %u: no launch plan", cfg->fft_size);
            } else if constexpr(G::T >= 256)
                setup_rc = want_split ? setup_launch_blu<G, 1, true>(h)
                                      : (cfg->capture_channels > 1) ? setup_launch_blu<G, 2, false>(h) : setup_launch_blu<G, 1, false>(h);
            else
                setup_rc = setup_launch_blu<G, 2, false>(h);
        } else if constexpr(G::N == 512) {
            switch(h->N) {
            case 256: setup_rc = setup_launch_dec<G, 1>(h); break;
            case 128: setup_rc = setup_launch_dec<G, 2>(h); break;
            default: setup_rc = setup_launch<G, 2>(h); break;
            }
        } else if constexpr(G::N >= 32768) {
            // one spectrum fills a CU's LDS: a stereo pair runs split, a single captured channel alone; mono mixdown of two
            // channels (which needs both in one workgroup) is not available at this size
            if(want_split)
                setup_rc = setup_launch_split<G>(h);
            else if(cfg->capture_channels == 1)
                setup_rc = setup_launch<G, 1>(h);
            else
                setup_rc = fail(h, WF_HIP_ERR_RUNTIME, "fft_size %u: no launch plan", cfg->fft_size);
        } else if constexpr(G::T >= 256)
            setup_rc = want_split ? setup_launch_split<G>(h) : (cfg->capture_channels > 1) ? setup_launch<G, 2>(h) : setup_launch<G, 1>(h);
        else
            setup_rc = setup_launch<G, 2>(h);
    });
    WF_CREATE_TRY(setup_rc);
    static_assert(sizeof(wf::cfloat) == sizeof(wf::cf), "twiddle layout");
    {
        std::vector<wf::cf> t1(tw1.size()), t2(tw2.size()), t3(tws.size());
        std::memcpy(t1.data(), tw1.data(), tw1.size() * sizeof(wf::cf));
        std::memcpy(t2.data(), tw2.data(), tw2.size() * sizeof(wf::cf));
        std::memcpy(t3.data(), tws.data(), tws.size() * sizeof(wf::cf));
        if(h->mr_passes > 0 && h->mr_radix[0] > 25) {
            // a mixed-radix plan that opens with a prime pass (wf::mr_pass_prime): its W_p^m goes where the power-of-two kernels keep
            // their pass-2 twiddles -- the tick kernel stages that table in LDS anyway and the mixed-radix passes do not use it
            std::vector<wf::cfloat> wp;
            wf::build_prime_twiddles(h->mr_radix[0], t2.size(), wp);
            t2.resize(wp.size());
            std::memcpy(t2.data(), wp.data(), wp.size() * sizeof(wf::cf));
            WF_CREATE_TRY(upload(h, &h->d_mr_wp, t2)); // (the large-FFT rows kernel reads it from device memory)
        }
        WF_CREATE_TRY(upload(h, &h->d_tw1, t1));
        WF_CREATE_TRY(upload(h, &h->d_tw2, t2));
        WF_CREATE_TRY(upload(h, &h->d_tws, t3));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream)); // the staging vectors die here
    }
    if(h->blu && h->mr_passes > 0) {
        // mixed radix: the window table the power-of-two kernels use (it is uploaded for every handle), W_(N/2)^m for the passes
        // and W_N^k for the real split; none of Bluestein's chirp tables
        std::vector<wf::cfloat> twf, wf_;
        wf::build_mixed_radix_tables(h->N, h->mr_passes, h->mr_radix, twf, h->mr_tw_off, wf_);
        std::vector<wf::cf> t1(twf.size()), t2(wf_.size());
        std::memcpy(t1.data(), twf.data(), t1.size() * sizeof(wf::cf));
        std::memcpy(t2.data(), wf_.data(), t2.size() * sizeof(wf::cf));
        WF_CREATE_TRY(upload(h, &h->d_mr_tw, t1));
        WF_CREATE_TRY(upload(h, &h->d_blu_w, t2));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
    } else if(h->blu) {
        wf::BluesteinTables bt;
        wf::build_bluestein(h->cfg, h->tab, bt);
        std::vector<wf::cf> ta(bt.a.size()), tb(bt.b.size());
        std::memcpy(ta.data(), bt.a.data(), ta.size() * sizeof(wf::cf));
        std::memcpy(tb.data(), bt.b.data(), tb.size() * sizeof(wf::cf));
        for(auto &v : ta) { // the window sits in this table on the Bluestein paths (in_scale)
            v.x *= h->in_scale;
            v.y *= h->in_scale;
        }
        WF_CREATE_TRY(upload(h, &h->d_blu_a, ta));
        WF_CREATE_TRY(upload(h, &h->d_blu_b, tb));
        std::vector<wf::cf> tq(bt.q.size()), tqr(bt.qr.size()), tw(bt.w.size());
        if(!tq.empty()) {
            std::memcpy(tq.data(), bt.q.data(), tq.size() * sizeof(wf::cf));
            std::memcpy(tqr.data(), bt.qr.data(), tqr.size() * sizeof(wf::cf));
            std::memcpy(tw.data(), bt.w.data(), tw.size() * sizeof(wf::cf));
        }
        WF_CREATE_TRY(upload(h, &h->d_blu_q, tq));
        WF_CREATE_TRY(upload(h, &h->d_blu_qr, tqr));
        WF_CREATE_TRY(upload(h, &h->d_blu_w, tw));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
    }
    if(h->big_mr) {
        // the rows' passes (a transform of R = n / 2 / C points) and the column step's W_C^(c k1)
        std::vector<wf::cfloat> twf, unused;
        wf::build_mixed_radix_tables(2u * (h->M / h->big_rows), h->mr_passes, h->mr_radix, twf, h->mr_tw_off, unused);
        std::vector<wf::cf> t1(twf.size()), wc(64, wf::cf{1.0f, 0.0f});
        std::memcpy(t1.data(), twf.data(), t1.size() * sizeof(wf::cf));
        const double two_pi = 6.283185307179586476925286766559;
        for(uint32_t k1 = 0; k1 < h->big_rows; ++k1)
            for(uint32_t c = 0; c < h->big_rows; ++c) {
                const double ang = -two_pi * (double)((c * k1) % h->big_rows) / (double)h->big_rows;
                wc[k1 * 8u + c] = wf::cf{(float)std::cos(ang), (float)std::sin(ang)};
            }
        WF_CREATE_TRY(upload(h, &h->d_mr_tw, t1));
        WF_CREATE_TRY(upload(h, &h->d_big_wc, wc));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
    }
    if(h->big_l) {
        std::vector<wf::cfloat> twb, twsb;
        wf::build_big_twiddles(h->big_l, h->big_rows, h->blu ? 0u : h->N, twb, twsb);
        std::vector<wf::cf> t1(twb.size()), t2(twsb.size());
        std::memcpy(t1.data(), twb.data(), t1.size() * sizeof(wf::cf));
        if(!t2.empty())
            std::memcpy(t2.data(), twsb.data(), t2.size() * sizeof(wf::cf));
        WF_CREATE_TRY(upload(h, &h->d_big_tw, t1));
        WF_CREATE_TRY(upload(h, &h->d_big_tws, t2));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
        if(h->big_fused) { // (no complex scratch: the rows kernel reads the ring and leaves magnitudes)
            WF_CREATE_TRY(dev_alloc(h, &h->d_big_mag, n_spec * 2u * 16384u));
        } else if(h->big_mr) { // (the rows read the ring themselves: one scratch buffer, for Z)
            WF_CREATE_TRY(dev_alloc(h, &h->d_big_z, n_spec * h->big_l));
        } else {
            WF_CREATE_TRY(dev_alloc(h, &h->d_big_v, n_spec * h->big_l));
            WF_CREATE_TRY(dev_alloc(h, &h->d_big_z, n_spec * h->big_l));
        }
        WF_CREATE_TRY(dev_alloc(h, &h->d_big_nz, n_spec));
    }
    {
        // lanes (see struct wf_hip): two slices once each still fills the chip a couple of times over.  Measured on MI355X
        // (cfg3, 8192 spectra per tick, back-to-back ticks): 1 lane 66 us per tick, 2 lanes 58 us.  WF_HIP_LANES overrides.
        // Two lanes pay once the batch fills the chip at least twice over (a lane's drain and ramp-up then fall under the other's
        // steady state); a batch of one round or less only pays the fork / join events: N = 4096 x 1024 streams -- exactly one
        // round of 4 workgroups per CU -- 0.625 on one lane, 0.545 on two; 3 and 4 lanes: -1..-4 % everywhere.
        const uint32_t wgs = (uint32_t)(n_spec / (h->split ? 1u : 2u));
        const uint32_t per_cu = std::max(1u, std::min(h->wg_lds ? (160u * 1024u) / h->wg_lds : 16u, h->wg_threads ? 1024u / h->wg_threads : 16u));
        const uint32_t round = per_cu * (uint32_t)std::max(prop.multiProcessorCount, 1);
        int lanes = wgs >= 2u * round ? 2 : 1;
        if(h->M <= 512 && !h->cfg.meter && !h->cfg.waveform && wgs >= 6u * round)
            lanes = 3; // the one-wavefront 8-point geometry in long launches: 0.714-0.717 against 0.682-0.683 of the HBM peak at
                       // 16384 streams (steady state, r02j); +-2 % on every other geometry
        if(const char *e = std::getenv("WF_HIP_LANES"))
            lanes = std::atoi(e);
        lanes = std::max(1, std::min({lanes, (int)wf_hip::MAX_LANES, (int)h->n_streams}));
        if(h->big_l)
            lanes = 1; // a handful of workgroups of a whole CU each: nothing to overlap
#ifdef WF_PHASE_TIMING
        lanes = 1;
#endif
        for(int l = 1; l < lanes; ++l) {
            WF_CREATE_HIP(hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking));
            WF_CREATE_HIP(hipEventCreateWithFlags(&h->ev_lane[l], hipEventDisableTiming));
        }
        if(lanes > 1)
            WF_CREATE_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        h->n_lanes = lanes;
    }
    WF_CREATE_TRY(wf_hip_reset(h, 0, h->n_streams));
    WF_CREATE_HIP(hipStreamSynchronize(h->stream));
#undef WF_CREATE_TRY
#undef WF_CREATE_HIP
    *out = h;
    return WF_HIP_OK;
}

void wf_hip_destroy(wf_hip *h)
{
    if(h == nullptr)
        return;
    (void)hipSetDevice(h->device);
    for(int l = 1; l < wf_hip::MAX_LANES; ++l)
        if(h->lane_stream[l])
            (void)hipStreamSynchronize(h->lane_stream[l]);
    if(h->stream)
        (void)hipStreamSynchronize(h->stream);
    for(int l = 1; l < wf_hip::MAX_LANES; ++l) {
        if(h->ev_lane[l]) (void)hipEventDestroy(h->ev_lane[l]);
        if(h->lane_stream[l]) (void)hipStreamDestroy(h->lane_stream[l]);
    }
    if(h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for(void *p : h->allocs)
        (void)hipFree(p);
    if(h->copy_stream)
        (void)hipStreamSynchronize(h->copy_stream);
    for(int i = 0; i < 2; ++i) {
        if(h->ev_copied[i]) (void)hipEventDestroy(h->ev_copied[i]);
        if(h->ev_consumed[i]) (void)hipEventDestroy(h->ev_consumed[i]);
    }
    if(h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if(h->read_stream)
        (void)hipStreamSynchronize(h->read_stream);
    for(int i = 0; i < 2; ++i) {
        if(h->ev_snap[i]) (void)hipEventDestroy(h->ev_snap[i]);
        if(h->ev_read[i]) (void)hipEventDestroy(h->ev_read[i]);
    }
    if(h->read_stream) (void)hipStreamDestroy(h->read_stream);
    for(int i = 0; i < 2; ++i) {
        if(h->h_frames_async[i]) (void)hipHostFree(h->h_frames_async[i]);
        if(h->h_sq_frames[i]) (void)hipHostFree(h->h_sq_frames[i]);
        if(h->ev_sq_consumed[i]) (void)hipEventDestroy(h->ev_sq_consumed[i]);
    }
    for(auto e : h->ev_bars_lane)
        if(e) (void)hipEventDestroy(e);
    if(h->ev0) (void)hipEventDestroy(h->ev0);
    if(h->ev1) (void)hipEventDestroy(h->ev1);
    if(h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// update(): m_rms_sync_buf empty, m_input_rms_buf = 0, m_input_rms = 0 (src/source.cpp:1144-1152); no-op unless the device
// producer is enabled
static int reset_rms_producer(wf_hip *h, uint32_t first, uint32_t count)
{
    if(h->d_rms_ring == nullptr)
        return WF_HIP_OK;
    const size_t nblk = h->rms_cap / wf::RMS_BLOCK;
    WF_HIP_TRY(h, hipMemsetAsync(h->d_rms_ring + (size_t)first * h->rms_cap, 0, (size_t)count * h->rms_cap * sizeof(float), h->stream));
    WF_HIP_TRY(h, hipMemsetAsync(h->d_rms_bsum + (size_t)first * nblk, 0, (size_t)count * nblk * sizeof(float), h->stream));
    WF_HIP_TRY(h, hipMemsetAsync(h->d_rend + first, 0, (size_t)count * sizeof(uint32_t), h->stream));
    WF_HIP_TRY(h, hipMemsetAsync(h->d_input_rms + first, 0, (size_t)count * sizeof(float), h->stream));
    return WF_HIP_OK;
}

int wf_hip_reset(wf_hip *h, uint32_t first, uint32_t count)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_TRY_RC(wait_rows_in_flight(h)); // the fills below overwrite rows a readback may still be copying
    const size_t spec0 = (size_t)first * h->cap_ch, nspec = (size_t)count * h->cap_ch;
    if(h->wave) {
        // update() in waveform mode (src/source.cpp:1142, :1172-1182, :1243-1248): rows = DB_MIN, rings = width zeros, m_waveform_ts = 0
        WF_HIP_TRY(h, hipMemsetAsync(h->d_ring + spec0 * h->ring_stride, 0, nspec * h->ring_stride * sizeof(float), h->stream));
        WF_HIP_TRY(h, hipMemsetAsync(h->d_flags + first, 0, (size_t)count * sizeof(uint32_t), h->stream));
        WF_HIP_TRY(h, hipMemsetAsync(h->d_cend + first, 0, (size_t)count * sizeof(uint32_t), h->stream));
        WF_HIP_TRY(h, hipMemsetAsync(h->d_wts + first, 0, (size_t)count * sizeof(unsigned long long), h->stream));
        const size_t ndb = (size_t)count * h->out_ch * h->M;
        hipLaunchKernelGGL(wf::fill_f32_kernel, dim3((unsigned)std::min<size_t>((ndb + 255) / 256, 4096)), dim3(256), 0, h->stream,
                           h->d_decibels + (size_t)first * h->out_ch * h->M, ndb, wf::db_min());
        hipLaunchKernelGGL(wf::fill_u32_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->d_wpos + first, (size_t)count,
                           h->N);
        WF_HIP_TRY(h, hipGetLastError());
        if(first == 0 && count == h->n_streams)
            h->all_aligned = true;
        return reset_rms_producer(h, first, count); // waveform batches normalise too (src/source_generic.cpp:376-388)
    }
    if(h->meter) {
        // update() in meter mode (src/source.cpp:1123-1127, :1181, :1243): empty rings (no zero pre-fill), meter buffer 0,
        // m_meter_buf = m_meter_val = DB_MIN, m_last_silent = false
        WF_HIP_TRY(h, hipMemsetAsync(h->d_ring + spec0 * h->ring_stride, 0, nspec * h->ring_stride * sizeof(float), h->stream));
        WF_HIP_TRY(h, hipMemsetAsync(h->d_flags + first, 0, (size_t)count * sizeof(uint32_t), h->stream));
        WF_HIP_TRY(h, hipMemsetAsync(h->d_wpos + first, 0, (size_t)count * sizeof(uint32_t), h->stream));
        WF_HIP_TRY(h, hipMemsetAsync(h->d_mend + first, 0, (size_t)count * sizeof(uint32_t), h->stream));
        const dim3 g((unsigned)((nspec + 255) / 256)), b(256);
        hipLaunchKernelGGL(wf::fill_f32_kernel, g, b, 0, h->stream, h->d_meter_buf + spec0, nspec, wf::db_min());
        hipLaunchKernelGGL(wf::fill_f32_kernel, g, b, 0, h->stream, h->d_meter_val + spec0, nspec, wf::db_min());
        hipLaunchKernelGGL(wf::fill_f32_kernel, g, b, 0, h->stream, h->d_bars + spec0, nspec, h->tab.border_bottom);
        WF_HIP_TRY(h, hipGetLastError());
        if(first == 0 && count == h->n_streams)
            h->all_aligned = true;
        return WF_HIP_OK;
    }
    // m_tsmooth_buf = 0, rings = zeros with N samples "written", m_decibels = DB_MIN, m_last_silent = false
    WF_HIP_TRY(h, hipMemsetAsync(h->d_tsmooth + spec0 * h->M, 0, nspec * h->M * sizeof(float), h->stream));
    WF_HIP_TRY(h, hipMemsetAsync(h->d_ring + spec0 * h->ring_stride, 0, nspec * h->ring_stride * sizeof(float), h->stream));
    for(uint32_t b = 0; b < h->flag_bufs; ++b)
        WF_HIP_TRY(h, hipMemsetAsync(h->d_flags + (size_t)b * h->n_streams + first, 0, (size_t)count * sizeof(uint32_t), h->stream));
    if(h->d_row_verdict)
        WF_HIP_TRY(h, hipMemsetAsync(h->d_row_verdict + spec0 * h->waves_per_spectrum, 0, nspec * h->waves_per_spectrum * sizeof(uint32_t), h->stream));
    if(h->d_verdict) // rows of DB_MIN: nothing above floor - 10
        for(uint32_t b = 0; b < 3; ++b)
            WF_HIP_TRY(h, hipMemsetAsync(h->d_verdict + (size_t)b * h->n_streams * h->cap_ch + spec0, 0, nspec * sizeof(uint32_t), h->stream));
    const size_t ndb = (size_t)count * h->out_ch * h->M;
    hipLaunchKernelGGL(wf::fill_f32_kernel, dim3((unsigned)std::min<size_t>((ndb + 255) / 256, 4096)), dim3(256), 0, h->stream,
                       h->d_decibels + (size_t)first * h->out_ch * h->M, ndb, wf::db_min());
    hipLaunchKernelGGL(wf::fill_u32_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->d_wpos + first, (size_t)count,
                       h->N);
    if(h->d_bars) {
        // what render_bars shows for rows of DB_MIN: every bar at border_bottom (zero height)
        const size_t nb = (size_t)count * h->disp_ch * h->num_bars;
        hipLaunchKernelGGL(wf::fill_f32_kernel, dim3((unsigned)std::min<size_t>((nb + 255) / 256, 4096)), dim3(256), 0, h->stream,
                           h->d_bars + (size_t)first * h->disp_ch * h->num_bars, nb, h->tab.border_bottom);
    }
    WF_HIP_TRY(h, hipGetLastError());
    int rrc = reset_rms_producer(h, first, count);
    if(rrc)
        return rrc;
    if(first == 0 && count == h->n_streams)
        h->all_aligned = true; // every write position is back at fft_size
    return WF_HIP_OK;
}

uint32_t wf_hip_fft_size(const wf_hip *h) { return h ? h->N : 0; }
uint32_t wf_hip_num_streams(const wf_hip *h) { return h ? h->n_streams : 0; }
uint32_t wf_hip_capture_channels(const wf_hip *h) { return h ? h->cap_ch : 0; }
uint32_t wf_hip_output_channels(const wf_hip *h) { return h ? h->out_ch : 0; }
uint32_t wf_hip_display_channels(const wf_hip *h) { return h ? h->disp_ch : 0; }
uint32_t wf_hip_num_bars(const wf_hip *h) { return h ? h->num_bars : 0; }
uint32_t wf_hip_ring_frames(const wf_hip *h) { return h ? h->ring_cap : 0; }

static int push_host(wf_hip *h, uint32_t first, uint32_t count, const float *samples, uint32_t frames, bool muted)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(samples == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "samples is NULL");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)count * h->cap_ch * frames;
    // the staging block may still feed a previous push: the copy below is ordered after it on the same stream
    rc = ensure_stage(h, n);
    if(rc)
        return rc;
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_stage, samples, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    rc = push_common(h, first, count, muted ? nullptr : h->d_stage, h->d_stage, frames);
    if(rc)
        return rc;
    // `samples` is borrowed only for the duration of the call (pageable memory: the copy has been staged by the
    // runtime when hipMemcpyAsync returns; pinned memory: wait for it)
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
    return WF_HIP_OK;
}

int wf_hip_push_audio(wf_hip *h, uint32_t first, uint32_t count, const float *samples, uint32_t frames)
{
    return push_host(h, first, count, samples, frames, false);
}

int wf_hip_push_audio_muted(wf_hip *h, uint32_t first, uint32_t count, const float *samples, uint32_t frames)
{
    return push_host(h, first, count, samples, frames, true);
}

int wf_hip_push_audio_async(wf_hip *h, uint32_t first, uint32_t count, const float *pinned_samples, uint32_t frames, uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(pinned_samples == nullptr || slot > 1)
        return fail(h, WF_HIP_ERR_INVALID, "samples is NULL or slot is not 0 / 1");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->copy_stream == nullptr) {
        WF_HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for(int i = 0; i < 2; ++i) {
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_consumed[i], hipEventDisableTiming));
        }
    }
    const size_t n = (size_t)count * h->cap_ch * frames;
    if(h->stage_async_floats[slot] < n) {
        if(h->slot_used[slot])
            WF_HIP_TRY(h, hipEventSynchronize(h->ev_consumed[slot])); // the old block may still feed an append
        dev_release(h, h->d_stage_async[slot]);
        h->d_stage_async[slot] = nullptr;
        const size_t want = grown(h->stage_async_floats[slot], n);
        h->stage_async_floats[slot] = 0;
        float *p = nullptr;
        rc = dev_alloc(h, &p, want);
        if(rc)
            return rc;
        h->d_stage_async[slot] = p;
        h->stage_async_floats[slot] = want;
    }
    // copy stream: wait until the previous append from this slot's staging block is done, then copy
    if(h->slot_used[slot])
        WF_HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->ev_consumed[slot], 0));
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_stage_async[slot], pinned_samples, n * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_copied[slot], h->copy_stream));
    // compute stream: the append waits for the copy; whatever is enqueued behind it (the tick) is ordered by the stream
    WF_HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_copied[slot], 0));
    rc = push_common(h, first, count, h->d_stage_async[slot], h->d_stage_async[slot], frames);
    if(rc)
        return rc;
    WF_HIP_TRY(h, hipEventRecord(h->ev_consumed[slot], h->stream));
    h->slot_used[slot] = true;
    return WF_HIP_OK;
}

int wf_hip_push_audio_ragged_async(wf_hip *h, uint32_t first, uint32_t count, const float *pinned_samples, const uint32_t *frames,
                                   uint32_t max_frames, uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(pinned_samples == nullptr || frames == nullptr || slot > 1 || max_frames == 0)
        return fail(h, WF_HIP_ERR_INVALID, "samples or frames is NULL, max_frames is 0 or slot is not 0 / 1");
    if(rms_follows_audio(h))
        return fail(h, WF_HIP_ERR_INVALID, "ragged pushes are not available while the device RMS producer follows the audio (wf_hip_enable_input_rms)");
    if(count > 65535u)
        return fail(h, WF_HIP_ERR_INVALID, "at most 65535 streams per ragged push");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->copy_stream == nullptr) {
        WF_HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for(int i = 0; i < 2; ++i) {
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_consumed[i], hipEventDisableTiming));
        }
    }
    const size_t n = (size_t)count * h->cap_ch * max_frames;
    if(h->slot_used[slot])
        WF_HIP_TRY(h, hipEventSynchronize(h->ev_consumed[slot])); // the slot's staging (samples and counts) is free again
    if(h->stage_async_floats[slot] < n) {
        dev_release(h, h->d_stage_async[slot]);
        h->d_stage_async[slot] = nullptr;
        const size_t want = grown(h->stage_async_floats[slot], n);
        h->stage_async_floats[slot] = 0;
        float *p = nullptr;
        rc = dev_alloc(h, &p, want);
        if(rc)
            return rc;
        h->d_stage_async[slot] = p;
        h->stage_async_floats[slot] = want;
    }
    if(h->frames_async_cap[slot] < count) {
        dev_release(h, h->d_frames_async[slot]);
        h->d_frames_async[slot] = nullptr;
        if(h->h_frames_async[slot])
            (void)hipHostFree(h->h_frames_async[slot]);
        h->h_frames_async[slot] = nullptr;
        h->frames_async_cap[slot] = 0;
        const size_t want = std::max<size_t>(count, 64);
        rc = dev_alloc(h, &h->d_frames_async[slot], want);
        if(rc)
            return rc;
        WF_HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_frames_async[slot]), want * sizeof(uint32_t), hipHostMallocDefault));
        h->frames_async_cap[slot] = want;
    }
    bool aligned = true;
    for(uint32_t i = 0; i < count; ++i) {
        h->h_frames_async[slot][i] = frames[i];
        aligned = aligned && (frames[i] % 4u) == 0;
    }
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_stage_async[slot], pinned_samples, n * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_frames_async[slot], h->h_frames_async[slot], (size_t)count * sizeof(uint32_t), hipMemcpyHostToDevice,
                                 h->copy_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_copied[slot], h->copy_stream));
    WF_HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_copied[slot], 0));
    hipLaunchKernelGGL(wf::ring_push_ragged_kernel, dim3(1, count), dim3(256), 0, h->stream, h->d_ring, h->d_wpos,
                       h->d_flags + (size_t)h->flag_cur * h->n_streams, h->ring_cap, h->ring_stride, h->cap_ch, first, h->d_stage_async[slot],
                       h->d_frames_async[slot], max_frames);
    WF_HIP_TRY(h, hipGetLastError());
    WF_HIP_TRY(h, hipEventRecord(h->ev_consumed[slot], h->stream));
    h->slot_used[slot] = true;
    if(!aligned)
        h->all_aligned = false;
    return WF_HIP_OK;
}

int wf_hip_ingest_done(wf_hip *h, uint32_t slot)
{
    if(h == nullptr || slot > 1)
        return WF_HIP_ERR_INVALID;
    if(!h->slot_used[slot] && !h->sq_slot_used[slot])
        return WF_HIP_OK;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_HIP_TRY(h, hipEventSynchronize(h->ev_copied[slot])); // the slot's last H2D copy (samples or squared peaks)
    return WF_HIP_OK;
}

void *wf_hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if(hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void wf_hip_host_free(void *p)
{
    if(p)
        (void)hipHostFree(p);
}

int wf_hip_push_audio_device(wf_hip *h, uint32_t first, uint32_t count, const float *d_samples, uint32_t frames)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(d_samples == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "d_samples is NULL");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    return push_common(h, first, count, d_samples, d_samples, frames);
}

int wf_hip_push_silence(wf_hip *h, uint32_t first, uint32_t count, uint32_t frames)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    return push_common(h, first, count, nullptr, nullptr, frames);
}

int wf_hip_push_synth(wf_hip *h, uint32_t first, uint32_t count, uint64_t seed, uint32_t stream_id0, uint64_t index0,
                      uint32_t frames)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(frames == 0)
        return WF_HIP_OK;
    if(rms_follows_audio(h) && frames > h->rms_cap)
        return fail(h, WF_HIP_ERR_INVALID, "push of %u frames exceeds the RMS ring capacity %u", frames, h->rms_cap);
    WF_HIP_TRY(h, hipSetDevice(h->device));
    const uint32_t gx = std::min<uint32_t>((frames + 255) / 256, 256);
    for(uint32_t off = 0; off < count; off += PUSH_SLICE) {
        const uint32_t cnt = std::min(PUSH_SLICE, count - off);
        hipLaunchKernelGGL(wf::ring_synth_kernel, dim3(gx, cnt * h->cap_ch), dim3(256), 0, h->stream, h->d_ring, h->d_wpos,
                           h->ring_cap, h->ring_stride, h->cap_ch, first + off, seed, stream_id0 + off, index0, frames);
        if(rms_follows_audio(h)) {
            hipLaunchKernelGGL(wf::rms_synth_kernel, dim3(gx, cnt), dim3(256), 0, h->stream, h->d_rms_ring, h->d_wpos, h->rms_cap,
                               h->cap_ch, first + off, seed, stream_id0 + off, index0, frames);
            rms_after_push(h, first + off, cnt, frames);
        }
    }
    hipLaunchKernelGGL(wf::wpos_advance_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->d_wpos,
                       h->d_flags + (size_t)h->flag_cur * h->n_streams, first, count, frames);
    WF_HIP_TRY(h, hipGetLastError());
    if(frames % 4u)
        h->all_aligned = false;
    return WF_HIP_OK;
}

int wf_hip_tick(wf_hip *h, const wf_hip_tick_params *p)
{
    if(h == nullptr || p == nullptr)
        return WF_HIP_ERR_INVALID;
    if((uint64_t)p->delay_frames + h->max_stream_delay + (h->wave ? h->wave_samples : h->N) > h->ring_cap)
        return fail(h, WF_HIP_ERR_INVALID, "delay_frames %u (+ per-stream %u) + fft_size %u exceeds the ring capacity %u", p->delay_frames,
                    h->max_stream_delay, h->N, h->ring_cap);
    if((p->flags & WF_HIP_TICK_NO_DECIBELS) && h->num_bars == 0)
        return fail(h, WF_HIP_ERR_INVALID, "WF_HIP_TICK_NO_DECIBELS on a configuration without bars or curve: the tick would produce nothing");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->wave) {
        WF_TRY_RC(wait_rows_in_flight(h)); // the waveform rows are read back the same way
        launch_input_rms(h, p);
        wf::WaveArgs w{};
        w.ring = h->d_ring;
        w.wpos = h->d_wpos;
        w.cend = h->d_cend;
        w.wts = h->d_wts;
        w.ring_mask = h->ring_cap - 1;
        w.ring_stride = h->ring_stride;
        w.delay = p->delay_frames;
        w.delay_stream = h->d_delay;
        w.rows = h->d_decibels;
        w.stream_flags = h->d_flags;
        w.audio_ts = p->audio_ts_ns;
        w.audio_ts_stream = h->d_audio_ts;
        w.step_ns = ((unsigned long long)h->cfg.meter_ms * 1000000ull) / h->N; // src/source_generic.cpp:299
        w.waveform_samples = h->wave_samples;
        w.width = h->N;
        w.sample_rate = h->cfg.sample_rate;
        w.n_streams = h->n_streams;
        w.cap_ch = h->cap_ch;
        w.out_ch = h->out_ch;
        w.stereo = h->cfg.stereo ? 1u : 0u;
        w.normalize = h->cfg.normalize_volume ? 1u : 0u;
        if(w.normalize) {
            const float rms_db = (p->input_rms > 0.0f) ? 20.0f * std::log10(p->input_rms) : wf::db_min();
            w.vol_comp = std::min(h->cfg.volume_target - rms_db, h->cfg.max_gain); // src/source_generic.cpp:381
            w.vol_comp_stream = h->d_vol_comp;
        }
        w.db_min = wf::db_min();
        hipLaunchKernelGGL(wf::waveform_tick_kernel, dim3((h->n_streams + wf::WAVE_STREAMS - 1) / wf::WAVE_STREAMS), dim3(wf::WAVE_THREADS), 0, h->stream, w);
        WF_HIP_TRY(h, hipGetLastError());
        return WF_HIP_OK;
    }
    if(h->meter) {
        const wf::MeterArgs m = make_meter_args(h, p);
        hipLaunchKernelGGL(wf::meter_tick_kernel, dim3(h->n_streams), dim3(wf::METER_THREADS), 0, h->stream, m);
        WF_HIP_TRY(h, hipGetLastError());
        return WF_HIP_OK;
    }
    const bool mono_mix_rows = !h->cfg.stereo && h->cap_ch > 1;
    wf_hip_tick_params p_rows;
    if((p->flags & WF_HIP_TICK_NO_DECIBELS) && (h->big_l || h->ext_outputs)) {
        // the outputs of this batch are derived from the stored rows (big_outputs_kernel): the rows are stored regardless --
        // the flag only ever promised that they MAY be stale
        p_rows = *p;
        p_rows.flags &= ~WF_HIP_TICK_NO_DECIBELS;
        p = &p_rows;
    }
    if((p->flags & WF_HIP_TICK_NO_DECIBELS) && !mono_mix_rows && h->d_stale_row == nullptr) {
        if(h->cfg.floor_db - 10 >= 0)
            return fail(h, WF_HIP_ERR_INVALID, "WF_HIP_TICK_NO_DECIBELS needs floor_db < 10 (a skipped channel's row must be negative)");
        int rc = dev_alloc(h, &h->d_stale_row, (size_t)h->M);
        if(rc)
            return rc;
        hipLaunchKernelGGL(wf::fill_f32_kernel, dim3((h->M + 255) / 256), dim3(256), 0, h->stream, h->d_stale_row, (size_t)h->M, wf::db_min());
        if(!h->split) {
            rc = dev_alloc(h, &h->d_row_verdict, (size_t)h->n_streams * h->cap_ch * h->waves_per_spectrum);
            if(rc)
                return rc;
        }
        rc = dev_alloc(h, &h->d_bars_only, 1);
        if(rc)
            return rc;
        const wf::BarsOnlyState st{h->d_row_verdict, h->d_stale_row, 0u}; // this tick's rows are still current
        WF_HIP_TRY(h, hipMemcpyAsync(h->d_bars_only, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
        WF_HIP_TRY(h, hipStreamSynchronize(h->stream)); // `st` dies here
        h->main_dirty = true;
    }
    WF_TRY_RC(wait_rows_in_flight(h)); // this tick's row stores wait for a readback still in flight
    if(h->d_rms_ring)
        WF_TRY_RC(join_lanes(h)); // (never pending: the RMS producer keeps the batch on one lane)
    launch_input_rms(h, p);
    wf::TickArgs a = make_args(h, p);
    const bool aligned = h->all_aligned && h->stream_delays_aligned && (p->delay_frames % 4u) == 0;
    const int lanes = h->d_rms_ring ? 1 : h->n_lanes; // update_input_rms runs on `stream` ahead of every tick: one lane
    if(lanes > 1 && h->main_dirty) {
        WF_HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
        for(int l = 1; l < lanes; ++l)
            WF_HIP_TRY(h, hipStreamWaitEvent(h->lane_stream[l], h->ev_fork, 0));
    }
    h->main_dirty = false;
    for(int l = 0; l < lanes; ++l) {
        const uint32_t lo = (uint32_t)((uint64_t)h->n_streams * l / lanes), hi = (uint32_t)((uint64_t)h->n_streams * (l + 1) / lanes);
        a.stream_base = lo;
        a.stream_count = hi - lo;
        h->launch_stream = l == 0 ? h->stream : h->lane_stream[l];
        h->launch_rc = WF_HIP_OK;
        if(h->ext_outputs) {
            // the tick kernel stores rows only; the display comes from them, one workgroup per displayed row
            wf::TickArgs rows_only = a;
            rows_only.bar.out = nullptr;
            h->launch(h, rows_only, aligned);
            if(hi > lo)
                hipLaunchKernelGGL(wf::big_outputs_kernel, dim3((hi - lo) * h->disp_ch), dim3(wf::GBig::T), h->big_out_lds, h->launch_stream, a);
        } else
            h->launch(h, a, aligned);
        if(h->launch_rc != WF_HIP_OK)
            return h->launch_rc;
        if(h->d_verts && hi > lo) { // the vertex fill of this slice, behind its bars
            wf::VertexArgs v{};
            v.bars = h->d_bars;
            v.verts = h->d_verts;
            v.cap_xy = h->d_cap_xy;
            v.stream_base = lo;
            v.stream_count = hi - lo;
            v.disp_ch = h->disp_ch;
            v.num_bars = (int)h->num_bars;
            v.per_row = h->vtab.per_row;
            v.per_bar = h->vtab.per_bar;
            v.mode = h->vtab.mode;
            v.bar_stride = h->vtab.bar_stride;
            v.bar_width = h->cfg.bar_width;
            v.cpos = h->vtab.cpos;
            v.bottom = h->vtab.bottom;
            v.channel_offset = h->vtab.channel_offset;
            v.cap_radius = h->vtab.cap_radius;
            v.rounded = h->cfg.rounded_caps ? 1 : 0;
            v.cap_tris = h->vtab.cap_tris;
            v.bottom_caps = h->vtab.bottom_caps;
            v.radial = h->vtab.radial;
            v.bot_offset = h->vtab.bot_offset;
            v.step_width = h->cfg.step_width;
            v.step_stride = h->vtab.step_stride;
            v.max_steps = h->vtab.max_steps;
            v.counts = h->d_vert_counts;
            hipLaunchKernelGGL(wf::vertex_fill_kernel, dim3((hi - lo) * h->disp_ch), dim3(256), 0, h->launch_stream, v);
        }
        if(l > 0)
            WF_HIP_TRY(h, hipEventRecord(h->ev_lane[l], h->lane_stream[l]));
    }
    if(lanes > 1)
        h->lanes_pending = true;
    WF_HIP_TRY(h, hipGetLastError());
    if(h->d_row_verdict && !h->verdict_tracking) {
        // this tick left a verdict for every row; later ticks read those (the flag flips behind this tick's kernels, on every lane)
        h->verdict_tracking = true;
        WF_TRY_RC(join_lanes(h));
        const wf::BarsOnlyState st{h->d_row_verdict, h->d_stale_row, 1u};
        WF_HIP_TRY(h, hipMemcpyAsync(h->d_bars_only, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
        WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if(h->split)
        h->flag_cur = (h->flag_cur + 1) % 3; // what the kernel wrote is what the next tick (and the readers) see
    return WF_HIP_OK;
}

int wf_hip_set_hidden(wf_hip *h, uint32_t first, uint32_t count, const uint8_t *mask)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(mask == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "mask is NULL");
    if(h->meter || h->wave)
        for(uint32_t i = 0; i < count; ++i)
            if(mask[i] == WF_HIP_STARVED)
                return fail(h, WF_HIP_ERR_INVALID, "WF_HIP_STARVED applies to spectrum batches only");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->mask_bytes < count) {
        rc = dev_alloc(h, &h->d_mask, (size_t)count);
        if(rc)
            return rc;
        h->mask_bytes = count;
    }
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_mask, mask, count, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(wf::set_hidden_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream,
                       h->d_flags + (size_t)h->flag_cur * h->n_streams, first, count, h->d_mask);
    WF_HIP_TRY(h, hipGetLastError());
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream)); // `mask` is borrowed for the call only
    return WF_HIP_OK;
}

int wf_hip_set_stream_delay(wf_hip *h, uint32_t first, uint32_t count, const uint32_t *delay_frames)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(delay_frames == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "delay_frames is NULL");
    uint32_t mx = 0;
    bool al = true;
    for(uint32_t i = 0; i < count; ++i) {
        mx = std::max(mx, delay_frames[i]);
        al = al && (delay_frames[i] % 4u) == 0;
    }
    const uint32_t window = h->wave ? h->wave_samples : h->N; // the same capacity term as wf_hip_tick
    if((uint64_t)mx + window > h->ring_cap)
        return fail(h, WF_HIP_ERR_INVALID, "stream delay %u + window %u exceeds the ring capacity %u", mx, window, h->ring_cap);
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->d_delay == nullptr) {
        rc = dev_alloc(h, &h->d_delay, (size_t)h->n_streams);
        if(rc)
            return rc;
        WF_HIP_TRY(h, hipMemsetAsync(h->d_delay, 0, (size_t)h->n_streams * sizeof(uint32_t), h->stream));
    }
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_delay + first, delay_frames, (size_t)count * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream)); // `delay_frames` is borrowed for the call only
    h->max_stream_delay = std::max(h->max_stream_delay, mx);
    h->stream_delays_aligned = h->stream_delays_aligned && al; // conservative: never switches back to the vector fetch
    return WF_HIP_OK;
}

int wf_hip_set_stream_audio_ts(wf_hip *h, uint32_t first, uint32_t count, const uint64_t *audio_ts_ns)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(!h->wave)
        return fail(h, WF_HIP_ERR_INVALID, "per-stream audio timestamps belong to waveform batches");
    if(audio_ts_ns == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "audio_ts_ns is NULL");
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "64-bit timestamps");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->d_audio_ts == nullptr) {
        rc = dev_alloc(h, &h->d_audio_ts, (size_t)h->n_streams);
        if(rc)
            return rc;
        WF_HIP_TRY(h, hipMemsetAsync(h->d_audio_ts, 0, (size_t)h->n_streams * sizeof(unsigned long long), h->stream));
    }
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_audio_ts + first, audio_ts_ns, (size_t)count * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream)); // `audio_ts_ns` is borrowed for the call only
    return WF_HIP_OK;
}

int wf_hip_set_input_rms(wf_hip *h, uint32_t first, uint32_t count, const float *rms)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(rms == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "rms is NULL");
    if(h->d_rms_ring)
        return fail(h, WF_HIP_ERR_INVALID, "m_input_rms is produced on the device (wf_hip_enable_input_rms); it cannot be set");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    // volume_compensation of every stream, reference src/source_generic.cpp:163 with dbfs() of src/source.hpp:293-299
    auto comp = [&](float r) {
        const float rms_db = (r > 0.0f) ? 20.0f * std::log10(r) : wf::db_min();
        return std::min(h->cfg.volume_target - rms_db, h->cfg.max_gain);
    };
    if(h->d_vol_comp == nullptr) {
        rc = dev_alloc(h, &h->d_vol_comp, (size_t)h->n_streams);
        if(rc)
            return rc;
        const std::vector<float> init(h->n_streams, comp(0.0f));
        WF_HIP_TRY(h, hipMemcpyAsync(h->d_vol_comp, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    std::vector<float> v(count);
    for(uint32_t i = 0; i < count; ++i)
        v[i] = comp(rms[i]);
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_vol_comp + first, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream)); // the staging vector dies here
    return WF_HIP_OK;
}

static int read_back(wf_hip *h, const void *d, void *out, size_t bytes);

static int enable_rms_producer(wf_hip *h, bool feed)
{
    if(h == nullptr)
        return WF_HIP_ERR_INVALID;
    if(!h->cfg.normalize_volume || h->meter)
        return fail(h, WF_HIP_ERR_INVALID, "the device RMS producer needs a spectrum or waveform batch with cfg.normalize_volume");
    if(h->d_rms_ring)
        return h->rms_feed == feed ? WF_HIP_OK : fail(h, WF_HIP_ERR_INVALID, "the device RMS producer is already enabled in the other mode");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_TRY_RC(join_lanes(h));
    h->rms_size = h->cfg.sample_rate & ~15u; // m_input_rms_size, src/source.cpp:1147
    if(h->rms_size == 0)
        return fail(h, WF_HIP_ERR_INVALID, "sample_rate %u is too small for the RMS window", h->cfg.sample_rate);
    // the window + every A/V-sync delay the audio rings admit + the two ragged blocks at its ends (feed mode: the window,
    // the ragged blocks and one feed of up to a window's length)
    h->rms_cap = feed ? next_pow2(2 * h->rms_size + 2 * wf::RMS_BLOCK)
                      : next_pow2(h->rms_size + (h->wave ? h->ring_cap : h->ring_cap - h->N) + 2 * wf::RMS_BLOCK);
    const size_t nblk = h->rms_cap / wf::RMS_BLOCK;
    float *ring = nullptr, *bsum = nullptr;
    int rc = dev_alloc(h, &ring, (size_t)h->n_streams * h->rms_cap);
    if(rc == WF_HIP_OK) rc = dev_alloc(h, &bsum, (size_t)h->n_streams * nblk);
    if(rc == WF_HIP_OK) rc = dev_alloc(h, &h->d_rend, (size_t)h->n_streams);
    if(rc == WF_HIP_OK) rc = dev_alloc(h, &h->d_input_rms, (size_t)h->n_streams);
    if(rc == WF_HIP_OK && h->d_vol_comp == nullptr) rc = dev_alloc(h, &h->d_vol_comp, (size_t)h->n_streams);
    if(rc)
        return rc;
    // audio captured before this call counts as silence (m_input_rms_buf starts as zeros); the first tick's kernel
    // fills d_vol_comp before the spectrum kernel reads it
    WF_HIP_TRY(h, hipMemsetAsync(ring, 0, (size_t)h->n_streams * h->rms_cap * sizeof(float), h->stream));
    WF_HIP_TRY(h, hipMemsetAsync(bsum, 0, (size_t)h->n_streams * nblk * sizeof(float), h->stream));
    WF_HIP_TRY(h, hipMemsetAsync(h->d_rend, 0, (size_t)h->n_streams * sizeof(uint32_t), h->stream));
    WF_HIP_TRY(h, hipMemsetAsync(h->d_input_rms, 0, (size_t)h->n_streams * sizeof(float), h->stream));
    h->d_rms_bsum = bsum;
    h->rms_feed = feed;
    h->d_rms_ring = ring; // from here on every push feeds it (or, feed mode, wf_hip_push_rms_ragged_async does)
    h->main_dirty = true;
    return WF_HIP_OK;
}

int wf_hip_enable_input_rms(wf_hip *h) { return enable_rms_producer(h, false); }
int wf_hip_enable_input_rms_feed(wf_hip *h) { return enable_rms_producer(h, true); }

int wf_hip_push_rms_ragged_async(wf_hip *h, uint32_t first, uint32_t count, const float *pinned_sq, const uint32_t *frames, uint32_t max_frames,
                                 uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(pinned_sq == nullptr || frames == nullptr || slot > 1 || max_frames == 0)
        return fail(h, WF_HIP_ERR_INVALID, "values or frames is NULL, max_frames is 0 or slot is not 0 / 1");
    if(!h->rms_feed)
        return fail(h, WF_HIP_ERR_INVALID, "wf_hip_push_rms_ragged_async needs wf_hip_enable_input_rms_feed");
    if(max_frames > h->rms_size)
        return fail(h, WF_HIP_ERR_INVALID, "a feed of %u values per stream exceeds the RMS window (%u)", max_frames, h->rms_size);
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->copy_stream == nullptr) {
        WF_HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for(int i = 0; i < 2; ++i) {
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_consumed[i], hipEventDisableTiming));
        }
    }
    if(h->ev_sq_consumed[0] == nullptr)
        for(int i = 0; i < 2; ++i)
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_sq_consumed[i], hipEventDisableTiming));
    if(h->sq_slot_used[slot])
        WF_HIP_TRY(h, hipEventSynchronize(h->ev_sq_consumed[slot])); // the slot's staging is free again (two feeds ago)
    const size_t n = (size_t)count * max_frames;
    if(h->sq_stage_floats[slot] < n) {
        dev_release(h, h->d_sq_stage[slot]);
        h->d_sq_stage[slot] = nullptr;
        const size_t want = grown(h->sq_stage_floats[slot], n);
        h->sq_stage_floats[slot] = 0;
        float *p = nullptr;
        rc = dev_alloc(h, &p, want);
        if(rc)
            return rc;
        h->d_sq_stage[slot] = p;
        h->sq_stage_floats[slot] = want;
    }
    if(h->sq_frames_cap[slot] < count) {
        dev_release(h, h->d_sq_frames[slot]);
        h->d_sq_frames[slot] = nullptr;
        if(h->h_sq_frames[slot])
            (void)hipHostFree(h->h_sq_frames[slot]);
        h->h_sq_frames[slot] = nullptr;
        h->sq_frames_cap[slot] = 0;
        const size_t want = std::max<size_t>(count, 64);
        rc = dev_alloc(h, &h->d_sq_frames[slot], want);
        if(rc)
            return rc;
        WF_HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_sq_frames[slot]), want * sizeof(uint32_t), hipHostMallocDefault));
        h->sq_frames_cap[slot] = want;
    }
    uint32_t longest = 0;
    for(uint32_t i = 0; i < count; ++i) {
        h->h_sq_frames[slot][i] = std::min(frames[i], max_frames);
        longest = std::max(longest, h->h_sq_frames[slot][i]);
    }
    if(longest == 0)
        return WF_HIP_OK; // nothing to consume this frame (sync_rms_buffer returns false for every stream)
    // rows are max_frames apart; only the part any stream uses crosses the bus when the rows are short
    if(longest == max_frames || count == 1)
        WF_HIP_TRY(h, hipMemcpyAsync(h->d_sq_stage[slot], pinned_sq, (count == 1 ? (size_t)longest : n) * sizeof(float), hipMemcpyHostToDevice,
                                     h->copy_stream));
    else
        WF_HIP_TRY(h, hipMemcpy2DAsync(h->d_sq_stage[slot], (size_t)max_frames * sizeof(float), pinned_sq, (size_t)max_frames * sizeof(float),
                                       (size_t)longest * sizeof(float), count, hipMemcpyHostToDevice, h->copy_stream));
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_sq_frames[slot], h->h_sq_frames[slot], (size_t)count * sizeof(uint32_t), hipMemcpyHostToDevice,
                                 h->copy_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_copied[slot], h->copy_stream));
    WF_TRY_RC(join_lanes(h));
    WF_HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_copied[slot], 0));
    hipLaunchKernelGGL(wf::rms_feed_ragged_kernel, dim3(count), dim3(256), 0, h->stream, h->d_rms_ring, h->d_rms_bsum, h->d_rend, h->rms_cap, first,
                       h->d_sq_stage[slot], h->d_sq_frames[slot], max_frames);
    WF_HIP_TRY(h, hipGetLastError());
    WF_HIP_TRY(h, hipEventRecord(h->ev_sq_consumed[slot], h->stream));
    h->sq_slot_used[slot] = true;
    h->main_dirty = true;
    return WF_HIP_OK;
}

int wf_hip_read_input_rms_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_out, uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(pinned_out == nullptr || slot > 1)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer is NULL or slot is not 0 / 1");
    if(h->d_input_rms == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "the device RMS producer is not enabled");
    if(!h->rows_in_flight[slot] || h->read_stream == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "wf_hip_read_input_rms_async rides on the slot's wf_hip_read_rows_async: call that first");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    // behind the rows' copy on the readback stream (which already waits for the tick); completes with wf_hip_readback_done(slot)
    WF_HIP_TRY(h, hipMemcpyAsync(pinned_out, h->d_input_rms + first, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, h->read_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_read[slot], h->read_stream));
    return WF_HIP_OK;
}

int wf_hip_read_input_rms(wf_hip *h, uint32_t first, uint32_t count, float *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->d_input_rms == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "the device RMS producer is not enabled (wf_hip_enable_input_rms)");
    return read_back(h, h->d_input_rms + first, out, (size_t)count * sizeof(float));
}

int wf_hip_sync(wf_hip *h)
{
    if(h == nullptr)
        return WF_HIP_ERR_INVALID;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_TRY_RC(join_lanes(h));
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
    return WF_HIP_OK;
}

static int read_back(wf_hip *h, const void *d, void *out, size_t bytes)
{
    if(out == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer is NULL");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_HIP_TRY(h, hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, h->stream));
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
    return WF_HIP_OK;
}

int wf_hip_read_decibels(wf_hip *h, uint32_t first, uint32_t count, float *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->meter)
        return fail(h, WF_HIP_ERR_INVALID, "meter batch: there is no m_decibels; read the levels with wf_hip_read_meter");
    const size_t per = (size_t)h->out_ch * h->M;
    return read_back(h, h->d_decibels + first * per, out, count * per * sizeof(float));
}

int wf_hip_read_bars(wf_hip *h, uint32_t first, uint32_t count, float *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->d_bars == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "configuration has no bars (cfg.bars == 0)");
    const size_t per = (size_t)h->disp_ch * h->num_bars;
    return read_back(h, h->d_bars + first * per, out, count * per * sizeof(float));
}

uint32_t wf_hip_num_vertices(const wf_hip *h) { return (h && h->d_verts) ? (uint32_t)h->vtab.per_row : 0u; }

int wf_hip_read_vertices(wf_hip *h, uint32_t first, uint32_t count, float *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->d_verts == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "configuration has no vertex fill (cfg.vertices == 0)");
    const size_t per = (size_t)h->disp_ch * h->vtab.per_row;
    return read_back(h, h->d_verts + first * per, out, count * per * sizeof(wf::f4));
}

int wf_hip_read_vertex_counts(wf_hip *h, uint32_t first, uint32_t count, uint32_t *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->d_vert_counts == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "configuration has no vertex fill (cfg.vertices == 0)");
    return read_back(h, h->d_vert_counts + (size_t)first * h->disp_ch, out, (size_t)count * h->disp_ch * sizeof(uint32_t));
}

const float *wf_hip_vertices_device(wf_hip *h)
{
    if(h == nullptr || h->d_verts == nullptr)
        return nullptr;
    (void)join_lanes(h);
    return reinterpret_cast<const float *>(h->d_verts);
}

int wf_hip_read_bars_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_out, uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->d_bars == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "configuration has no bars (cfg.bars == 0)");
    if(pinned_out == nullptr || slot > 1)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer is NULL or slot is not 0 / 1");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->read_stream == nullptr) {
        WF_HIP_TRY(h, hipStreamCreateWithFlags(&h->read_stream, hipStreamNonBlocking));
        for(int i = 0; i < 2; ++i) {
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_snap[i], hipEventDisableTiming));
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_read[i], hipEventDisableTiming));
        }
    }
    const size_t per = (size_t)h->disp_ch * h->num_bars, n = count * per;
    if(h->read_used[slot])
        WF_HIP_TRY(h, hipEventSynchronize(h->ev_read[slot])); // the slot's previous copy must have left its snapshot
    if(h->snap_floats[slot] < n) { // (the slot's previous copy has left its snapshot: waited for above)
        dev_release(h, h->d_snap[slot]);
        h->d_snap[slot] = nullptr;
        const size_t want = grown(h->snap_floats[slot], n);
        h->snap_floats[slot] = 0;
        float *p = nullptr;
        rc = dev_alloc(h, &p, want);
        if(rc)
            return rc;
        h->d_snap[slot] = p;
        h->snap_floats[slot] = want;
    }
    // compute stream: snapshot behind the ticks enqueued so far (device to device, a few MB at most)
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_snap[slot], h->d_bars + first * per, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_snap[slot], h->stream));
    // readback stream: the D2H copy of the snapshot; later ticks do not wait for it
    WF_HIP_TRY(h, hipStreamWaitEvent(h->read_stream, h->ev_snap[slot], 0));
    WF_HIP_TRY(h, hipMemcpyAsync(pinned_out, h->d_snap[slot], n * sizeof(float), hipMemcpyDeviceToHost, h->read_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_read[slot], h->read_stream));
    h->read_used[slot] = true;
    return WF_HIP_OK;
}

// m_last_silent of streams [first, first+count) as bytes (for the D2H copy of wf_hip_read_rows_async)
__global__ void silent_bytes_kernel(const uint32_t *flags, uint32_t first, uint32_t count, uint8_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < count)
        out[i] = (flags[first + i] & wf::WF_STREAM_LAST_SILENT) ? 1 : 0;
}

int wf_hip_read_rows_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_rows, uint8_t *pinned_last_silent, uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->meter)
        return fail(h, WF_HIP_ERR_INVALID, "meter batch: there is no m_decibels");
    if(pinned_rows == nullptr || pinned_last_silent == nullptr || slot > 1)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer is NULL or slot is not 0 / 1");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->read_stream == nullptr) {
        WF_HIP_TRY(h, hipStreamCreateWithFlags(&h->read_stream, hipStreamNonBlocking));
        for(int i = 0; i < 2; ++i) {
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_snap[i], hipEventDisableTiming));
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_read[i], hipEventDisableTiming));
        }
    }
    if(h->read_used[slot])
        WF_HIP_TRY(h, hipEventSynchronize(h->ev_read[slot])); // the slot's previous copy has landed
    if(h->silent_bytes_cap[slot] < count) {
        dev_release(h, h->d_silent_bytes[slot]);
        h->d_silent_bytes[slot] = nullptr;
        h->silent_bytes_cap[slot] = 0;
        const size_t want = std::max<size_t>(count, 256);
        rc = dev_alloc(h, &h->d_silent_bytes[slot], want);
        if(rc)
            return rc;
        h->silent_bytes_cap[slot] = want;
    }
    hipLaunchKernelGGL(silent_bytes_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->d_flags + (size_t)h->flag_cur * h->n_streams,
                       first, count, h->d_silent_bytes[slot]);
    WF_HIP_TRY(h, hipGetLastError());
    WF_HIP_TRY(h, hipEventRecord(h->ev_snap[slot], h->stream));
    WF_HIP_TRY(h, hipStreamWaitEvent(h->read_stream, h->ev_snap[slot], 0));
    const size_t per = (size_t)h->out_ch * h->M;
    WF_HIP_TRY(h, hipMemcpyAsync(pinned_rows, h->d_decibels + first * per, count * per * sizeof(float), hipMemcpyDeviceToHost, h->read_stream));
    WF_HIP_TRY(h, hipMemcpyAsync(pinned_last_silent, h->d_silent_bytes[slot], count, hipMemcpyDeviceToHost, h->read_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_read[slot], h->read_stream));
    h->read_used[slot] = true;
    h->rows_in_flight[slot] = true;
    return WF_HIP_OK;
}

int wf_hip_read_meter_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_levels, uint8_t *pinned_last_silent, uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(!h->meter)
        return fail(h, WF_HIP_ERR_INVALID, "not a meter batch (cfg.meter == 0)");
    if(pinned_levels == nullptr || pinned_last_silent == nullptr || slot > 1)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer is NULL or slot is not 0 / 1");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    if(h->read_stream == nullptr) {
        WF_HIP_TRY(h, hipStreamCreateWithFlags(&h->read_stream, hipStreamNonBlocking));
        for(int i = 0; i < 2; ++i) {
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_snap[i], hipEventDisableTiming));
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_read[i], hipEventDisableTiming));
        }
    }
    const size_t n = (size_t)count * h->cap_ch;
    if(h->read_used[slot])
        WF_HIP_TRY(h, hipEventSynchronize(h->ev_read[slot])); // the slot's previous copy has left its snapshot
    if(h->snap_floats[slot] < n) {
        dev_release(h, h->d_snap[slot]);
        h->d_snap[slot] = nullptr;
        const size_t want = std::max<size_t>(grown(h->snap_floats[slot], n), 64);
        h->snap_floats[slot] = 0;
        float *p = nullptr;
        rc = dev_alloc(h, &p, want);
        if(rc)
            return rc;
        h->d_snap[slot] = p;
        h->snap_floats[slot] = want;
    }
    if(h->silent_bytes_cap[slot] < count) {
        dev_release(h, h->d_silent_bytes[slot]);
        h->d_silent_bytes[slot] = nullptr;
        h->silent_bytes_cap[slot] = 0;
        const size_t want = std::max<size_t>(count, 256);
        rc = dev_alloc(h, &h->d_silent_bytes[slot], want);
        if(rc)
            return rc;
        h->silent_bytes_cap[slot] = want;
    }
    // compute stream: a snapshot of the few floats behind the ticks enqueued so far (the next tick overwrites m_meter_val)
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_snap[slot], h->d_meter_val + (size_t)first * h->cap_ch, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    hipLaunchKernelGGL(silent_bytes_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->d_flags, first, count, h->d_silent_bytes[slot]);
    WF_HIP_TRY(h, hipGetLastError());
    WF_HIP_TRY(h, hipEventRecord(h->ev_snap[slot], h->stream));
    WF_HIP_TRY(h, hipStreamWaitEvent(h->read_stream, h->ev_snap[slot], 0));
    WF_HIP_TRY(h, hipMemcpyAsync(pinned_levels, h->d_snap[slot], n * sizeof(float), hipMemcpyDeviceToHost, h->read_stream));
    WF_HIP_TRY(h, hipMemcpyAsync(pinned_last_silent, h->d_silent_bytes[slot], count, hipMemcpyDeviceToHost, h->read_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_read[slot], h->read_stream));
    h->read_used[slot] = true;
    return WF_HIP_OK;
}

int wf_hip_readback_done(wf_hip *h, uint32_t slot)
{
    if(h == nullptr || slot > 1)
        return WF_HIP_ERR_INVALID;
    if(!h->read_used[slot])
        return WF_HIP_OK;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_HIP_TRY(h, hipEventSynchronize(h->ev_read[slot]));
    return WF_HIP_OK;
}

int wf_hip_copy_bars_device(wf_hip *h, uint32_t first, uint32_t count, void *d_out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->d_bars == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "configuration has no bars (cfg.bars == 0)");
    if(d_out == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer is NULL");
    const size_t per = (size_t)h->disp_ch * h->num_bars;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_HIP_TRY(h, hipMemcpyAsync(d_out, h->d_bars + first * per, count * per * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
    return WF_HIP_OK;
}

int wf_hip_copy_bars_device_async(wf_hip *h, uint32_t first, uint32_t count, void *d_out, void *consumer_stream)
{
    if(h == nullptr)
        return WF_HIP_ERR_INVALID;
    if(count == 0 || first >= h->n_streams || count > h->n_streams - first)
        return fail(h, WF_HIP_ERR_INVALID, "stream range [%u, %u+%u) outside 0..%u", first, first, count, h->n_streams);
    if(h->d_bars == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "configuration has no bars (cfg.bars == 0)");
    if(d_out == nullptr || consumer_stream == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer or consumer stream is NULL");
    const size_t per = (size_t)h->disp_ch * h->num_bars;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t cs = static_cast<hipStream_t>(consumer_stream);
    // The lanes are NOT joined: a join would put the next tick's lanes behind the handle's stream again and take away the
    // overlap of one tick's tail with the next one's head (measured: 110 -> 143 us per tick at 8192 streams).  Every lane
    // copies the bars of its own slice on its own stream, behind the tick it has just run and in front of its next one.
    const int lanes = h->lanes_pending ? h->n_lanes : 1;
    for(int l = 0; l < lanes; ++l) {
        const uint32_t lo = lanes == 1 ? 0u : (uint32_t)((uint64_t)h->n_streams * l / lanes);
        const uint32_t hi = lanes == 1 ? h->n_streams : (uint32_t)((uint64_t)h->n_streams * (l + 1) / lanes);
        const uint32_t a = std::max(lo, first), b = std::min(hi, first + count);
        if(a >= b)
            continue;
        hipStream_t st = l == 0 ? h->stream : h->lane_stream[l];
        if(h->ev_bars_lane[l] == nullptr)
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_bars_lane[l], hipEventDisableTiming));
        WF_HIP_TRY(h, hipMemcpyAsync(static_cast<float *>(d_out) + (size_t)(a - first) * per, h->d_bars + (size_t)a * per,
                                     (size_t)(b - a) * per * sizeof(float), hipMemcpyDeviceToDevice, st));
        WF_HIP_TRY(h, hipEventRecord(h->ev_bars_lane[l], st));
        WF_HIP_TRY(h, hipStreamWaitEvent(cs, h->ev_bars_lane[l], 0));
    }
    return WF_HIP_OK;
}

int wf_hip_wait_event(wf_hip *h, void *event)
{
    if(h == nullptr || event == nullptr)
        return WF_HIP_ERR_INVALID;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    hipEvent_t ev = static_cast<hipEvent_t>(event);
    WF_HIP_TRY(h, hipStreamWaitEvent(h->stream, ev, 0));
    for(int l = 1; l < h->n_lanes; ++l)
        if(h->lane_stream[l])
            WF_HIP_TRY(h, hipStreamWaitEvent(h->lane_stream[l], ev, 0));
    return WF_HIP_OK;
}

int wf_hip_read_meter(wf_hip *h, uint32_t first, uint32_t count, float *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(!h->meter)
        return fail(h, WF_HIP_ERR_INVALID, "not a meter batch (cfg.meter == 0)");
    return read_back(h, h->d_meter_val + (size_t)first * h->cap_ch, out, (size_t)count * h->cap_ch * sizeof(float));
}

int wf_hip_read_tsmooth(wf_hip *h, uint32_t first, uint32_t count, float *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(h->meter || h->wave)
        return fail(h, WF_HIP_ERR_INVALID, "meter / waveform batch: there is no m_tsmooth_buf");
    const size_t per = (size_t)h->cap_ch * h->M;
    return read_back(h, h->d_tsmooth + first * per, out, count * per * sizeof(float));
}

int wf_hip_write_tsmooth(wf_hip *h, uint32_t first, uint32_t count, const float *in)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(in == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "input pointer is NULL");
    if(h->meter || h->wave)
        return fail(h, WF_HIP_ERR_INVALID, "meter / waveform batch: there is no m_tsmooth_buf");
    const size_t per = (size_t)h->cap_ch * h->M;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_HIP_TRY(h, hipMemcpyAsync(h->d_tsmooth + first * per, in, count * per * sizeof(float), hipMemcpyHostToDevice, h->stream));
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
    return WF_HIP_OK;
}

int wf_hip_read_last_silent(wf_hip *h, uint32_t first, uint32_t count, uint8_t *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    std::vector<uint32_t> tmp(count);
    rc = read_back(h, h->d_flags + (size_t)h->flag_cur * h->n_streams + first, tmp.data(), count * sizeof(uint32_t));
    if(rc)
        return rc;
    for(uint32_t i = 0; i < count; ++i)
        out[i] = (tmp[i] & wf::WF_STREAM_LAST_SILENT) ? 1 : 0;
    return WF_HIP_OK;
}

float *wf_hip_decibels_device(wf_hip *h) { return h ? h->d_decibels : nullptr; }
float *wf_hip_bars_device(wf_hip *h) { return h ? h->d_bars : nullptr; }
void *wf_hip_stream(wf_hip *h)
{
    if(h == nullptr)
        return nullptr;
    (void)join_lanes(h); // whatever the caller orders behind this stream is ordered behind every tick enqueued so far
    return (void *)h->stream;
}

size_t wf_hip_table_window(const wf_hip *h, const float **out, float *window_sum)
{
    if(window_sum) *window_sum = h->tab.window_sum;
    if(out) *out = h->tab.window.empty() ? nullptr : h->tab.window.data();
    return h->tab.window.size();
}
size_t wf_hip_table_slope(const wf_hip *h, const float **out)
{
    if(out) *out = h->tab.slope.empty() ? nullptr : h->tab.slope.data();
    return h->tab.slope.size();
}
size_t wf_hip_table_rolloff(const wf_hip *h, const float **out)
{
    if(out) *out = h->tab.rolloff.empty() ? nullptr : h->tab.rolloff.data();
    return h->tab.rolloff.size();
}
size_t wf_hip_table_interp_indices(const wf_hip *h, const float **out)
{
    if(out) *out = h->tab.interp_indices.empty() ? nullptr : h->tab.interp_indices.data();
    return h->tab.interp_indices.size();
}
size_t wf_hip_table_band_widths(const wf_hip *h, const int **out)
{
    if(out) *out = h->tab.band_widths.empty() ? nullptr : h->tab.band_widths.data();
    return h->tab.band_widths.size();
}
size_t wf_hip_table_interp_weights(const wf_hip *h, const float **out, int *radius, int *taps)
{
    if(radius) *radius = h->tab.interp_radius;
    if(taps) *taps = h->tab.interp_taps;
    if(out) *out = h->tab.interp_weights.empty() ? nullptr : h->tab.interp_weights.data();
    return h->tab.interp_weights.size();
}
float wf_hip_gravity(const wf_hip *h, float seconds) { return wf::gravity_for(h->cfg, seconds); }
float wf_hip_db_min(void) { return wf::db_min(); }

int wf_hip_time_begin(wf_hip *h)
{
    if(h == nullptr)
        return WF_HIP_ERR_INVALID;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_TRY_RC(join_lanes(h));
    WF_HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    return WF_HIP_OK;
}

int wf_hip_time_end(wf_hip *h, float *elapsed_ms)
{
    if(h == nullptr || elapsed_ms == nullptr)
        return WF_HIP_ERR_INVALID;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_TRY_RC(join_lanes(h));
    WF_HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    WF_HIP_TRY(h, hipEventSynchronize(h->ev1));
    WF_HIP_TRY(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return WF_HIP_OK;
}

int wf_hip_time_ticks(wf_hip *h, const wf_hip_tick_params *p, uint32_t ticks, uint32_t hop, float *avg_kernel_ms)
{
    if(h == nullptr || p == nullptr || ticks == 0 || avg_kernel_ms == nullptr)
        return WF_HIP_ERR_INVALID;
    // the walk starts over at the oldest window when it has reached the newest sample
    const uint32_t period = hop ? p->delay_frames / hop + 1 : ticks;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_TRY_RC(join_lanes(h));
    // events recorded on the handle's own stream, around the fused kernels only (the lanes fork behind ev0 and are joined
    // in front of ev1)
    WF_HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    wf_hip_tick_params q = *p;
    for(uint32_t i = 0; i < ticks; ++i) {
        q.delay_frames = p->delay_frames - (i % period) * hop;
        int rc = wf_hip_tick(h, &q);
        if(rc)
            return rc;
    }
    WF_TRY_RC(join_lanes(h));
    WF_HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    WF_HIP_TRY(h, hipEventSynchronize(h->ev1));
    float ms = 0.0f;
    WF_HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_kernel_ms = ms / (float)ticks;
    return WF_HIP_OK;
}

// Test aid: the streams' sample counters (write position, RMS / meter / waveform consumption points) move on by `frames`, as
// if that much more audio had been captured before what the rings hold now -- `frames` must be a multiple of every ring
// capacity, so that positions keep addressing the same ring cells.  Lets a test reach the 2^32-sample wrap-around (a day
// of audio at 48 kHz) without pushing a day of audio.
extern "C" int wf_hip_debug_age(wf_hip *h, uint32_t first, uint32_t count, uint32_t frames)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(frames % h->ring_cap || (h->d_rms_ring && frames % h->rms_cap))
        return fail(h, WF_HIP_ERR_INVALID, "frames must be a multiple of the ring capacity %u%s", h->ring_cap, h->d_rms_ring ? " and of the RMS ring's" : "");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_TRY_RC(join_lanes(h));
    hipLaunchKernelGGL(wf::age_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->d_wpos, h->d_rend, h->d_mend, h->d_cend, first, count,
                       frames);
    WF_HIP_TRY(h, hipGetLastError());
    h->main_dirty = true;
    return WF_HIP_OK;
}

#ifdef WF_PHASE_TIMING
// development aid: copies the per-workgroup s_memtime stamps of the last tick (16 per workgroup)
extern "C" int wf_hip_debug_phase_clock(wf_hip *h, unsigned long long *out, size_t n)
{
    if(hipMemcpy(out, h->d_phase_clock, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return WF_HIP_ERR_RUNTIME;
    return WF_HIP_OK;
}
#endif

const char *wf_hip_kernel_name(const wf_hip *h) { return h ? h->kernel_name.c_str() : ""; }

uint32_t wf_hip_launches_per_tick(const wf_hip *h)
{
    if(h == nullptr)
        return 0;
    if(h->wave || h->meter)
        return 1;
    return (uint32_t)(h->d_rms_ring ? 1 : h->n_lanes) * (h->split_mono ? 2u : 1u);
}

uint64_t wf_hip_algorithmic_bytes_per_tick(const wf_hip *h, uint32_t flags)
{
    if(h == nullptr)
        return 0;
    if(h->wave) // read + write every row (the shift), plus the samples picked from the rings (not counted: <= width per row)
        return (uint64_t)h->n_streams * h->out_ch * h->N * 8ull;
    if(h->meter) // read the meter buffer of every captured channel; state, level and bar are a few floats per channel
        return (uint64_t)h->n_streams * h->cap_ch * (4ull * h->N + 20ull);
    // read the N-sample window of every captured channel, read + write the smoothing
    // state (M floats each way) when temporal smoothing is on, write M dB values per displayed/output channel
    // (stereo: both channels; mono mixdown: one), plus the bar heights when the configuration has bars.
    const uint64_t n_spec = (uint64_t)h->n_streams * h->cap_ch;
    uint64_t bytes = n_spec * 4ull * h->N;
    if(h->cfg.tsmoothing != WF_TSMOOTH_NONE)
        bytes += n_spec * 8ull * h->M;
    const bool mono_mix = !h->cfg.stereo && h->cap_ch > 1;
    const uint64_t out_rows = (uint64_t)h->n_streams * (mono_mix ? 1u : h->out_ch);
    if(!(flags & WF_HIP_TICK_NO_DECIBELS) || mono_mix || h->big_l || h->ext_outputs) // the mono-mixdown row is stored in either mode;
                                                                                       // so are rows the outputs are derived from
        bytes += out_rows * 4ull * h->M;
    bytes += (uint64_t)h->n_streams * h->disp_ch * h->num_bars * 4ull;
    return bytes;
}

} // extern "C"
