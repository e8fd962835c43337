// wf_hip.hip -- the entry points of the C ABI in include/wf_hip.h other than create / destroy (wf_hip_plan.hip): audio
// ingest, the tick, per-stream settings, readbacks, measurement -- host side + the launches of the small kernels (rings,
// level meter, waveform display, RMS, vertex fill).  The fused spectrum kernel is launched through wf_hip::launch
// (wf_tick_geom.hip, wf_big_dispatch.hip).  gfx950 only.  There is no CPU fallback: every entry point either drives the
// device or fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "wf_hip_internal.hpp"
#include "wf_geometry.hpp"
#include "wf_ring.hpp"
#include "wf_meter.hpp"
#include "wf_rms.hpp"
#include "wf_wave.hpp"
#include "wf_vertex.hpp"

namespace {

using namespace wf::host;

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
        a.bar.piece_mode = h->bar_piece_mode ? 1 : 0;
        a.bar.ps_tab = h->d_ps_tab;
        a.bar.ps_lanes = h->bar_ps_lanes;
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
        a.bar.pre_out = h->d_bars_pre;
        a.bar.out2_n = (int)h->mirror_n;
        for(uint32_t j = 0; j < h->mirror_n; ++j)
            a.bar.out2_delta[j] = (long long)(h->bars_mirror[h->mirror_next][j] - h->d_bars);
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
    a.slope_step = h->tab.slope.empty() ? 0.0f : (float)(3.0 * (double)h->cfg.slope / (double)(h->M - 1));
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
        a.mr.half = h->mr_half;
        a.mr.s3 = h->mr_s3;
        a.mr.lds_cf = h->mr_lds_cf;
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
        a.mr.half = (int)wf::GBig::M / 2; // (the rows kernel's two halves of the 132 KB buffer; its Z goes to device memory)
        a.mr.s3 = h->big_mrw ? a.mr.half / 4 + 4 : 0; // (big_mr_whole_kernel leaves a row's Z in the buffer: mr_z_addr's four planes)
        a.mr.lds_cf = 0;
        a.big_c = h->big_rows;
        a.big_r = h->M / h->big_rows;
        a.big_wc = h->d_big_wc;
    }
    if(h->big_br) { // rows by Bluestein inside LDS: the container geometry's tables in place of the batch geometry's
        a.tw1 = h->d_br_tw1;
        a.tw2 = h->d_br_tw2;
        a.blu_b = h->d_br_bhat;
        a.blu_q = h->d_br_q;
        a.big_c = h->big_rows;
        a.big_r = h->M / h->big_rows;
    }
    if(h->big_l) {
        a.big_z = h->d_big_z;
        a.big_tws = h->d_big_tws;
        a.big_tw = h->d_big_tw;
        a.big_nz_out = h->d_big_nz;
        a.big_nz = h->d_big_nz;
        a.big_m = h->blu ? 0u : h->N / 2;
        a.big_l = h->big_br ? h->big_rows * h->br_rs : h->big_l;
        a.big_rs = h->br_rs;
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

} // namespace

int wf::host::upload_words(wf_hip *h, void *d_dst, const void *src, size_t bytes)
{
    const uint32_t k = h->words_next++ & 1u;
    if(h->ev_words[k] == nullptr)
        WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_words[k], hipEventDisableTiming));
    else
        WF_HIP_TRY(h, hipEventSynchronize(h->ev_words[k])); // the copy that used this block two calls ago (long done)
    if(h->h_words_bytes[k] < bytes) {
        if(h->h_words[k])
            (void)hipHostFree(h->h_words[k]);
        h->h_words[k] = nullptr;
        h->h_words_bytes[k] = 0;
        const size_t want = std::max<size_t>(bytes, 4096);
        WF_HIP_TRY(h, hipHostMalloc(&h->h_words[k], want, hipHostMallocDefault));
        h->h_words_bytes[k] = want;
    }
    std::memcpy(h->h_words[k], src, bytes);
    WF_HIP_TRY(h, hipMemcpyAsync(d_dst, h->h_words[k], bytes, hipMemcpyHostToDevice, h->stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_words[k], h->stream));
    return WF_HIP_OK;
}

namespace {

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
    // (its guard entry goes with it: the next block may get the same address with another size)
    h->guards.erase(std::remove_if(h->guards.begin(), h->guards.end(), [p](const auto &g) { return g.first == p; }), h->guards.end());
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
        if(h->d_bars_pre)
            hipLaunchKernelGGL(wf::fill_f32_kernel, dim3(((size_t)count * h->disp_ch + 255) / 256), dim3(256), 0, h->stream,
                               h->d_bars_pre + (size_t)first * h->disp_ch, (size_t)count * h->disp_ch, h->tab.border_bottom);
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
    if(samples != nullptr && h != nullptr && h->d_rms_ring != nullptr && !h->rms_feed)
        return push_host(h, first, count, samples, frames, true); // the RMS producer takes the packet's samples
    // a packet without data, or nobody to read it: CircularBuffer::push_back_zero
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    return push_common(h, first, count, nullptr, nullptr, frames);
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

// roctx ranges around the tick ("Tracing": the reference has none either; the timeline of a profiled process
// then shows the host's part of every tick next to the kernels).  Opt-in -- WF_HIP_ROCTX=1 in the environment -- and resolved at
// run time from the profiler's own marker library (librocprofiler-sdk-roctx.so, else libroctx64.so): the product links neither.
extern "C++" {
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = std::getenv("WF_HIP_ROCTX");
        if(e == nullptr || e[0] == '0')
            return;
        for(const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if(void *lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
                if(push && pop)
                    return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
const Roctx &roctx()
{
    static const Roctx r; // (thread-safe initialisation)
    return r;
}
} // namespace
} // extern "C++"

static int wf_hip_tick_impl(wf_hip *h, const wf_hip_tick_params *p);
int wf_hip_tick(wf_hip *h, const wf_hip_tick_params *p)
{
    const Roctx &rx = roctx();
    if(rx.push == nullptr)
        return wf_hip_tick_impl(h, p);
    rx.push("wf_hip_tick");
    const int rc = wf_hip_tick_impl(h, p);
    rx.pop();
    return rc;
}

static int wf_hip_tick_impl(wf_hip *h, const wf_hip_tick_params *p)
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
                big_outputs_launch(h, a, (hi - lo) * h->disp_ch, h->launch_stream);
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
    if(h->mirror_n)
        h->mirror_fresh = true;
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
    WF_TRY_RC(upload_words(h, h->d_delay + first, delay_frames, (size_t)count * sizeof(uint32_t))); // (staged: `delay_frames` is borrowed for the call only, and the call does not wait)
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
    WF_TRY_RC(upload_words(h, h->d_audio_ts + first, audio_ts_ns, (size_t)count * sizeof(uint64_t))); // (staged, no wait: see upload_words)
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

int wf_hip_enable_input_rms(wf_hip *h, int feed) { return enable_rms_producer(h, feed != 0); }

int wf_hip_push_rms_ragged_async(wf_hip *h, uint32_t first, uint32_t count, const float *pinned_sq, const uint32_t *frames, uint32_t max_frames,
                                 uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(pinned_sq == nullptr || frames == nullptr || slot > 1 || max_frames == 0)
        return fail(h, WF_HIP_ERR_INVALID, "values or frames is NULL, max_frames is 0 or slot is not 0 / 1");
    if(!h->rms_feed)
        return fail(h, WF_HIP_ERR_INVALID, "wf_hip_push_rms_ragged_async needs wf_hip_enable_input_rms(h, 1)");
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

int wf_hip_sync(wf_hip *h)
{
    if(h == nullptr)
        return WF_HIP_ERR_INVALID;
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_TRY_RC(join_lanes(h));
    WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
    if(h->canary) { // (everything that can write has drained: the side streams too)
        if(h->copy_stream) WF_HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
        if(h->read_stream) WF_HIP_TRY(h, hipStreamSynchronize(h->read_stream));
        return check_canaries(h);
    }
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

uint32_t wf_hip_num_vertices(const wf_hip *h) { return (h && h->d_verts) ? (uint32_t)h->vtab.per_row : 0u; }

const float *wf_hip_vertices_device(wf_hip *h)
{
    if(h == nullptr || h->d_verts == nullptr)
        return nullptr;
    (void)join_lanes(h);
    return reinterpret_cast<const float *>(h->d_verts);
}

static int read_bars_snapshot_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_out, uint32_t slot)
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

static int read_rows_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_rows, uint8_t *pinned_last_silent, uint32_t slot)
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

static int read_premirror_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_out, uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(slot > 1 || pinned_out == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer is NULL or slot is not 0 / 1");
    if(h->d_bars_pre == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "the configuration has no mirrored display (cfg.mirror_freq_axis == 0, or no bars / curve)");
    if(!h->rows_in_flight[slot] || h->read_stream == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "wf_hip_read_async: premirror rides behind rows");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_HIP_TRY(h, hipMemcpyAsync(pinned_out, h->d_bars_pre + (size_t)first * h->disp_ch, (size_t)count * h->disp_ch * sizeof(float), hipMemcpyDeviceToHost,
                                 h->read_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_read[slot], h->read_stream));
    return WF_HIP_OK;
}

static int read_display_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_bars, float *pinned_vertices, uint32_t *pinned_counts,
                              uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(slot > 1)
        return fail(h, WF_HIP_ERR_INVALID, "slot is not 0 / 1");
    if(h->d_bars == nullptr || h->meter)
        return fail(h, WF_HIP_ERR_INVALID, "the configuration displays neither bars nor a curve (cfg.bars == 0 and cfg.curve == 0)");
    if((pinned_vertices != nullptr || pinned_counts != nullptr) && h->d_verts == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "configuration has no vertex fill (cfg.vertices == 0)");
    if(!h->rows_in_flight[slot] || h->read_stream == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "wf_hip_read_async: bars / vertices ride behind rows");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    // behind the rows' copy on the readback stream (which already waits for the tick, its bars and its vertex fill on every lane);
    // the next tick waits, on the device, for this slot's event before it overwrites any of them
    const size_t per = (size_t)h->disp_ch * h->num_bars;
    if(pinned_bars)
        WF_HIP_TRY(h, hipMemcpyAsync(pinned_bars, h->d_bars + first * per, count * per * sizeof(float), hipMemcpyDeviceToHost, h->read_stream));
    if(pinned_vertices) {
        const size_t pv = (size_t)h->disp_ch * h->vtab.per_row;
        WF_HIP_TRY(h, hipMemcpyAsync(pinned_vertices, h->d_verts + first * pv, count * pv * sizeof(wf::f4), hipMemcpyDeviceToHost, h->read_stream));
    }
    if(pinned_counts)
        WF_HIP_TRY(h, hipMemcpyAsync(pinned_counts, h->d_vert_counts + (size_t)first * h->disp_ch, (size_t)count * h->disp_ch * sizeof(uint32_t),
                                     hipMemcpyDeviceToHost, h->read_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_read[slot], h->read_stream));
    return WF_HIP_OK;
}

static int read_meter_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_levels, uint8_t *pinned_last_silent, uint32_t slot)
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

// m_input_rms behind the rows' copy on the readback stream (which already waits for the tick)
static int read_input_rms_async(wf_hip *h, uint32_t first, uint32_t count, float *pinned_out, uint32_t slot)
{
    if(h->d_input_rms == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "the device RMS producer is not enabled (wf_hip_enable_input_rms)");
    WF_HIP_TRY(h, hipSetDevice(h->device));
    WF_HIP_TRY(h, hipMemcpyAsync(pinned_out, h->d_input_rms + first, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, h->read_stream));
    WF_HIP_TRY(h, hipEventRecord(h->ev_read[slot], h->read_stream));
    return WF_HIP_OK;
}

int wf_hip_read_async(wf_hip *h, uint32_t first, uint32_t count, const wf_hip_readback *dst, uint32_t slot)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    if(dst == nullptr || slot > 1)
        return fail(h, WF_HIP_ERR_INVALID, "destination set is NULL or slot is not 0 / 1");
    const bool riders = dst->premirror || dst->vertices || dst->vertex_counts || dst->input_rms;
    if(dst->meter) { // meter batches: level + m_last_silent, from snapshots
        if(dst->last_silent == nullptr || dst->rows || dst->bars || riders)
            return fail(h, WF_HIP_ERR_INVALID, "wf_hip_read_async: meter goes with last_silent and nothing else");
        return read_meter_async(h, first, count, dst->meter, dst->last_silent, slot);
    }
    if(dst->rows == nullptr) { // the bars alone: from a snapshot per slot
        if(dst->bars == nullptr || dst->last_silent || riders)
            return fail(h, WF_HIP_ERR_INVALID, "wf_hip_read_async: without rows only the bars can be read (rows + last_silent lead every other combination)");
        return read_bars_snapshot_async(h, first, count, dst->bars, slot);
    }
    if(dst->last_silent == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "wf_hip_read_async: rows go with last_silent");
    rc = read_rows_async(h, first, count, dst->rows, dst->last_silent, slot);
    if(rc == WF_HIP_OK && dst->input_rms)
        rc = read_input_rms_async(h, first, count, dst->input_rms, slot);
    if(rc == WF_HIP_OK && (dst->bars || dst->vertices || dst->vertex_counts))
        rc = read_display_async(h, first, count, dst->bars, dst->vertices, dst->vertex_counts, slot);
    if(rc == WF_HIP_OK && dst->premirror)
        rc = read_premirror_async(h, first, count, dst->premirror, slot);
    return rc;
}

// where an output lives on the device and how large it is per stream; nullptr + a text when the batch has none
static const void *output_source(const wf_hip *h, wf_hip_output what, size_t *per_stream, const char **why)
{
    *per_stream = 0;
    *why = "";
    switch(what) {
    case WF_HIP_OUT_DECIBELS:
        if(h->meter) { *why = "meter batch: there is no m_decibels; read the levels (WF_HIP_OUT_METER)"; return nullptr; }
        *per_stream = (size_t)h->out_ch * h->M * sizeof(float);
        return h->d_decibels;
    case WF_HIP_OUT_BARS:
        if(h->d_bars == nullptr) { *why = "configuration has no bars (cfg.bars == 0 and cfg.curve == 0)"; return nullptr; }
        *per_stream = (size_t)h->disp_ch * h->num_bars * sizeof(float);
        return h->d_bars;
    case WF_HIP_OUT_PREMIRROR:
        if(h->d_bars_pre == nullptr) { *why = "the configuration has no mirrored display (cfg.mirror_freq_axis == 0, or no bars / curve)"; return nullptr; }
        *per_stream = (size_t)h->disp_ch * sizeof(float);
        return h->d_bars_pre;
    case WF_HIP_OUT_VERTICES:
        if(h->d_verts == nullptr) { *why = "configuration has no vertex fill (cfg.vertices == 0)"; return nullptr; }
        *per_stream = (size_t)h->disp_ch * h->vtab.per_row * sizeof(wf::f4);
        return h->d_verts;
    case WF_HIP_OUT_VERTEX_COUNTS:
        if(h->d_vert_counts == nullptr) { *why = "configuration has no vertex fill (cfg.vertices == 0)"; return nullptr; }
        *per_stream = (size_t)h->disp_ch * sizeof(uint32_t);
        return h->d_vert_counts;
    case WF_HIP_OUT_LAST_SILENT:
        *per_stream = sizeof(uint8_t); // (read as flag words, narrowed on the host)
        return h->d_flags;
    case WF_HIP_OUT_TSMOOTH:
        if(h->meter || h->wave) { *why = "meter / waveform batch: there is no m_tsmooth_buf"; return nullptr; }
        *per_stream = (size_t)h->cap_ch * h->M * sizeof(float);
        return h->d_tsmooth;
    case WF_HIP_OUT_METER:
        if(!h->meter) { *why = "not a meter batch (cfg.meter == 0)"; return nullptr; }
        *per_stream = (size_t)h->cap_ch * sizeof(float);
        return h->d_meter_val;
    case WF_HIP_OUT_INPUT_RMS:
        if(h->d_input_rms == nullptr) { *why = "the device RMS producer is not enabled (wf_hip_enable_input_rms)"; return nullptr; }
        *per_stream = sizeof(float);
        return h->d_input_rms;
    case WF_HIP_OUT_WAVEFORM_TS:
        if(!h->wave || h->d_wts == nullptr) { *why = "m_waveform_ts belongs to waveform batches"; return nullptr; }
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t));
        *per_stream = sizeof(uint64_t);
        return h->d_wts;
    }
    *why = "unknown output";
    return nullptr;
}

size_t wf_hip_output_bytes(const wf_hip *h, wf_hip_output what)
{
    if(h == nullptr)
        return 0;
    size_t per = 0;
    const char *why = nullptr;
    return output_source(h, what, &per, &why) ? per : 0;
}

int wf_hip_read(wf_hip *h, wf_hip_output what, uint32_t first, uint32_t count, void *out)
{
    int rc = check_range(h, first, count);
    if(rc)
        return rc;
    size_t per = 0;
    const char *why = nullptr;
    const void *src = output_source(h, what, &per, &why);
    if(src == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "%s", why);
    if(out == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "output pointer is NULL");
    if(what == WF_HIP_OUT_LAST_SILENT) { // the flag words of the buffer the newest tick wrote, narrowed to one byte per stream
        std::vector<uint32_t> tmp(count);
        rc = read_back(h, h->d_flags + (size_t)h->flag_cur * h->n_streams + first, tmp.data(), count * sizeof(uint32_t));
        if(rc)
            return rc;
        for(uint32_t i = 0; i < count; ++i)
            static_cast<uint8_t *>(out)[i] = (tmp[i] & wf::WF_STREAM_LAST_SILENT) ? 1 : 0;
        return WF_HIP_OK;
    }
    return read_back(h, static_cast<const char *>(src) + (size_t)first * per, out, (size_t)count * per);
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

int wf_hip_set_bars_mirrors(wf_hip *h, uint32_t n, void *const *d_out0, void *const *d_out1)
{
    if(h == nullptr)
        return WF_HIP_ERR_INVALID;
    if(n > 8 || (n > 0 && (d_out0 == nullptr || d_out1 == nullptr)))
        return fail(h, WF_HIP_ERR_INVALID, "wf_hip_set_bars_mirrors: at most 8 buffers per set, both sets given");
    if(n > 0) {
        if(h->d_bars == nullptr || h->meter || h->wave)
            return fail(h, WF_HIP_ERR_INVALID, "configuration has no bars (cfg.bars == 0 and cfg.curve == 0, or a level-meter / waveform batch)");
        if(h->ext_outputs || h->big_l != 0 || h->blu)
            return fail(h, WF_HIP_ERR_UNSUPPORTED, "fft_size %u: only the power-of-two sizes up to 32768 whose display the tick kernel finishes itself write further bars buffers; copy the bars with wf_hip_copy_bars_device_async", h->N);
        for(uint32_t j = 0; j < n; ++j)
            if(d_out0[j] == nullptr || d_out1[j] == nullptr || d_out0[j] == d_out1[j] || d_out0[j] == (void *)h->d_bars || d_out1[j] == (void *)h->d_bars)
                return fail(h, WF_HIP_ERR_INVALID, "wf_hip_set_bars_mirrors: buffer %u of a set is NULL, the same in both sets or the handle's own", j);
    }
    // ticks still in flight write the old buffers: they must have run before the sets are replaced (and the caller frees them)
    if(h->mirror_n) {
        WF_HIP_TRY(h, hipSetDevice(h->device));
        WF_HIP_TRY(h, hipStreamSynchronize(h->stream));
        for(int l = 1; l < h->n_lanes; ++l)
            if(h->lane_stream[l])
                WF_HIP_TRY(h, hipStreamSynchronize(h->lane_stream[l]));
    }
    for(uint32_t j = 0; j < 8; ++j) {
        h->bars_mirror[0][j] = j < n ? static_cast<float *>(d_out0[j]) : nullptr;
        h->bars_mirror[1][j] = j < n ? static_cast<float *>(d_out1[j]) : nullptr;
    }
    h->mirror_n = n;
    h->mirror_next = 0;
    h->mirror_fresh = false;
    return WF_HIP_OK;
}

int wf_hip_bars_mirror_ready(wf_hip *h, void *consumer_stream, void **d_out)
{
    if(h == nullptr || d_out == nullptr)
        return WF_HIP_ERR_INVALID;
    if(h->mirror_n == 0)
        return fail(h, WF_HIP_ERR_INVALID, "no mirror buffers set (wf_hip_set_bars_mirrors)");
    if(consumer_stream == nullptr)
        return fail(h, WF_HIP_ERR_INVALID, "consumer stream is NULL");
    float *const *set = h->bars_mirror[h->mirror_next];
    WF_HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t cs = static_cast<hipStream_t>(consumer_stream);
    const size_t per = (size_t)h->disp_ch * h->num_bars;
    // as wf_hip_copy_bars_device_async: the lanes are not joined -- the consumer waits for each of them
    const int lanes = h->lanes_pending ? h->n_lanes : 1;
    for(int l = 0; l < lanes; ++l) {
        hipStream_t st = l == 0 ? h->stream : h->lane_stream[l];
        if(h->ev_bars_lane[l] == nullptr)
            WF_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_bars_lane[l], hipEventDisableTiming));
        if(!h->mirror_fresh) {
            // no tick has written this set: the handle's own bars stand in (each lane copies the slice its ticks write)
            const uint32_t lo = lanes == 1 ? 0u : (uint32_t)((uint64_t)h->n_streams * l / lanes);
            const uint32_t hi = lanes == 1 ? h->n_streams : (uint32_t)((uint64_t)h->n_streams * (l + 1) / lanes);
            for(uint32_t j = 0; j < h->mirror_n && hi > lo; ++j)
                WF_HIP_TRY(h, hipMemcpyAsync(set[j] + (size_t)lo * per, h->d_bars + (size_t)lo * per, (size_t)(hi - lo) * per * sizeof(float), hipMemcpyDefault, st));
        }
        WF_HIP_TRY(h, hipEventRecord(h->ev_bars_lane[l], st));
        WF_HIP_TRY(h, hipStreamWaitEvent(cs, h->ev_bars_lane[l], 0));
    }
    *d_out = set[0];
    h->mirror_next ^= 1u;
    h->mirror_fresh = false;
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

float *wf_hip_decibels_device(wf_hip *h) { return h ? h->d_decibels : nullptr; }
float *wf_hip_bars_device(wf_hip *h) { return h ? h->d_bars : nullptr; }
void *wf_hip_stream(wf_hip *h)
{
    if(h == nullptr)
        return nullptr;
    (void)join_lanes(h); // whatever the caller orders behind this stream is ordered behind every tick enqueued so far
    return (void *)h->stream;
}

size_t wf_hip_table(const wf_hip *h, wf_hip_table_id which, const void **out)
{
    if(out)
        *out = nullptr;
    if(h == nullptr)
        return 0;
    auto give = [&](const auto &v) -> size_t {
        if(out)
            *out = v.empty() ? nullptr : static_cast<const void *>(v.data());
        return v.size();
    };
    switch(which) {
    case WF_HIP_TABLE_WINDOW: return give(h->tab.window);
    case WF_HIP_TABLE_WINDOW_SUM:
        if(out)
            *out = &h->tab.window_sum;
        return 1;
    case WF_HIP_TABLE_SLOPE: return give(h->tab.slope);
    case WF_HIP_TABLE_ROLLOFF: return give(h->tab.rolloff);
    case WF_HIP_TABLE_INTERP_INDICES: return give(h->tab.interp_indices);
    case WF_HIP_TABLE_BAND_WIDTHS: return give(h->tab.band_widths);
    case WF_HIP_TABLE_INTERP_WEIGHTS: return give(h->tab.interp_weights);
    case WF_HIP_TABLE_INTERP_SHAPE:
        if(out)
            *out = h->interp_shape;
        return 2;
    }
    return 0;
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
#ifdef WF_DEV_BUILD
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
#endif

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
