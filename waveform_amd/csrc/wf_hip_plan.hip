// wf_hip_plan.hip -- the plan step of the C ABI (include/wf_hip.h): wf_hip_create builds a handle from a wf_config -- what
// WAVSource::update() does per source (buffers, FFT plan, window / slope / roll-off / interpolation tables: reference
// src/source.cpp:1169-1290, :837-918), here for a batch: device memory, the tables of wf_host_tables.cpp uploaded, the FFT
// geometry and kernel instantiation chosen (wf_tick_geom.hip / wf_big_dispatch.hip), the display plan, the lanes -- and
// wf_hip_destroy gives it all back (free_bufs, src/source.cpp:782-808).  Host code only; gfx950 kernels are launched by the
// other translation units.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "wf_hip_internal.hpp"
#include "wf_geometry.hpp"

namespace wf::host {

thread_local std::string g_create_error;

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

int guard_block(wf_hip *h, void *p, size_t payload_bytes)
{
    hipError_t e = hipMemsetAsync(static_cast<char *>(p) + payload_bytes, GUARD_BYTE, GUARD_BYTES, h->stream);
    if(e != hipSuccess)
        return fail(h, WF_HIP_ERR_RUNTIME, "hipMemsetAsync of a guard failed: %s", hipGetErrorString(e));
    h->guards.emplace_back(p, payload_bytes);
    return WF_HIP_OK;
}

// the guard bytes of every live block, read back and compared (the caller has synchronised the handle's streams)
int check_canaries(wf_hip *h)
{
    if(!h->canary)
        return WF_HIP_OK;
    unsigned char buf[GUARD_BYTES];
    for(size_t i = 0; i < h->guards.size(); ++i) {
        const auto &g = h->guards[i];
        hipError_t e = hipMemcpy(buf, static_cast<char *>(g.first) + g.second, GUARD_BYTES, hipMemcpyDeviceToHost);
        if(e != hipSuccess)
            return fail(h, WF_HIP_ERR_RUNTIME, "reading a guard back failed: %s", hipGetErrorString(e));
        for(size_t b = 0; b < GUARD_BYTES; ++b)
            if(buf[b] != (unsigned char)GUARD_BYTE)
                return fail(h, WF_HIP_ERR_RUNTIME, "WF_HIP_CANARY: device block %zu of %zu (%zu payload bytes) was written %zu bytes past its end",
                            i, h->guards.size(), g.second, b + 1);
    }
    return WF_HIP_OK;
}

} // namespace wf::host

namespace {

using namespace wf::host;

} // namespace

extern "C" {

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
    h->interp_shape[0] = h->tab.interp_radius;
    h->interp_shape[1] = h->tab.interp_taps;
    h->device = device;
    if(const char *e = std::getenv("WF_HIP_CANARY")) // guard bytes behind every device block, checked by wf_hip_sync
        h->canary = e[0] == '1';
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
        bool big_direct = true;
#ifdef WF_DEV_BUILD
        if(const char *no_mr = std::getenv("WF_HIP_NO_MIXED_RADIX")) // (development: A/B against Bluestein)
            big_direct = no_mr[0] != '1';
#endif
        if(h->blu && h->big_l && big_direct) {
            const uint32_t np = h->N / 2;
            // ... and where n/2 = C R with C = 8 or 4 and R <= 4096: the rows by Bluestein over the 8192- / 16384-sample geometry INSIDE
            // LDS (big_br_*_kernel) -- every multiple of 16 up here, the slider's 768 positions among them
            bool rows_ok = true;
#ifdef WF_DEV_BUILD
            if(const char *no_br = std::getenv("WF_HIP_NO_BLUESTEIN_ROWS")) // (development: A/B against Bluestein through device memory)
                rows_ok = no_br[0] != '1';
#endif
            // C = 16 where it divides n/2 (every multiple of 32), else 8: the smaller the container the more workgroups a CU holds --
            // 48064 x 256 streams 0.254 ms with 8 rows over 8192 points, 0.194 with 16 over 4096, 0.209 with 32 over 2048
            // (32 rows over 1024 / 2048 points: 0.209 -- profiles/r05m_bluestein_rows_ab.txt; not compiled in any more)
            uint32_t br_c = 0, br_first = 16u;
#ifdef WF_DEV_BUILD
            if(const char *e = std::getenv("WF_HIP_BR_ROWS")) // 8: every size on 8 rows (A/B)
                br_first = std::atoi(e) == 8 ? 8u : 16u;
#endif
            for(uint32_t c = br_first; c >= 8u && !br_c && rows_ok; c >>= 1)
                if(np % c == 0 && np / c <= 4096u && np / c >= 512u)
                    br_c = c;
            bool mrw_ok = true;
#ifdef WF_DEV_BUILD
            if(const char *e = std::getenv("WF_HIP_MR_WHOLE")) // 0: two rows through rows + epilogue like the others (A/B)
                mrw_ok = e[0] != '0';
#endif
            for(uint32_t c = 2; c <= 8 && !h->big_mr; ++c) {
                if(np % c || np / c > 8192u)
                    continue;
                int radix[4] = {0, 0, 0, 0};
                int passes = 0;
                bool whole = false;
                if(c == 2u && mrw_ok) { // two rows: on 512 threads where a plan exists -- one kernel (big_mr_whole_kernel)
                    passes = wf::plan_mixed_radix(np / c, 512u, radix);
                    whole = passes > 0;
                }
                if(passes <= 0)
                    passes = wf::plan_mixed_radix(np / c, 1024u, radix);
                // A plan that opens with a prime pass (29 ... 127: wf::mr_pass_prime, p products per point) loses to the Bluestein rows:
                // of the slider's 251 such positions 215 are faster there, by up to 40 % (113x8x9: 0.53 -> 0.31 ms at 256 streams), the
                // other 36 slower by 6 % on average (profiles/r05_sizes_large_before.jsonl)
                if(passes > 0 && !(br_c && radix[0] > 25)) {
                    h->big_mr = true;
                    h->big_mrw = whole;
                    h->mr_passes = passes;
                    std::copy(radix, radix + 4, h->mr_radix);
                    h->blu = false;      // no chirp tables, no chirped window: the plain packed real transform
                    h->big_l = np;       // (complex points per spectrum in the scratch buffer)
                    h->big_rows = c;
                }
            }
            if(!h->big_mr && br_c) {
                h->big_br = true;
                h->blu = false; // (as above: the plain packed real transform, its rows by chirp-z)
                h->big_l = np;
                h->big_rows = br_c;
                h->br_l = 2048u; // (build_bluestein_rows' container length for rows of more than 512 points)
                while(h->br_l < 2u * (np / br_c) - 1u)
                    h->br_l <<= 1;
                h->br_rs = (np / br_c + 1u) & ~1u;
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
#ifdef WF_DEV_BUILD
        if(const char *e = std::getenv("WF_HIP_RING_PAD"))
            pad = (uint32_t)std::strtoul(e, nullptr, 10) & ~3u;
#endif
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
#ifdef WF_DEV_BUILD
    if(const char *e = std::getenv("WF_HIP_SPLIT"))
        want_split = (e[0] == '1') && h->geom_n >= 8192;
#endif
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
    if(h->num_bars && cfg->mirror_freq_axis && !cfg->meter && !cfg->waveform) // the value render_bars / render_curve see above the middle before the mirror (BarArgs::pre_out)
        WF_CREATE_TRY(dev_alloc(h, &h->d_bars_pre, (size_t)h->n_streams * h->disp_ch));
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
        // 2^40 up to 4096 samples, one power of two less per doubling beyond (2^36 at 65536): |X|^2 overflows only above an amplitude
        // of 2^64 / (N * in_scale) = 4096 (+72 dBFS) at every size from 4096 up, and still answers down to |X| ~ 2e-30
        {
            int lg = 0;
            while((1u << lg) < h->N)
                ++lg;
            h->in_scale = std::ldexp(1.0f, std::min(40, 52 - lg));
        }
        if(h->big_l && h->blu)
            h->in_scale = 0x1p24f;
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
    // will this size run as a mixed-radix transform inside the Bluestein instantiation?  (setup_launch_blu asks the same question)
    auto mixed_radix_direct = [&]() -> bool {
        if(!h->blu || h->big_l)
            return false;
        int radix[4] = {0, 0, 0, 0};
        bool direct = false;
        wf::dispatch_geometry(h->geom_n, [&](auto g) {
            using G = decltype(g);
            if constexpr(G::N >= 32768)
                direct = wf::plan_mixed_radix(h->N / 2, (uint32_t)wf::GBig::T, radix, (uint64_t)wf::GBig::M) > 0;
            else
                direct = G::N >= 1024 && wf::plan_mixed_radix(h->N / 2, (uint32_t)G::T, radix, (uint64_t)G::M) > 0;
        });
#ifdef WF_DEV_BUILD
        if(const char *off = std::getenv("WF_HIP_NO_MIXED_RADIX"))
            direct = direct && off[0] != '1';
#endif
        return direct;
    };
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
        if(!own_kernel && mixed_radix_direct()) // the exchange buffer is sized by the transform there (MrPlan::lds_cf, setup_launch_blu)
            lds_floats = 2u * (size_t)wf::mr_exchange_cf(h->N / 2, (uint32_t)(lds_floats / 2));
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
                            true; // (the kernels that exist with BOTH: wf_tick_geom.hip, setup_launch)
#ifdef WF_DEV_BUILD
            h->curve_both = h->curve_both && std::getenv("WF_HIP_TLDS") == nullptr;
            if(const char *e = std::getenv("WF_HIP_CURVE_BOTH"))
                h->curve_both = h->curve_both && e[0] != '0';
#endif
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
            // wave-private pieces first (no barrier, DPP scan, last-arriver sum: wf_host_tables.hpp BarPieceTables); not with the
            // filter (its inputs are staged by bar index behind a barrier anyway) nor on the zero-padded sizes
            wf::BarPieceTables pieces;
            bool want_pieces = h->tab.gauss_radius == 0 && h->N >= 512u;
            if(h->blu) // Bluestein proper keeps bar_segments' layouts (its instantiations are compiled without this one); the sizes
                       // that will run as a mixed-radix transform take it
                want_pieces = want_pieces && mixed_radix_direct();
#ifdef WF_DEV_BUILD
            if(const char *e = std::getenv("WF_HIP_BAR_PIECES"))
                want_pieces = want_pieces && e[0] != '0';
#endif
            // prefix-sum layout first (BarPsTables: float64 prefix sums of the row in registers, one lane per sub-band -- no
            // per-thread coefficient table at all); power-of-two sizes from 512 samples, no Gaussian filter
            wf::BarPsTables ps;
            bool want_ps = h->tab.gauss_radius == 0 && h->N >= 512u && !h->blu;
#ifdef WF_DEV_BUILD
            if(const char *e = std::getenv("WF_HIP_BAR_PS"))
                want_ps = want_ps && e[0] != '0';
#endif
            if(want_ps && wf::bar_ps(h->tab, threads, ps) && wf::ps_lds_floats(h->M) <= lds_floats) {
                h->bar_ps_lanes = ps.num_lanes;
                h->out_steps = 1;
                WF_PLAN_TRY(upload(h, &h->d_ps_tab, ps.tab));
                WF_PLAN_HIP(hipStreamSynchronize(h->stream)); // the staging vector dies here
                want_pieces = false;
            }
            if(want_pieces && wf::bar_pieces(h->tab, threads, points, points / 4 + 2, pieces) &&
               (size_t)h->M + (size_t)pieces.num_slots <= lds_floats) {
                h->bar_piece_mode = true;
                h->bar_segs = pieces.num_segs;
                h->bar_blocks = pieces.blocks;
                h->out_steps = 1;
                WF_PLAN_TRY(upload(h, &h->d_lane_coef, pieces.coef));
                WF_PLAN_TRY(upload(h, &h->d_lane_base, pieces.base));
                WF_PLAN_TRY(upload(h, &h->d_seg_group, pieces.info));
                WF_PLAN_TRY(upload(h, &h->d_bar_seg, pieces.bar_piece));
                WF_PLAN_HIP(hipStreamSynchronize(h->stream)); // the staging vectors die here
            }
            bool local = h->tab.gauss_radius == 0;
#ifdef WF_DEV_BUILD
            if(const char *e = std::getenv("WF_HIP_BARS_WAVE_LOCAL"))
                local = local && e[0] != '0';
#endif
            if(!h->bar_piece_mode && h->bar_ps_lanes == 0 && wf::bar_segments(h->tab, threads, points / 4 + 2, lanes, local)) {
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
                int cap = 2048;
#ifdef WF_DEV_BUILD
                if(const char *e = std::getenv("WF_HIP_BIG_TASK")) // (development: the task size, a multiple of 64)
                    cap = std::max(64, std::atoi(e) & ~63);
#endif
                for(size_t bq = 0; bq + 1 < h->tab.bar_off.size(); ++bq) {
                    bar_task[bq] = (int)(task.size() / 4);
                    const int e0 = h->tab.bar_off[bq], e1 = h->tab.bar_off[bq + 1];
                    const int parts = std::max(1, (e1 - e0 + cap - 1) / cap);
                    const int per = (((e1 - e0 + parts - 1) / parts) + 63) & ~63;
                    for(int q = 0; q < parts; ++q) {
                        const int lo = std::min(e0 + q * per, e1), hi = std::min(lo + per, e1);
                        if(q == 0 || lo < hi) {
                            // a task whose entries walk consecutive bins (every bar of an interpolated display does: a band and its
                            // taps) says where it starts: the kernel then forms the bins' addresses instead of loading them first
                            bool run = lo < hi;
                            for(int e = lo + 1; e < hi && run; ++e)
                                run = h->tab.bar_bin[(size_t)e] == h->tab.bar_bin[(size_t)lo] + (e - lo);
                            task.push_back((int)bq);
                            task.push_back(lo);
                            task.push_back(hi);
                            task.push_back(run ? h->tab.bar_bin[(size_t)lo] : -1);
                        }
                    }
                }
                bar_task.back() = (int)(task.size() / 4);
                h->big_num_tasks = (int)(task.size() / 4);
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
        if(h->bar_segs == 0 && h->bar_ps_lanes == 0 && !h->curve && !own_kernel) { // chunked form: a chunk holds at least one whole bar
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
#ifdef WF_DEV_BUILD
        if(const char *e = std::getenv("WF_HIP_EXT_OUTPUTS")) // 1: the display from the stored rows by big_outputs_kernel even where the tick kernel could finish it (A/B)
            if(e[0] == '1' && orc == WF_HIP_OK)
                orc = WF_HIP_ERR_UNSUPPORTED;
#endif
        if(orc == WF_HIP_ERR_UNSUPPORTED && h->big_l == 0) {
            // give back what the first plan uploaded, forget what it decided, plan again for big_outputs_kernel
            WF_CREATE_HIP(hipStreamSynchronize(h->stream));
            while(h->allocs.size() > mark) {
                void *gone = h->allocs.back();
                h->guards.erase(std::remove_if(h->guards.begin(), h->guards.end(), [gone](const auto &g) { return g.first == gone; }), h->guards.end());
                (void)hipFree(gone);
                h->allocs.pop_back();
            }
            h->d_bar_coef = nullptr; h->d_bar_bin = nullptr; h->d_bar_off = nullptr; h->d_band_widths = nullptr; h->d_bar_chunk = nullptr;
            h->d_cur_coef = nullptr; h->d_cur_base = nullptr; h->d_cur_x = nullptr; h->d_gauss = nullptr; h->d_gauss_wsum = nullptr;
            h->d_lane_coef = nullptr; h->d_lane_base = nullptr; h->d_bar_seg = nullptr; h->d_seg_group = nullptr;
            h->d_lead_bar = nullptr; h->d_lead_end = nullptr;
            h->d_ps_tab = nullptr; h->bar_ps_lanes = 0;
            h->curve = h->curve_both = h->curve_catrom = h->stream_steps = h->bar_wave_local = h->bar_piece_mode = false;
            h->out_steps = h->bar_segs = h->bar_blocks = h->bar_chunks = h->bar_stage_off = 0;
            h->bar_lpb = 1;
            chunks.clear();
            h->ext_outputs = true;
            orc = plan_outputs(true);
        }
        if(orc)
            return bail(orc);
        if(h->ext_outputs && h->big_out_lds)
            WF_CREATE_TRY(wf::host::big_outputs_set_lds(h));
    }

    // FFT plan: twiddle tables for the geometry of this fft_size + the kernel instantiation
    int setup_rc = WF_HIP_ERR_UNSUPPORTED;
    std::vector<wf::cfloat> tw1, tw2, tws;
    wf::dispatch_geometry(h->geom_n, [&](auto g) {
        using G = decltype(g);
        wf::build_twiddles(G::M, G::R1, G::R2, G::R3, tw1, tw2, tws);
        h->waves_per_spectrum = G::T / 64;
        // transforms beyond a CU's LDS (wf_big_dispatch.hip), else the fused kernel of this geometry (wf_tick_geom.hip: one object per geometry)
        if(h->big_l) {
            if constexpr(G::N == 32768)
                setup_rc = wf::host::setup_launch_big(h);
        } else if constexpr(G::N == 512)
            setup_rc = wf::host::setup_tick_512(h, want_split);
        else if constexpr(G::N == 1024)
            setup_rc = wf::host::setup_tick_1024(h, want_split);
        else if constexpr(G::N == 2048)
            setup_rc = wf::host::setup_tick_2048(h, want_split);
        else if constexpr(G::N == 4096)
            setup_rc = wf::host::setup_tick_4096(h, want_split);
        else if constexpr(G::N == 8192)
            setup_rc = wf::host::setup_tick_8192(h, want_split);
        else if constexpr(G::N == 16384)
            setup_rc = wf::host::setup_tick_16384(h, want_split);
        else
            setup_rc = wf::host::setup_tick_32768(h, want_split);
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
    std::vector<wf::cfloat> br_rowtw;
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
    if(h->big_br) {
        // the container transform's twiddles, FFT(chirp), the closing chirp (the column step is a radix-C butterfly in registers); the table of column
        // twiddle x opening chirp goes where the other paths keep their column twiddles (d_big_tw, below)
        std::vector<wf::cfloat> bhat, q, t1f, t2f, unused;
        if(wf::build_bluestein_rows(h->big_l, h->big_rows, br_rowtw, bhat, q) != h->br_l)
            return bail(fail(h, WF_HIP_ERR_RUNTIME, "Bluestein rows: container length"));
        wf::dispatch_geometry(2u * h->br_l, [&](auto g) {
            using G = decltype(g);
            wf::build_twiddles(G::M, G::R1, G::R2, G::R3, t1f, t2f, unused);
        });
        auto as_cf = [](const std::vector<wf::cfloat> &v) {
            std::vector<wf::cf> o(v.size());
            std::memcpy(o.data(), v.data(), v.size() * sizeof(wf::cf));
            return o;
        };
        const std::vector<wf::cf> s1 = as_cf(t1f), s2 = as_cf(t2f), s3 = as_cf(bhat), s4 = as_cf(q); // (alive until the copies are through)
        WF_CREATE_TRY(upload(h, &h->d_br_tw1, s1));
        WF_CREATE_TRY(upload(h, &h->d_br_tw2, s2));
        WF_CREATE_TRY(upload(h, &h->d_br_bhat, s3));
        WF_CREATE_TRY(upload(h, &h->d_br_q, s4));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
    }
    if(h->big_l) {
        std::vector<wf::cfloat> twb, twsb;
        wf::build_big_twiddles(h->big_l, h->big_rows, h->blu ? 0u : h->N, twb, twsb);
        if(h->big_br)
            twb = br_rowtw;
        std::vector<wf::cf> t1(twb.size()), t2(twsb.size());
        std::memcpy(t1.data(), twb.data(), t1.size() * sizeof(wf::cf));
        if(!t2.empty())
            std::memcpy(t2.data(), twsb.data(), t2.size() * sizeof(wf::cf));
        WF_CREATE_TRY(upload(h, &h->d_big_tw, t1));
        WF_CREATE_TRY(upload(h, &h->d_big_tws, t2));
        WF_CREATE_HIP(hipStreamSynchronize(h->stream));
        if(h->big_whole || h->big_mrw) { // (fft_size 65536 and the two-row mixed-radix sizes: no scratch at all, the magnitudes stay in registers)
        } else if(h->big_br) { // (columns -> rows in place -> epilogue)
            WF_CREATE_TRY(dev_alloc(h, &h->d_big_z, n_spec * h->big_rows * h->br_rs));
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
        if(wgs >= 3u * round && wgs < 6u * round && !h->blu && h->M >= 2048 && !(h->split && h->num_bars))
            lanes = 3; // round 6, the display-specific kernels (shorter workgroups): three to five rounds of workgroups as three slices --
                       // headline 0.813 -> 0.824, N = 4096 with bars 0.766 -> 0.777, N = 16384 x 1024 streams 0.710 -> 0.719 (with bars +-0: the split
                       // kernels with a display keep two); eight rounds (8192 streams) -0.4 %
                       // without a display, +1 % with bars: two there (profiles/r06w_lanes_slim_kernels.txt)
        if(wgs >= 6u * round && !h->blu && h->M >= 2048 && !h->split && h->num_bars)
            lanes = 3; // longer batches with a bars display: +0.4 ... +1.4 % in four sweeps (8192 streams, bars-only ticks 0.677 -> 0.6815; r06w, r06y)
        if(h->M <= 512 && !h->cfg.meter && !h->cfg.waveform && wgs >= 6u * round)
            lanes = 3; // the one-wavefront 8-point geometry in long launches: 0.714-0.717 against 0.682-0.683 of the HBM peak at
                       // 16384 streams (steady state, r02j; N = 512 x 16384 streams 0.614 -> 0.636, r06y); from three rounds on instead:
                       // N = 512 x 8192 streams 0.519 -> 0.508 (profiles/r06y_lanes_rule_ab.txt)
        if(per_cu == 1 && wgs >= 2u * round)
            lanes = 3; // one workgroup per CU (32768 samples): fetch, transform and the end of the tick take turns inside a CU, and the
                       // launches of three slices drift apart: 256 streams 0.465 -> 0.513 (two) -> 0.533 (three), 2048 streams 0.472 -> 0.470 -> 0.495
        if(h->big_l) // the transforms through device memory: launch chains of small kernels, nothing to overlap -- except fft_size 65536 in
                     // its one kernel, a CU per workgroup again: 256 streams 0.445 -> 0.557 (two) / 0.49 (three), 64 streams (half a round) 0.259 -> 0.253
            // (the rows of the other sizes up here -- mixed radix, Bluestein in LDS -- likewise from two spectra per CU on: their column /
            // rows / epilogue kernels are bound by different things and two slices' chains overlap: 48016 x 256 streams 0.266 -> 0.250 ms,
            // 48000 x 256 0.170 -> 0.160, 17488 x 512 0.164 -> 0.158, 48016 x 1024 1.07 -> 1.02; three lanes +-2 % around two)
            lanes = ((h->big_whole || h->big_mr || h->big_br) && n_spec >= 2u * (uint32_t)std::max(prop.multiProcessorCount, 1)) ? 2 : 1;
#ifdef WF_DEV_BUILD
        if(const char *e = std::getenv("WF_HIP_LANES"))
            lanes = std::atoi(e);
#endif
        lanes = std::max(1, std::min({lanes, (int)wf_hip::MAX_LANES, (int)h->n_streams}));
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
    for(int i = 0; i < 2; ++i) {
        if(h->ev_words[i]) (void)hipEventDestroy(h->ev_words[i]);
        if(h->h_words[i]) (void)hipHostFree(h->h_words[i]);
    }
    if(h->ev0) (void)hipEventDestroy(h->ev0);
    if(h->ev1) (void)hipEventDestroy(h->ev1);
    if(h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

} // extern "C"
