// wf_tick_geom.hip -- kernel dispatch of ONE FFT geometry: compiled once per geometry (-DWF_TU_GEOM=512 ... 32768, see the
// Makefile), so that the ~70 instantiations of spectrum_tick_kernel build in parallel instead of in one translation unit.
// Host side: the launch functions wf_hip_tick calls through wf_hip::launch, and setup_tick_<N>() which wf_hip_create
// (wf_hip_plan.hip) calls to pick one.  gfx950 only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "wf_hip_internal.hpp"
#include "wf_kernels.hpp"

#ifndef WF_TU_GEOM
#error "compile with -DWF_TU_GEOM=<fft size of the geometry>"
#endif

namespace {

using wf::host::fail;

// The power-of-two kernels come in display-specific instantiations (spectrum_tick_kernel<.., MIR, DISP>, wf_kernels.hpp): what a
// launch needs is known on the host -- no display at all (DISP 2), bars in the prefix-sum layout and nothing else (DISP 1), anything
// else (DISP 0); with MIR the stores into wf_hip_set_bars_mirrors' buffers.  One launch = one of them, picked here.
inline int display_kind(const wf::TickArgs &a)
{
#ifdef WF_DEV_BUILD
    static const bool off = getenv("WF_HIP_DISP") && atoi(getenv("WF_HIP_DISP")) == 0; // (A/B aid: the general instantiation for everything)
    if(off)
        return 0;
#endif
    if(a.bar.out == nullptr)
        return 2;
    return (a.bar.ps_lanes > 0 && a.bar.curve == 0) ? 1 : 0;
}
template<class G, int SPW, bool ALIGNED, bool SPLIT, int DEC, bool TLDS, bool BOTH, bool MIR, int DISP>
void launch_pow2_one(dim3 grid, dim3 block, size_t lds, hipStream_t st, const wf::TickArgs &a)
{
    hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, wf::Variant{.spw = SPW, .aligned = ALIGNED, .split = SPLIT, .dec = DEC, .tlds = TLDS, .both = BOTH, .mir = MIR, .disp = DISP}>), grid,
                       block, lds, st, a);
}
// SPECIAL: whether this family has display-specific instantiations at all (the curve-sharing and decimated kernels keep the general one)
template<class G, int SPW, bool SPLIT, int DEC, bool TLDS, bool BOTH, bool SPECIAL>
void launch_pow2(dim3 grid, dim3 block, size_t lds, hipStream_t st, const wf::TickArgs &a, bool aligned)
{
    constexpr bool SPLIT_MIR = SPLIT; // (the split kernels serve the mirrors in their only instantiation: spectrum_tick_kernel's MIRROR)
    const bool mir = !SPLIT_MIR && a.bar.out2_n > 0;
    const int disp = SPECIAL ? display_kind(a) : 0;
#define WF_L(AL, MIR_, DISP_) launch_pow2_one<G, SPW, AL, SPLIT, DEC, TLDS, BOTH, MIR_, DISP_>(grid, block, lds, st, a)
    if constexpr(SPECIAL) {
        if(disp == 2) {
            if(aligned) WF_L(true, false, 2); else WF_L(false, false, 2);
            return;
        }
        if(disp == 1) {
            if constexpr(SPLIT_MIR) {
                if(aligned) WF_L(true, false, 1); else WF_L(false, false, 1);
            } else if(mir) {
                if(aligned) WF_L(true, true, 1); else WF_L(false, true, 1);
            } else {
                if(aligned) WF_L(true, false, 1); else WF_L(false, false, 1);
            }
            return;
        }
    }
    if constexpr(SPLIT_MIR) {
        if(aligned) WF_L(true, false, 0); else WF_L(false, false, 0);
    } else if(mir) {
        if constexpr(DEC > 0)
            WF_L(false, true, 0); // (the decimated sizes: the scalar fetch for both alignments)
        else {
            if(aligned) WF_L(true, true, 0); else WF_L(false, true, 0);
        }
    } else {
        if(aligned) WF_L(true, false, 0); else WF_L(false, false, 0);
    }
#undef WF_L
}
template<class G, int SPW, bool SPLIT, int DEC, bool TLDS, bool BOTH, bool SPECIAL>
int setup_pow2_lds(wf_hip *h, int lds)
{
#define WF_A(AL, MIR_, DISP_)                                                                                                                  \
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(                                                                          \
                                          &wf::spectrum_tick_kernel<G, wf::Variant{.spw = SPW, .aligned = AL, .split = SPLIT, .dec = DEC, .tlds = TLDS, .both = BOTH, .mir = MIR_, .disp = DISP_}>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    WF_A(true, false, 0);
    WF_A(false, false, 0);
    if constexpr(!SPLIT) {
        WF_A(false, true, 0);
        if constexpr(DEC == 0)
            WF_A(true, true, 0);
    }
    if constexpr(SPECIAL) {
        WF_A(true, false, 1);
        WF_A(false, false, 1);
        WF_A(true, false, 2);
        WF_A(false, false, 2);
        if constexpr(!SPLIT) {
            WF_A(true, true, 1);
            WF_A(false, true, 1);
        }
    }
#undef WF_A
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
        launch_pow2<G, 1, true, 0, false, false, true>(grid, block, lds, h->launch_stream, a, aligned);
    }
}

template<class G> int setup_launch_split(wf_hip *h)
{
    const int lds = (int)wf::tick_lds_bytes<G, 1>();
    WF_TRY_RC((setup_pow2_lds<G, 1, true, 0, false, false, true>(h, lds)));
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
    launch_pow2<G, 2, false, DEC, false, false, false>(grid, block, lds, h->launch_stream, a, aligned);
}

template<class G, int DEC> int setup_launch_dec(wf_hip *h)
{
    const int lds = (int)wf::tick_lds_bytes<G, 2>();
    WF_TRY_RC((setup_pow2_lds<G, 2, false, DEC, false, false, false>(h, lds)));
    h->launch = &launch_tick_dec<G, DEC>;
    h->wg_lds = (uint32_t)lds;
    h->wg_threads = (uint32_t)G::T * 2u;
    char name[96];
    snprintf(name, sizeof(name), "spectrum_tick_kernel<N=%d zero-padded to %d,T=%d,R=%dx%dx%d,SPW=2>", G::N >> DEC, G::N, G::T, G::R1, G::R2, G::R3);
    h->kernel_name = name;
    return WF_HIP_OK;
}

// Bluestein path (FFT sizes that are not powers of two): always the scalar fetch
template<class G, int SPW, bool SPLIT, bool MR = false, bool MRS = false> void launch_tick_blu(wf_hip *h, const wf::TickArgs &a0, bool)
{
    const uint32_t n_spec = a0.stream_count * a0.cap_ch;
    const dim3 block(G::T * SPW);
    // (mixed radix: the exchange buffers by the transform's size, MrPlan::lds_cf)
    const size_t lds = wf::tick_lds_bytes<G, SPW>() - (MR ? (size_t)SPW * ((size_t)G::LDS_CF - (size_t)h->mr_lds_cf) * sizeof(wf::cf) : 0);
    const bool two = SPLIT && h->split_mono; // mono mixdown in two launches (TickArgs::split_ch)
    for(int pass = 0; pass < (two ? 2 : 1); ++pass) {
        wf::TickArgs a = a0;
        a.split_ch = two ? (uint32_t)(1 - pass) : 0xffffffffu;
        const dim3 grid(two ? a.stream_count : (n_spec + SPW - 1) / SPW);
        // (no display: the instantiation without any display code -- spectrum_tick_kernel<.., DISP = 2>; the sizes with a
        // compile-time plan: the instantiation of that plan alone -- <.., PLAN>, wf_hip::mr_plan_id)
        const bool nodisp = MRS && display_kind(a) == 2; // (the small-radix instantiation only: N = 4160 on the all-radix one measured -1.8 %)
#define WF_LP(PLAN_)                                                                                                                                            \
    do {                                                                                                                                                       \
        if(nodisp)                                                                                                                                             \
            hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, wf::Variant{.spw = SPW, .split = SPLIT, .blu = true, .mr = MR, .mrs = MRS, .disp = 2, .plan = PLAN_}>), grid, block, lds, h->launch_stream, a); \
        else                                                                                                                                                   \
            hipLaunchKernelGGL((wf::spectrum_tick_kernel<G, wf::Variant{.spw = SPW, .split = SPLIT, .blu = true, .mr = MR, .mrs = MRS, .plan = PLAN_}>), grid, block, lds, h->launch_stream, a); \
    } while(0)
        if constexpr(MRS && (G::N == 2048 || G::N == 4096)) {
            constexpr int P0 = G::N == 2048 ? 1 : 5; // the container's four fixed plans (spectrum_tick_kernel's PLAN)
            switch(h->mr_plan_id - P0) {
            case 0: WF_LP(P0); break;
            case 1: WF_LP(P0 + 1); break;
            case 2: WF_LP(P0 + 2); break;
            case 3: WF_LP(P0 + 3); break;
            default: WF_LP(0); break;
            }
        } else
            WF_LP(0);
#undef WF_LP
    }
}

template<class G, int SPW, bool SPLIT, bool MR = false, bool MRS = false> int setup_launch_blu(wf_hip *h)
{
    if constexpr(!MR && G::N >= 1024) { // (the smallest container a size that is not a power of two ever gets: wf::bluestein_length)
        // sizes with small prime factors take the same instantiation's fetch and epilogue around a direct transform
        bool direct = true;
#ifdef WF_DEV_BUILD
        if(const char *off = std::getenv("WF_HIP_NO_MIXED_RADIX")) // (development: A/B against Bluestein)
            direct = off[0] != '1';
#endif
        const int passes = direct ? wf::plan_mixed_radix(h->N / 2, (uint32_t)G::T, h->mr_radix, (uint64_t)G::M) : 0;
        if(passes > 0) {
            h->mr_passes = passes;
#ifdef WF_DEV_BUILD
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
#endif
            // one-wavefront containers: plans made of small radices take the instantiation that carries only those (five waves per SIMD)
            if constexpr(G::T <= 256 && G::P > 8) {
                bool small = wf::mr_small_radices(h->mr_radix, h->mr_passes);
#ifdef WF_DEV_BUILD
                if(const char *e = std::getenv("WF_HIP_MR_SMALL")) // 0: the instantiation with every radix (A/B)
                    small = small && e[0] != '0';
#endif
                if(small)
                    return setup_launch_blu<G, SPW, SPLIT, true, true>(h);
            }
            return setup_launch_blu<G, SPW, SPLIT, true>(h);
        }
    }
    int lds = (int)wf::tick_lds_bytes<G, SPW>();
    // (the attribute belongs to the kernel, not to this handle: always the container's size -- another handle of another fft size
    // on the same instantiation may need all of it)
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, wf::Variant{.spw = SPW, .split = SPLIT, .blu = true, .mr = MR, .mrs = MRS}>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, wf::Variant{.spw = SPW, .split = SPLIT, .blu = true, .mr = MR, .mrs = MRS, .disp = 2}>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    h->mr_plan_id = 0;
    if constexpr(MRS && (G::N == 2048 || G::N == 4096)) {
        // the sizes the plugin picks by itself run their plan as compile-time constants, in an instantiation of their own
        static const int fixed[8][3] = {{5, 10, 8}, {5, 12, 8}, {10, 6, 6}, {11, 5, 8}, {10, 8, 10}, {10, 8, 12}, {10, 10, 10}, {10, 8, 11}};
        constexpr int P0 = G::N == 2048 ? 1 : 5;
        for(int i = 0; i < 4 && h->mr_passes == 3; ++i)
            if(h->mr_radix[0] == fixed[P0 - 1 + i][0] && h->mr_radix[1] == fixed[P0 - 1 + i][1] && h->mr_radix[2] == fixed[P0 - 1 + i][2])
                h->mr_plan_id = P0 + i;
#ifdef WF_DEV_BUILD
        if(const char *e = std::getenv("WF_HIP_MR_PLAN_KERNEL")) // 0: the instantiation that carries every plan (A/B)
            if(e[0] == '0')
                h->mr_plan_id = 0;
#endif
#define WF_AP(PLAN_)                                                                                                                                          \
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, wf::Variant{.spw = SPW, .split = SPLIT, .blu = true, .mr = MR, .mrs = MRS, .plan = PLAN_}>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                                                       \
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::spectrum_tick_kernel<G, wf::Variant{.spw = SPW, .split = SPLIT, .blu = true, .mr = MR, .mrs = MRS, .disp = 2, .plan = PLAN_}>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds))
        WF_AP(P0);
        WF_AP(P0 + 1);
        WF_AP(P0 + 2);
        WF_AP(P0 + 3);
#undef WF_AP
    }
    if constexpr(MR) {
        // the spectrum's exchange buffer by the transform's size, not the container's (MrPlan::lds_cf): room for more spectra per CU
        h->mr_half = (int)wf::mr_exchange_half(h->N / 2);
        h->mr_lds_cf = (int)wf::mr_exchange_cf(h->N / 2, (uint32_t)G::LDS_CF);
        h->mr_s3 = h->mr_half / 4 + 4;
        // the compile-time plans of the one-wavefront container run IN PLACE (mr_transform_fixed_inplace): without a display -- whose
        // planner has sized its staging by the two-halves buffer already -- the spectrum needs its points and the four-plane result only
        if(G::T == 64 && h->mr_plan_id != 0 && h->num_bars == 0)
            h->mr_lds_cf = std::min(h->mr_lds_cf, std::max(h->mr_half, 4 * h->mr_s3));
        lds -= SPW * ((int)G::LDS_CF - h->mr_lds_cf) * (int)sizeof(wf::cf);
    }
    h->launch = &launch_tick_blu<G, SPW, SPLIT, MR, MRS>;
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
    launch_pow2<G, SPW, false, 0, TLDS, BOTH, !BOTH>(grid, block, lds, h->launch_stream, a, aligned);
}

template<class G, int SPW, bool TLDS, bool BOTH = false> int setup_launch_impl(wf_hip *h)
{
    const int lds = (int)wf::tick_lds_bytes<G, SPW>();
    WF_TRY_RC((setup_pow2_lds<G, SPW, false, 0, TLDS, BOTH, !BOTH>(h, lds)));
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
#ifdef WF_DEV_BUILD
        if(const char *e = std::getenv("WF_HIP_TLDS"))
            tlds = e[0] == '1';
#endif
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

// the kernel of this handle on geometry G: which of the instantiations above its configuration takes
template<class G> int setup_tick_geometry(wf_hip *h, bool want_split)
{
    const wf_config *cfg = &h->cfg;
    h->waves_per_spectrum = G::T / 64;
    if(h->blu) {
        if constexpr(G::N >= 32768) {
            // (the Bluestein and mixed-radix instantiations of this container keep 1024 threads of 16 points: a mixed-radix
            // plan's last pass has one butterfly per thread at most, and 39 sizes have no plan on 512 threads)
            using GB = wf::GBig;
            h->waves_per_spectrum = GB::T / 64;
            if(want_split)
                return setup_launch_blu<GB, 1, true>(h);
            if(cfg->capture_channels == 1)
                return setup_launch_blu<GB, 1, false>(h);
            return fail(h, WF_HIP_ERR_RUNTIME, "fft_size %u: no launch plan", cfg->fft_size);
        } else if constexpr(G::T >= 256)
            return want_split ? setup_launch_blu<G, 1, true>(h) : (cfg->capture_channels > 1) ? setup_launch_blu<G, 2, false>(h) : setup_launch_blu<G, 1, false>(h);
        else
            return setup_launch_blu<G, 2, false>(h);
    } else if constexpr(G::N == 512) {
        switch(h->N) {
        case 256: return setup_launch_dec<G, 1>(h);
        case 128: return setup_launch_dec<G, 2>(h);
        default: return setup_launch<G, 2>(h);
        }
    } else if constexpr(G::N >= 32768) {
        // one spectrum fills a CU's LDS: a stereo pair runs split, a single captured channel alone; mono mixdown of two
        // channels runs split too, as two launches (TickArgs::split_ch)
        if(want_split)
            return setup_launch_split<G>(h);
        if(cfg->capture_channels == 1)
            return setup_launch<G, 1>(h);
        return fail(h, WF_HIP_ERR_RUNTIME, "fft_size %u: no launch plan", cfg->fft_size);
    } else if constexpr(G::T >= 256)
        return want_split ? setup_launch_split<G>(h) : (cfg->capture_channels > 1) ? setup_launch<G, 2>(h) : setup_launch<G, 1>(h);
    else
        return setup_launch<G, 2>(h);
}

} // namespace

namespace wf::host {

#define WF_CAT2(a, b) a##b
#define WF_CAT(a, b) WF_CAT2(a, b)
int WF_CAT(setup_tick_, WF_TU_GEOM)(wf_hip *h, bool want_split)
{
#if defined(WF_GEOM_ONLY) && (WF_GEOM_ONLY != WF_TU_GEOM)
    (void)want_split; // development builds: one geometry only (tools/variant.sh)
    return fail(h, WF_HIP_ERR_UNSUPPORTED, "development build: only the %d-sample geometry is compiled in", WF_GEOM_ONLY);
#else
    return setup_tick_geometry<WF_CAT(wf::G, WF_TU_GEOM)>(h, want_split);
#endif
}

} // namespace wf::host
