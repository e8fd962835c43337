// wf_big_dispatch.hip -- kernel dispatch of the FFT sizes whose transform does not fit a CU's LDS (wf_big.hpp: 65536 samples and
// the other sizes above 16384: columns -> rows (twice for Bluestein) -> epilogue -> outputs through device memory), and of
// big_outputs_kernel for the displays that are finished behind the tick kernel from its stored rows.  gfx950 only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "wf_hip_internal.hpp"
#include "wf_big.hpp"

namespace {

using wf::host::fail;

// fft_size 65536, one kernel: both rows of a spectrum and the end of its tick in one workgroup (wf_big.hpp: big_whole_kernel)
int launch_tick_big_whole(wf_hip *h, const wf::TickArgs &a0, bool aligned)
{
    hipStream_t st = h->launch_stream;
    const size_t lds = wf::big_rows_lds_bytes<2>();
    for(int pass = 0; pass < (h->split_mono ? 2 : 1); ++pass) { // mono mixdown: channel 1 of every stream, then channel 0
        wf::TickArgs a = a0;
        a.split_ch = h->split_mono ? (uint32_t)(1 - pass) : 0xffffffffu;
        const dim3 grid(h->split_mono ? a.stream_count : a.stream_count * a.cap_ch);
        if(aligned)
            hipLaunchKernelGGL(wf::big_whole_kernel<true>, grid, dim3(wf::GFold::T), lds, st, a);
        else
            hipLaunchKernelGGL(wf::big_whole_kernel<false>, grid, dim3(wf::GFold::T), lds, st, a);
    }
    if(a0.bar.out != nullptr)
        hipLaunchKernelGGL(wf::big_outputs_kernel, dim3(a0.stream_count * a0.bar.disp_ch), dim3(wf::GBig::T), h->big_out_lds, st, a0);
    WF_HIP_TRY(h, hipGetLastError());
    return WF_HIP_OK;
}

#ifdef WF_DEV_BUILD
// The chain through device memory: columns -> rows (twice, with a pointwise product in between, for Bluestein's direct form) ->
// epilogue.  Rounds 2-4 ran the sizes above 16384 on it; since round 5 every legal size has a faster path (big_whole_kernel, mixed-radix
// rows, Bluestein rows inside LDS), and the chain is compiled into the development builds only, as the A/B baseline
// (WF_HIP_BIG_WHOLE=0, WF_HIP_NO_MIXED_RADIX=1 + WF_HIP_NO_BLUESTEIN_ROWS=1).
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

#endif // WF_DEV_BUILD

// fft sizes above 16384 whose n/2 is two rows of a mixed-radix transform: both rows and the end of the tick in one workgroup
// (wf_big.hpp: big_mr_whole_kernel)
int launch_tick_big_mrw(wf_hip *h, const wf::TickArgs &a0)
{
    hipStream_t st = h->launch_stream;
    const size_t lds = (size_t)(2 * a0.mr.half + 128) * sizeof(wf::cf);
    for(int pass = 0; pass < (h->split_mono ? 2 : 1); ++pass) { // mono mixdown: channel 1 of every stream, then channel 0
        wf::TickArgs a = a0;
        a.split_ch = h->split_mono ? (uint32_t)(1 - pass) : 0xffffffffu;
        const dim3 grid(h->split_mono ? a.stream_count : a.stream_count * a.cap_ch);
        hipLaunchKernelGGL(wf::big_mr_whole_kernel, grid, dim3(wf::GFold::T), lds, st, a);
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

// fft sizes above 16384 with a large prime factor: the column step, the rows by Bluestein inside LDS (in place), the epilogue on the
// rows where they lie (wf_big.hpp)
template<class G, int C> int launch_tick_big_br(wf_hip *h, const wf::TickArgs &a0)
{
    const uint32_t n_spec = a0.stream_count * a0.cap_ch;
    hipStream_t st = h->launch_stream;
    const uint32_t spec_base = a0.stream_base * a0.cap_ch;
    WF_HIP_TRY(h, hipMemsetAsync(h->d_big_nz + spec_base, 0, (size_t)n_spec * sizeof(uint32_t), st));
    hipLaunchKernelGGL((wf::big_br_columns_kernel<C>), dim3((a0.big_r + 255u) / 256u, n_spec), dim3(256), 0, st, a0);
    const dim3 grow(h->big_rows * ((n_spec + 7u) & ~7u));
    hipLaunchKernelGGL((wf::big_br_rows_kernel<G>), grow, dim3(G::T), wf::big_br_lds_bytes<G>(), st, a0);
    const uint32_t parts = (h->M + (uint32_t)wf::BIG_TP - 1u) / (uint32_t)wf::BIG_TP;
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

template<int C> int launch_tick_big_br_g(wf_hip *h, const wf::TickArgs &a)
{
    switch(h->br_l) { // (16 rows: 2048 points up to n = 32768, 4096 above; 8 rows: 4096 / 8192)
    case 2048u: return launch_tick_big_br<wf::G4096, C>(h, a);
    case 4096u: return launch_tick_big_br<wf::G8192, C>(h, a);
    default: return launch_tick_big_br<wf::G16384, C>(h, a);
    }
}
int launch_tick_big_br_c(wf_hip *h, const wf::TickArgs &a)
{
    return h->big_rows == 16u ? launch_tick_big_br_g<16>(h, a) : launch_tick_big_br_g<8>(h, a);
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
            h->launch_rc = h->big_mrw ? launch_tick_big_mrw(h, s) : launch_tick_big_mr(h, s);
            continue;
        }
        if(h->big_br) {
            h->launch_rc = launch_tick_big_br_c(h, s);
            continue;
        }
        if(h->big_whole) {
            h->launch_rc = launch_tick_big_whole(h, s, aligned);
            continue;
        }
#ifdef WF_DEV_BUILD
        switch(h->big_rows) {
        case 2: h->launch_rc = launch_tick_big_l<2>(h, s); break;
        case 4: h->launch_rc = launch_tick_big_l<4>(h, s); break;
        default: h->launch_rc = launch_tick_big_l<8>(h, s); break;
        }
#else
        h->launch_rc = fail(h, WF_HIP_ERR_UNSUPPORTED, "fft_size %u: no kernel for this size in this build", h->N); // (unreachable: setup_launch_big refuses)
#endif
    }
}

#ifdef WF_DEV_BUILD
template<int L1> int setup_big_rows(wf_hip *h)
{
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_rows_kernel<L1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)wf::big_rows_lds_bytes<L1>()));
    return WF_HIP_OK;
}
#endif

} // namespace

namespace wf::host {

int setup_launch_big(wf_hip *h)
{
    int rc = WF_HIP_OK;
    if(h->big_mr) {
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_mr_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)((size_t)(wf::GBig::LDS_CF + 128) * sizeof(wf::cf))));
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_mr_whole_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)((size_t)(wf::GBig::M + 128) * sizeof(wf::cf))));
    }
    else if(h->big_br) {
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_br_rows_kernel<wf::G4096>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)wf::big_br_lds_bytes<wf::G4096>()));
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_br_rows_kernel<wf::G8192>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)wf::big_br_lds_bytes<wf::G8192>()));
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_br_rows_kernel<wf::G16384>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)wf::big_br_lds_bytes<wf::G16384>()));
    } else {
#ifdef WF_DEV_BUILD
        rc = h->big_rows == 2 ? setup_big_rows<2>(h) : h->big_rows == 4 ? setup_big_rows<4>(h) : setup_big_rows<8>(h);
#else
        if(h->blu || h->big_rows != 2) // (no legal size gets here: every multiple of 16 above 16384 has rows of one kind or the other)
            return fail(h, WF_HIP_ERR_UNSUPPORTED, "fft_size %u: neither a power of two nor a length with a row decomposition (a multiple of 16 has one)", h->N);
#endif
    }
    if(rc)
        return rc;
    // fft_size 65536 (the one power of two up here): both rows and the end of the tick in one kernel, no scratch in device memory.
    // WF_HIP_BIG_WHOLE=0 (development builds) sends it through the columns -> rows -> epilogue chain every other size up here takes
    h->big_whole = !h->blu && !h->big_mr && !h->big_br && h->big_rows == 2;
#ifdef WF_DEV_BUILD
    if(const char *e = std::getenv("WF_HIP_BIG_WHOLE"))
        h->big_whole = h->big_whole && e[0] != '0';
#endif
    if(h->big_whole) {
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_whole_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)wf::big_rows_lds_bytes<2>()));
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_whole_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)wf::big_rows_lds_bytes<2>()));
    }
    if(h->big_out_lds)
        WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_outputs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)h->big_out_lds));
    h->launch = &launch_tick_big;
    h->split = true;
    h->flag_bufs = 3;
    char name[256];
    if(h->big_mr) {
        char rad[48];
        const char *form = h->big_mrw ? "big_mr_whole_kernel<N=%u: both of its %u rows of %u complex points as mixed radix %s and the end of the tick in one workgroup>"
                                      : "big_mr_rows_kernel + big_epilogue_kernel<N=%u: %u rows of %u complex points as mixed radix %s, column step folded into the fetch>";
        int o = 0;
        for(int i = 0; i < h->mr_passes; ++i)
            o += snprintf(rad + o, sizeof(rad) - (size_t)o, "%s%d", i ? "x" : "", h->mr_radix[i]);
        snprintf(name, sizeof(name), form, h->N, h->big_rows, h->M / h->big_rows, rad);
    } else if(h->big_br)
        snprintf(name, sizeof(name), "big_br_{columns,rows}_kernel + big_epilogue_kernel<N=%u: %u rows of %u complex points by Bluestein over %u points inside LDS>",
                 h->N, h->big_rows, h->M / h->big_rows, h->br_l);
    else if(h->big_whole)
        snprintf(name, sizeof(name), "big_whole_kernel<N=%u: both rows of 16384 complex points and the end of the tick in one workgroup>", h->N);
    else if(h->blu)
        snprintf(name, sizeof(name), "big_{columns,rows,epilogue}_kernel<N=%u by Bluestein over %u = %u x 16384 complex points through device memory>",
                 h->N, h->big_l, h->big_rows);
    else
        snprintf(name, sizeof(name), "big_{columns,rows,epilogue}_kernel<N=%u: %u = %u x 16384 complex points through device memory>", h->N,
                 h->big_l, h->big_rows);
    h->kernel_name = name;
    return WF_HIP_OK;
}

int big_outputs_set_lds(wf_hip *h)
{
    WF_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wf::big_outputs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)h->big_out_lds));
    return WF_HIP_OK;
}

// the display of `rows` displayed rows (a.stream_base ...) from the rows the tick kernel has just stored, one workgroup each
void big_outputs_launch(wf_hip *h, const wf::TickArgs &a, uint32_t rows, hipStream_t st)
{
    hipLaunchKernelGGL(wf::big_outputs_kernel, dim3(rows), dim3(wf::GBig::T), h->big_out_lds, st, a);
}

} // namespace wf::host
