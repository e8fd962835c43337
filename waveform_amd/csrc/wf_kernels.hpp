// wf_kernels.hpp -- gfx950 kernels of libwaveform_hip.so (device code only; hipcc).
//
//   spectrum_tick_kernel<G, Variant>        the fused per-tick pass: ring fetch -> window -> r2c FFT
//                                           in LDS -> |X| -> slope -> temporal smoothing -> dBFS
//                                           (-> volume normalisation -> roll-off -> bars / curve), one HBM
//                                           pass.  Replaces WAVSource*::tick_spectrum (reference
//                                           src/source_generic.cpp:26-180) for a whole batch of sources.
//   (ring maintenance and state initialisation: wf_ring.hpp)
//
// Work decomposition: a spectrum (one channel of one stream) is owned by T = G::T threads
// (1..8 wavefronts, wf_geometry.hpp); a workgroup holds SPW spectra, laid out so that the
// channels of a stream share a workgroup (silence state machine, mono mixdown, reference :63-95, :150-154).
#pragma once
#include <hip/hip_runtime.h>
#include "wf_geometry.hpp"
#include "wf_tick_phases.hpp"
#include "wf_mixed.hpp"
#include "wf_hip.h"
#ifndef WF_EXP_NO_TAIL
#define WF_EXP_NO_TAIL 0 // 1 (development builds, measurement only: the bars come out wrong): the display phase skipped -- what a tick would cost if the tail were free
#endif
#include "wf_dev_guard.hpp"

namespace wf {

template<class G> __device__ __forceinline__ void spectrum_sync()
{
    // T == 64: the spectrum lives in one wavefront; LDS operations of a wave execute in
    // program order, so a scheduling fence is all that is needed between exchange phases.
    if constexpr(G::T > 64)
        __syncthreads();
    else
        __builtin_amdgcn_wave_barrier();
}

// global -> LDS copy of a BYTES-sized table by the LDS-DMA path (global_load_lds: destination = wave-uniform base +
// lane * width, no VGPR staging); the waves of the workgroup share the chunks.  Tracked by vmcnt: complete after the
// next wait for vector memory + barrier.
template<int BYTES> __device__ __forceinline__ void lds_dma_copy(const void *src, void *lds_dst, int wave, int n_waves, int lane)
{
    constexpr int W = (BYTES % 1024 == 0) ? 16 : 4; // bytes per lane
    constexpr int PER = 64 * W;                      // bytes per wave-wide request
    static_assert(BYTES % PER == 0, "table size must be a multiple of the request size");
    constexpr int CALLS = BYTES / PER;
#pragma unroll
    for(int c = 0; c < CALLS; ++c) {
        if((c % n_waves) == wave) { // wave-uniform
            const char *g = static_cast<const char *>(src) + c * PER + lane * W;
            char *l = static_cast<char *>(lds_dst) + c * PER;
            if constexpr(W == 16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)l, 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)l, 4, 0, 0);
        }
    }
}

// Dynamic LDS: SPW exchange buffers of G::LDS_CF complex each, the pass-2 twiddle table [R2][R3], then one int of
// per-wavefront facts per wave.
template<class G, int SPW> constexpr size_t tick_lds_bytes()
{
    return (size_t)SPW * G::LDS_CF * sizeof(cf) + (size_t)G::R2 * G::R3 * sizeof(cf) + 2 * (size_t)SPW * (G::T / 64) * sizeof(int) + 16;
}

// register budget: waves per SIMD the kernel is compiled for (launch_bounds' second argument) -- 3 only for the
// one-wavefront 16-point geometry (N = 2048), whose register prefetch of the smoothing state needs ~140 VGPRs
#ifndef WF_TRACK
#define WF_TRACK 1
#endif
#ifndef WF_DEFER_STATE
#define WF_DEFER_STATE 0 // 1: smoothing-state stores issued behind the display's table requests (p4_split_smooth<.., DEFER>)
#endif
#ifndef WF_WPS_SMALL
#define WF_WPS_SMALL 4 // 8-point geometry (N = 1024): 5 waves per SIMD was +4 % in round 1 (84 VGPRs); with what the kernel has
                       // learned since (paused streams, underflow, bars-only tracking, lanes) it spilled 36 B per lane at the
                       // 96-register cap and ran 10 % slower than at 4 (tests/test_cpu_units.py now checks for scratch)
#endif
#ifndef WF_WPS_2048_BLU
#define WF_WPS_2048_BLU 4 // (+3-4 % over three waves and no scratch, measured: 52 B of scratch per lane weigh less than a fourth wave) the Bluestein instantiation of the same geometry (fft sizes 528 ... 1008, the automatic size 800 among them)
#endif
#ifndef WF_MR_FIXED_PLANS
#define WF_MR_FIXED_PLANS 1 // 0: every mixed-radix size through the run-time plan (A/B)
#endif
#ifndef WF_WPS_2048_MRS
#define WF_WPS_2048_MRS 5 // 96 registers, 12 B of scratch per lane
#endif
#ifndef WF_WPS_2048
#define WF_WPS_2048 3
#endif
#ifndef WF_WPS_512
#define WF_WPS_512 6 // four points per thread: 80 VGPRs; 0.60 of the HBM peak at N = 512 against 0.53 at 4 and 5 waves, 0.44 at 8 (spills)
#endif
#ifndef WF_WPS_LARGE
#define WF_WPS_LARGE 4 // sixteen points per thread on two wavefronts and more: 104-115 VGPRs.  Five waves per SIMD (96 VGPRs: the
                       // register file hands out blocks of 8) were compiled for the record at the end of round 3
                       // (profiles/r03_occupancy5_isa.txt): the N = 4096 kernel then keeps 72 B per lane in scratch, the 16384 one 48 B --
                       // and a fifth workgroup per CU would also need its LDS down from 36.9 KB to 32 KB
#endif
#define WF_WAVES_PER_SIMD(G) ((G::P > 16) ? 2 : (G::P > 8 && G::T <= 64) ? WF_WPS_2048 : (G::P <= 4) ? WF_WPS_512 : (G::P <= 8) ? WF_WPS_SMALL : WF_WPS_LARGE)

#ifdef WF_PHASE_TIMING
#define WF_STAMP(i)                                                                      \
    do {                                                                                 \
        if(threadIdx.x == 0 && a.phase_clock)                                            \
            a.phase_clock[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
    } while(0)
// where the workgroup ran: HW_ID (wave/simd/cu/sh/se) and XCC_ID, for per-CU residency timelines
#define WF_STAMP_HWID()                                                                                          \
    do {                                                                                                         \
        if(threadIdx.x == 0 && a.phase_clock) {                                                                  \
            a.phase_clock[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4); \
            a.phase_clock[(size_t)blockIdx.x * 16 + 12] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20); \
        }                                                                                                        \
    } while(0)
#elif defined(WF_EXP_CUT_AT) // development: the kernel ends at phase boundary WF_EXP_CUT_AT (what do the phases in front of it cost?)
#define WF_STAMP(i)         \
    do {                    \
        if((i) == WF_EXP_CUT_AT) \
            return;         \
    } while(0)
#define WF_STAMP_HWID()
#else
#define WF_STAMP(i)
#define WF_STAMP_HWID()
#endif

// SPLIT (SPW == 1, two captured channels, stereo display): the channels of a stream run in different workgroups, so that
// the largest geometry (139 KB of LDS for a stereo pair) fits two workgroups per CU and one loads while the other computes
// (N = 16384: 45 -> 54 % of the HBM peak).  What couples the channels is the silence state machine; a channel whose window
// has a non-zero sample needs nothing from its partner (plan_stream: it is processed and the stream is not silent), and the
// rare all-zero channel gets the partner's facts without racing it: "has a non-zero sample" by scanning the partner's
// window itself, "previous row entirely <= floor - 10" and m_last_silent from words the *previous* tick left (verdict_in,
// stream_flags) while this tick writes the next tick's copies (verdict_out, flags_out).
//
// DEC > 0: FFT sizes below the smallest geometry (N >> DEC = 256, 128 on the 512-point one).  The N >> DEC samples are
// transformed zero-padded to N points -- bin o of the small transform is bin o << DEC of the padded one, exactly -- so
// passes 1-3 run unchanged on the rows that hold samples and the epilogue keeps every (1 << DEC)-th bin: the first
// (M >> DEC) / 4 threads of the spectrum own four consecutive output bins each; rows, state and tables have M >> DEC entries.
//
// TLDS (SPW == 2): the two spectra of a workgroup use the same window and pass-1 twiddle operands, thread for thread.
// Instead of every thread loading its 2 * R1 - 1 table vectors (two thirds of the fetch burst's bytes through the
// vector-memory path, twice per workgroup), the workgroup copies both tables once by LDS-DMA into the exchange buffers --
// free until pass 1 stores into them -- and the threads take their operands from LDS; one more barrier separates those
// reads from pass 1's stores.
//
// BLU: FFT sizes that are not powers of two (Bluestein, see p1_fetch_blu): the geometry's M-point complex transform runs
// twice -- on the chirped window, then (conjugated) on its product with the chirp's transform -- and the epilogue takes
// |c_k| of the first a.row_bins bins directly, no real split.  Rows and state have a.row_bins (a run-time number) entries;
// the threads whose groups of four bins lie beyond it sit the epilogue out.
//
// BOTH (SPW == 2, mono mixdown with a curve display): the stream's one displayed row is finished by the threads of both
// spectra.  A template parameter, not a run-time flag: the extra code cost the 2048-point kernel a VGPR too many (129: three
// waves per SIMD instead of four) and 5-15 % even on configurations that never take the path.
//
// MRS (with MR; containers of one, two and four wavefronts): the mixed-radix instantiation for plans of the radices 2 ... 12 (N = 800
// as 5 x 10 x 8, 960 as 5 x 12 x 8 ...).  Without the large in-register DFTs the kernel fits 96 registers -- five waves per SIMD instead
// of four, and the tick of these sizes scales with the spectra in flight (profiles/r04g_n800_phases.txt).
// DISP: what the instantiation can display.  0: every layout (bars in the prefix-sum, piece and segment layouts, curves, the Gaussian
// filter) behind run-time flags -- the general kernel.  1: bars in the prefix-sum layout and nothing else (the host picks it when the
// handle's display is exactly that: a.bar.out != nullptr, ps_lanes > 0).  2: no display (a.bar.out == nullptr).  The run-time flags
// cost the paths that do not take them: the allocator serves the worst path and the table requests of the unused layouts stay in the
// instruction stream (profiles/r05j_bars_ps_park_cuts.txt: entering the display branch alone 0.720 -> 0.688).
// PLAN (with MRS): the mixed-radix instantiation of ONE compile-time plan (mr_fixed_plan below: the sizes the plugin picks by itself) --
// the run-time plan's dispatch over every radix, which sets the register allocation of the instantiation that carries it, is not
// compiled in.  0: every fixed plan of the container behind run-time tests, then the run-time plan.
// MIR: the instantiation that also serves wf_hip_set_bars_mirrors (every bar store repeated into up to eight further buffers,
// BarArgs::out2_delta).  Eight conditional stores at every output site and sixteen scalar registers of offsets: carried by every
// kernel (round 5) they cost the two-spectra kernels that never use them 0.5-2.4 % (profiles/r06h_mirror_instantiation_ab.txt:
// headline 0.7855 -> 0.7893, bars-only 0.623 -> 0.638) -- the handle launches this instantiation only while mirror buffers are
// set.  The split kernels (one spectrum per workgroup, N >= 8192) keep the stores in their only instantiation: MIRROR below.
// Which of the kernel's variants an instantiation is: ONE template argument with named fields (a C++20 structural type; call sites
// read `Variant{.spw = 2, .aligned = true, .disp = 2}`), each field described in the comments above.
struct Variant {
    int spw = 2;          // SPW: spectra per workgroup (2: the channels of a stream share it)
    bool aligned = false; // ALIGNED: the window starts on a 16-byte boundary of the ring for every stream (16-byte fetch)
    bool split = false;   // SPLIT
    int dec = 0;          // DEC
    bool tlds = false;    // TLDS
    bool blu = false;     // BLU
    bool both = false;    // BOTH
    bool mr = false;      // MR
    bool mrs = false;     // MRS
    bool mir = false;     // MIR
    int disp = 0;         // DISP
    int plan = 0;         // PLAN
};
template<class G, Variant V> constexpr int tick_waves_per_simd()
{
    return V.mrs ? WF_WPS_2048_MRS : (V.blu && G::P > 8 && G::T <= 64) ? WF_WPS_2048_BLU : WF_WAVES_PER_SIMD(G);
}
template<class G, Variant V>
__global__ __launch_bounds__(V.spw * G::T, (tick_waves_per_simd<G, V>())) void spectrum_tick_kernel(const TickArgs a)
{
    constexpr int SPW = V.spw, DEC = V.dec, DISP = V.disp, PLAN = V.plan;
    constexpr bool ALIGNED = V.aligned, SPLIT = V.split, TLDS = V.tlds, BLU = V.blu, BOTH = V.both, MR = V.mr, MRS = V.mrs, MIR = V.mir;
    static_assert(!MRS || (MR && G::T <= 256 && G::P > 8), "the small-radix instantiation belongs to the containers of one, two and four wavefronts");
    static_assert(!BOTH || (SPW == 2 && !SPLIT && DEC == 0 && !BLU), "shared curve row: two spectra per workgroup, power-of-two sizes");
    static_assert(!BLU || (DEC == 0 && !TLDS && !ALIGNED), "Bluestein path: scalar fetch, no decimation, no staged tables");
    static_assert(!MR || BLU, "the mixed-radix transform (wf_mixed.hpp) runs inside the Bluestein instantiation's fetch and epilogue");
    static_assert(!TLDS || (SPW == 2 && !SPLIT && DEC == 0 && (size_t)SPW * G::LDS_CF * sizeof(cf) >= 2u * G::N * sizeof(float)),
                  "staged tables: window + pass-1 twiddles must fit the workgroup's exchange buffers");
    static_assert(!SPLIT || SPW == 1, "split mode: one spectrum per workgroup");
    static_assert(DEC == 0 || (!SPLIT && G::T == 64 && (G::R1 >> DEC) >= 1 && (G::M >> DEC) >= 64), "decimated path: one-wavefront geometry");
    constexpr bool MIRROR = MIR || SPLIT;                   // this instantiation stores into wf_hip_set_bars_mirrors' buffers
    constexpr int MO_C = G::M >> DEC;                       // bins per output row (power-of-two paths)
    using RG = RowG<DEC ? MO_C / 4 : G::T, DEC ? 4 : G::P>; // threads x bins per thread that own the output rows
    constexpr int RP = RG::P;
    const int MO = BLU ? (int)a.row_bins : MO_C;            // run-time only for the Bluestein path
    const int NB = MO;                                      // row bound of the guarded row helpers (BLU)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int T = G::T, P = G::P, WPS = G::T / 64;
    const int tid = (int)threadIdx.x;
    const int sub = __builtin_amdgcn_readfirstlane(tid / T); // which spectrum of the workgroup (wave-uniform: T % 64 == 0)
    const int t = tid % T;            // thread within the spectrum
    const int lane = tid & 63;
    const int wave_in_block = tid >> 6;
    // bars in the prefix-sum layout (BarPsTables): the first wavefront of every spectrum finishes its sub-bands
    // (formed where they are used, not held from here: two lane masks across the whole kernel were four scalar registers too many
    // for the 2048-sample kernel, whose spill slot then took it over 128 vector registers)
    constexpr bool PS_OK = !BLU && DEC == 0 && !BOTH;
    static_assert(DISP != 1 || PS_OK, "the prefix-sum bars belong to the power-of-two kernels");
    // (with a display-specific instantiation the tests below are compile-time constants and the other layouts' code is gone)
    const bool has_display = DISP == 1 ? true : DISP == 2 ? false : (a.bar.out != nullptr);
#define WF_PS_MODE (PS_OK && (DISP == 1 || (DISP == 0 && a.bar.out != nullptr && a.bar.ps_lanes > 0)))
#define WF_PS_FINISHER (WF_PS_MODE && (t >> 6) == 0)
    // Prologue: nothing here may wait for memory before the window fetch is in flight.  The spectrum index is clamped
    // (no branch), stream/channel come from a shift (cap_ch is 1 or 2), and the two per-stream words (write position,
    // flags) are scalar loads issued together.
    const uint32_t n_spec = (a.stream_base + a.stream_count) * a.cap_ch; // end of this launch's slice
    uint32_t spec_raw = a.stream_base * a.cap_ch + blockIdx.x * SPW + (uint32_t)sub;
    if constexpr(SPLIT) {
        if(a.split_ch != 0xffffffffu) // one channel of every stream per launch (mono mixdown in two launches, see TickArgs)
            spec_raw = 2u * (a.stream_base + blockIdx.x) + a.split_ch;
    }
    const bool active = spec_raw < n_spec;
    const uint32_t spec = active ? spec_raw : n_spec - 1;
    const uint32_t cap_shift = a.cap_ch - 1;
    const uint32_t stream = spec >> cap_shift;
    const uint32_t ch = spec & cap_shift;
    const bool stereo = (a.mode & WF_MODE_STEREO) != 0;
    const bool mono_mix = (a.mode & WF_MODE_MONO_MIX) != 0;
    const uint32_t wpos = a.wpos[stream];
    const uint32_t sflags = a.stream_flags[stream];
    // the stream's volume-normalisation gain: read here with the other per-stream words, as a scalar load (possible while
    // the kernel has not stored anything yet).  Read where it is used -- behind the smoothing-state stores -- it was a vector
    // load under a branch whose destination register the common path had to "wait for" before reusing it: vmcnt(0), i.e.
    // the round trip of those stores, in front of the row stores of every workgroup.
    const float vol_comp = a.vol_comp_stream ? a.vol_comp_stream[stream] : a.vol_comp;
    // split mode: the previous tick's verdicts on both rows of the stream, requested with the other per-stream words
    uint32_t vin0 = 0, vin1 = 0;
    if constexpr(SPLIT) {
        vin0 = a.verdict_in[2u * stream];
        vin1 = a.verdict_in[2u * stream + 1u];
    }

    // a spectrum's exchange buffer: the geometry's, or -- mixed-radix sizes -- what the transform needs (MrPlan::lds_cf)
    const int lds_cf = MR ? a.mr.lds_cf : (int)G::LDS_CF;
    cf *lds = reinterpret_cast<cf *>(smem_raw) + (size_t)sub * lds_cf;
    cf *tw2_lds = reinterpret_cast<cf *>(smem_raw) + (size_t)SPW * lds_cf;
    int *facts = reinterpret_cast<int *>(tw2_lds + G::R2 * G::R3);
    // bars / curve display, several wavefronts per spectrum: how many of them are done with the last reads of the exchange
    // buffer (see "arrivals" below).  In the spare words behind the facts.
    int *arrivals = facts + 2 * SPW * WPS + sub;
    const float *x = a.ring + (size_t)spec * a.ring_stride;
    const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const uint32_t start = (wpos - delay - (BLU ? a.blu_n : (uint32_t)(G::N >> DEC))) & a.ring_mask;
    float *ts = a.tsmooth + (size_t)spec * MO;
    float *rows = a.decibels + (size_t)stream * a.out_ch * MO; // m_decibels[0..out_ch) of this stream
    const bool row_thread = DEC == 0 || t < RG::T;             // this thread owns bins of the output row

    // a paused stream (its source was not ticked this frame) walks through like a hidden one that is already silent: the
    // reference returns at once for those (:36-37) -- nothing is read, reset or written; its flags word is kept as it is
    const bool paused = (sflags & WF_STREAM_PAUSED) != 0;
    const bool hidden = (sflags & (WF_STREAM_HIDDEN | WF_STREAM_PAUSED)) != 0;
    const bool was_silent = (sflags & WF_STREAM_LAST_SILENT) != 0 || paused;

    WF_STAMP(0);
    WF_STAMP_HWID();
    // ---- fetch the window; per-wavefront facts for the silence state machine (reference :55-95) ----------
    // (hidden streams are fetched too: the flags word is not waited for before the loads are issued)
    P1Regs<G> r1;
    bool nz = false;
    float mr_touch[2] = {0.0f, 0.0f};
    MrOps<G> mr_ops;
    constexpr bool MR_EARLY = BLU && MR && WF_MR_PREFETCH && G::T == 64; // the epilogue's operands requested behind the window (one-wavefront containers)
    if constexpr(BLU && MR) {
        if(active) // windowed sample pairs straight into the exchange buffer, natural order (wf_mixed.hpp)
            nz = mr_fetch<G>(a, t, x, start, lds) && !hidden;
        if constexpr(MR_EARLY)
            mr_ops_request<G, true, false>(a, t, ts, mr_ops);
        // one dword of every 64-byte line of the smoothing state and the slope table of this row, requested now and never
        // used: the epilogue's own loads (it keeps no operands across the passes) find the lines in the L2 (+2.5 ... 4 % from
        // 128 threads per spectrum; -2 % on the one-wavefront geometries, where it is left out)
        if(G::T > 64 && active && 16 * t < MO) {
            mr_touch[0] = (a.mode & WF_MODE_TSMOOTH) ? ts[16 * t] : 0.0f;
            mr_touch[1] = a.slope[16 * t];
        }
    } else if constexpr(BLU) {
        if(active)
            nz = (blu_table_via_lds<G>() ? p1_fetch_blu<G>(a, t, x, start, r1) : p1_fetch_blu_direct<G>(a, t, x, start, r1)) && !hidden;
    } else if(active)
        nz = p1_fetch<G, ALIGNED, DEC, TLDS>(a, t, x, start, r1) && !hidden;
    if constexpr(TLDS) {
        lds_dma_copy<G::N * (int)sizeof(float)>(a.window, smem_raw, wave_in_block, T * SPW / 64, lane);
        lds_dma_copy<G::M * (int)sizeof(cf)>(a.tw1, smem_raw + G::N * sizeof(float), wave_in_block, T * SPW / 64, lane);
    }
    if constexpr(BLU && !MR && blu_table_via_lds<G>()) // the chirped window: staged in the (still free) exchange buffer, see p1_fetch_blu
        blu_table_to_lds(a, smem_raw, wave_in_block, T * SPW / 64, lane);
    // the workgroup's copy of the pass-2 twiddles: LDS-DMA (no staging registers), requested behind the window so that it
    // costs no round trip of its own; complete at the barrier below
    lds_dma_copy<G::R2 * G::R3 * (int)sizeof(cf)>(a.tw2, tw2_lds, wave_in_block, T * SPW / 64, lane);
    // (the Bluestein instantiations -- not the mixed-radix ones -- are at the register cap through two transforms: they fetch these
    // few words where they are used instead of holding them from here)
    BarPre bar_pre_early{0, 0, 1, 0, 0, 0, -1};
    if constexpr(!(BLU && !MR))
        if constexpr(DISP == 0)
            bar_pre_early = bars_preload<G, true, PS_OK>(a.bar, t);
    const bool wave_nz = __any(nz) != 0;
    WF_STAMP(1);
    bool wave_below = true;
    if(active && !hidden && !wave_nz) { // wave-uniform and rare: the whole slice of this wave is digital silence
        if(!SPLIT && (WF_TRACK && a.bars_only != nullptr) && a.bars_only->use_verdict) // rows in HBM are stale: the word this wave left instead
            wave_below = a.bars_only->row_verdict[(size_t)(spec - (stereo ? 0u : ch)) * WPS + (wave_in_block - sub * WPS)] == 0u;
        else
            wave_below = __all(!row_thread || row_all_below<RG, BLU>(rows + (size_t)(stereo ? ch : 0u) * MO, t, a.silent_floor, NB)) != 0;
    }

    bool nz0 = wave_nz, nz1 = false, below0 = wave_below, below1 = true;
    if(lane == 0)
        facts[wave_in_block] = (wave_nz ? 1 : 0) | (wave_below ? 2 : 0);
    if(T > 64 && t == 0)
        *arrivals = 0;
    if(PS_OK && tid < 2)
        facts[2 * SPW * WPS + 2 + tid] = 0; // (prefix-sum layout: the counters of the wavefronts that have parked their part of the row)
    __syncthreads(); // facts + the LDS twiddle table are visible to the whole workgroup
    if constexpr(SPLIT) {
        int orf = 0;
#pragma unroll
        for(int w = 0; w < WPS; ++w)
            orf |= facts[w];
        const bool nz_own = (orf & 1) != 0;
        bool nz_other = false;
        if(!nz_own && !hidden) { // workgroup-uniform and rare: this channel's window is digital silence
            const float *xo = a.ring + (size_t)(spec ^ 1u) * a.ring_stride;
            uint32_t acc = 0;
            const uint32_t wlen = BLU ? a.blu_n : (uint32_t)G::N; // the partner's window is as long as this one
            for(uint32_t i = (uint32_t)tid; i < wlen; i += (uint32_t)T)
                acc |= f32_bits(xo[(start + i) & a.ring_mask]);
            const bool w_nz = __any((acc & 0x7fffffffu) != 0) != 0;
            if(lane == 0)
                facts[WPS + wave_in_block] = w_nz ? 1 : 0;
            __syncthreads();
            int o2 = 0;
#pragma unroll
            for(int w = 0; w < WPS; ++w)
                o2 |= facts[WPS + w];
            nz_other = o2 != 0;
        }
        nz0 = ch == 0 ? nz_own : nz_other;
        nz1 = ch == 0 ? nz_other : nz_own;
        below0 = vin0 == 0u;
        below1 = stereo ? (vin1 == 0u) : (vin0 == 0u); // mono display: channel 1 inspects row 0 too (reference :81)
    } else if(T > 64 || a.cap_ch > 1) {
        const int sb0 = (sub - (int)ch) * WPS; // first wavefront of the subgroup that owns channel 0 of this stream
        int or0 = 0, and0 = 3, or1 = 0, and1 = 3;
#pragma unroll
        for(int w = 0; w < WPS; ++w) {
            const int f0 = facts[sb0 + w];
            or0 |= f0;
            and0 &= f0;
            if(a.cap_ch > 1) {
                const int f1 = facts[sb0 + WPS + w];
                or1 |= f1;
                and1 &= f1;
            }
        }
        nz0 = (or0 & 1) != 0;
        below0 = (and0 & 2) != 0;
        nz1 = (or1 & 1) != 0;
        below1 = (and1 & 2) != 0;
    }
    StreamPlan plan = plan_stream(was_silent, a.cap_ch, stereo, nz0, nz1, below0, below1);
    // Fewer samples captured than window + A/V-sync delay (reference :55-61: `continue`, the channel is left alone and
    // m_last_silent stays).  The ring is zero history before the first sample, so without this the window would be analysed
    // as digital silence.  Only ever true in the first `delay` samples after a reset (wpos starts at fft_size).
    const uint32_t fft_n = BLU ? a.blu_n : (uint32_t)(G::N >> DEC);
    const bool underflow = (!(sflags & WF_STREAM_WRAPPED) && (wpos - fft_n) < delay) || (sflags & WF_STREAM_STARVED) != 0;
    if(underflow) {
        plan.process0 = plan.process1 = false;
        plan.last_silent = was_silent;
    }
    const bool process = active && !hidden && (ch == 0 ? plan.process0 : plan.process1);
    const bool do_db = active && !hidden && !plan.last_silent; // reference :138-139

    // ---- the FFT path --------------------------------------------------------------------------------------
    cf v[P];
    float mag[RP];
    WF_STAMP(2);
    P4Regs<G> r4;
    if constexpr(BLU && MR) {
        // FFT sizes with no prime factor above 5: the n/2-point transform itself, two to four mixed-radix passes between the two
        // halves of the exchange buffer (wf_mixed.hpp) instead of Bluestein's two power-of-two transforms
        // The sizes the plugin picks by itself (sample_rate / fps & -16 at 48 and 44.1 kHz, 60 / 50 / 30 / 25 / 24 fps) run their plan as
        // compile-time constants (mr_transform_fixed): N = 800 0.430 -> 0.462 of the HBM peak at 8192 streams, 0.316 -> 0.358 at 2048.
        auto plan_is = [&](int r0, int r1, int r2) { return a.mr.passes == 3 && a.mr.radix[0] == r0 && a.mr.radix[1] == r1 && a.mr.radix[2] == r2; };
        auto sp_sync = [] { spectrum_sync<G>(); };
        static_assert(PLAN == 0 || (MRS && ((G::N == 2048 && PLAN >= 1 && PLAN <= 4) || (G::N == 4096 && PLAN >= 5 && PLAN <= 8))), "a fixed plan belongs to its container's small-radix instantiation");
        if constexpr(PLAN == 1) // 800
            mr_transform_fixed<G, 5, 10, 8>(a.mr, process, t, lds, sp_sync);
        else if constexpr(PLAN == 2) // 960
            mr_transform_fixed<G, 5, 12, 8>(a.mr, process, t, lds, sp_sync);
        else if constexpr(PLAN == 3) // 720
            mr_transform_fixed<G, 10, 6, 6>(a.mr, process, t, lds, sp_sync);
        else if constexpr(PLAN == 4) // 880
            mr_transform_fixed<G, 11, 5, 8>(a.mr, process, t, lds, sp_sync);
        else if constexpr(PLAN == 5) // 1600
            mr_transform_fixed<G, 10, 8, 10>(a.mr, process, t, lds, sp_sync);
        else if constexpr(PLAN == 6) // 1920
            mr_transform_fixed<G, 10, 8, 12>(a.mr, process, t, lds, sp_sync);
        else if constexpr(PLAN == 7) // 2000
            mr_transform_fixed<G, 10, 10, 10>(a.mr, process, t, lds, sp_sync);
        else if constexpr(PLAN == 8) // 1760
            mr_transform_fixed<G, 10, 8, 11>(a.mr, process, t, lds, sp_sync);
        else if(MRS && WF_MR_FIXED_PLANS && G::N == 2048 && plan_is(5, 10, 8)) // 800
            mr_transform_fixed<G, 5, 10, 8>(a.mr, process, t, lds, sp_sync);
        else if(MRS && WF_MR_FIXED_PLANS && G::N == 2048 && plan_is(5, 12, 8)) // 960
            mr_transform_fixed<G, 5, 12, 8>(a.mr, process, t, lds, sp_sync);
        else if(MRS && WF_MR_FIXED_PLANS && G::N == 2048 && plan_is(10, 6, 6)) // 720
            mr_transform_fixed<G, 10, 6, 6>(a.mr, process, t, lds, sp_sync);
        else if(MRS && WF_MR_FIXED_PLANS && G::N == 2048 && plan_is(11, 5, 8)) // 880
            mr_transform_fixed<G, 11, 5, 8>(a.mr, process, t, lds, sp_sync);
        else if(MRS && WF_MR_FIXED_PLANS && G::N == 4096 && plan_is(10, 8, 10)) // 1600
            mr_transform_fixed<G, 10, 8, 10>(a.mr, process, t, lds, sp_sync);
        else if(MRS && WF_MR_FIXED_PLANS && G::N == 4096 && plan_is(10, 8, 12)) // 1920
            mr_transform_fixed<G, 10, 8, 12>(a.mr, process, t, lds, sp_sync);
        else if(MRS && WF_MR_FIXED_PLANS && G::N == 4096 && plan_is(10, 10, 10)) // 2000
            mr_transform_fixed<G, 10, 10, 10>(a.mr, process, t, lds, sp_sync);
        else if(MRS && WF_MR_FIXED_PLANS && G::N == 4096 && plan_is(10, 8, 11)) // 1760
            mr_transform_fixed<G, 10, 8, 11>(a.mr, process, t, lds, sp_sync);
        else
            mr_transform<G, MRS>(a.mr, process, (int)a.row_bins, t, lds, tw2_lds, [] { spectrum_sync<G>(); }); // (tw2_lds: the prime pass's W_p^m, staged where the power-of-two kernels keep their pass-2 twiddles: TickArgs::tw2 points at it)
    } else {
        if constexpr(TLDS) {
            cf o1[G::R1][G::B1];
            if(process) {
                p1_tables_from_lds<G>(t, reinterpret_cast<const cf *>(smem_raw), r1);
                p1_window_dft<G>(r1, o1);
            }
            __syncthreads(); // every thread has taken its operands: pass 1 may overwrite the staged tables
            if(process)
                p1_store<G>(t, lds, o1);
        }
        if constexpr(BLU && blu_table_via_lds<G>()) {
            cf o1[G::R1][G::B1];
            if(process) {
                blu_products_from_lds<G>(a, t, reinterpret_cast<const cf *>(smem_raw), r1);
                p1_window_dft<G>(r1, o1);
            }
            __syncthreads(); // every thread has taken its table entries: pass 1 may overwrite the staged table
            if(process)
                p1_store<G>(t, lds, o1);
        }
        if(process) {
            if constexpr(!TLDS && !(BLU && blu_table_via_lds<G>()))
                p1_window_pass1<G>(a, t, r1, lds);
            if constexpr(DEC > 0)
                p4_prefetch_dec<G, DEC>(a, t, ts, r4);
            else if constexpr(!BLU && !Policy<G>::PREFETCH_LATE) // the Bluestein epilogue loads its (shorter) rows itself
                p4_prefetch<G>(a, t, ts, r4);
        }
        __builtin_amdgcn_sched_barrier(0); // keep the prefetch up here: do not sink it to its first use in P4
        WF_STAMP(3);
        spectrum_sync<G>();
        if(process)
            p2_read<G>(t, lds, v);
        spectrum_sync<G>();
        WF_STAMP(4);
        if(process)
            p2_pass2_write<G>(tw2_lds, t, lds, v);
        WF_STAMP(5);
        spectrum_sync<G>();
        if(process)
            p3_read<G>(t, lds, v);
        spectrum_sync<G>();
        WF_STAMP(6);
        if(process)
            p3_pass3_write<G>(t, lds, v);
        if constexpr(!BLU && DEC == 0 && Policy<G>::PREFETCH_LATE) {
            if(process)
                p4_prefetch<G>(a, t, ts, r4);
            __builtin_amdgcn_sched_barrier(0);
        }
        WF_STAMP(7);
        spectrum_sync<G>();
        if constexpr(BLU) {
            // second transform: conj(FFT(a) . FFT(b)) read from the natural-order buffer, then passes 1-3 again.
            // The thread index goes through an opaque move first: with the same `t` the compiler shares the exchange-buffer address
            // arithmetic of the two transforms and keeps the first one's addresses alive across everything in between (44-116 B of
            // scratch per lane became 24-68 B); recomputing a few integer operations is free
            int t2 = t;
    #if defined(__HIPCC__)
            asm volatile("" : "+v"(t2));
    #endif
            if(process)
                blu_mid<G>(a, t2, lds, r1);
            spectrum_sync<G>(); // every thread has read its points: pass 1 may overwrite the buffer
            if(process)
                p1_window_pass1<G>(a, t2, r1, lds);
            spectrum_sync<G>();
            if(process)
                p2_read<G>(t2, lds, v);
            spectrum_sync<G>();
            if(process)
                p2_pass2_write<G>(tw2_lds, t2, lds, v);
            spectrum_sync<G>();
            if(process)
                p3_read<G>(t2, lds, v);
            spectrum_sync<G>();
            if(process)
                p3_pass3_write<G>(t2, lds, v);
            spectrum_sync<G>();
        }
    }
    WF_STAMP(8);
    // the bar tables of this thread: requested here, in front of P4 and its state stores, where the geometry has the registers
    // (Policy<G>::BAR_COEF_EARLY) -- else behind the dB math below
    constexpr bool COEF_EARLY = Policy<G>::BAR_COEF_EARLY && !BLU && DEC == 0 && !BOTH;
    BarEntries<G> bar_entries;
    bar_entries.base = 0;
    if constexpr(COEF_EARLY) {
        if constexpr(DISP != 2)
            bars_fetch_entries<G, PS_OK, DISP == 1>(a.bar, t, bar_entries, WF_PS_FINISHER);
        if(!process && has_display)
            wait_vmem_all(); // (the rare path that skips P4 and its wait)
    }
    if(process) {
        if constexpr(BLU) {
            if constexpr(MR) {
                asm volatile("" ::"v"(mr_touch[0]), "v"(mr_touch[1])); // (waited for here at the latest; see the fetch)
#if WF_MR_PREFETCH
                if constexpr(MR_EARLY)
                    mr_ops_request<G, false, true>(a, t, ts, mr_ops);
                else
                    mr_ops_request<G, true, true>(a, t, ts, mr_ops); // all of the epilogue's operands at once, from the lines touched above
                p4_mr<G>(a, t, lds, ts, mr_ops, mag);
#else
                p4_direct<G, MR>(a, t, lds, ts, mag);
#endif
            } else
                p4_direct<G, MR>(a, t, lds, ts, mag);
        } else if constexpr(DEC > 0) {
            if(row_thread)
                p4_split_smooth_dec<G, DEC>(a, t, lds, ts, r1.wb, r4, mag);
        } else {
            p4_split_smooth<G, WF_DEFER_STATE != 0>(a, t, lds, ts, r1.wb, r4, mag);
            if(WF_DEFER_STATE && mono_mix) // (the mixdown below overwrites mag[]: no deferral there)
                p4_store_state<G>(a, t, ts, mag);
            if(Policy<G>::TOUCH_STATE)
                asm volatile("" ::"v"(r4.touch[0]), "v"(r4.touch[1])); // the touched dwords are only ever waited for
        }
    } else if(do_db && (!(mono_mix && ch == 1) || underflow) && row_thread) {
        // skipped channel of a live stream: its stale row is re-dBFS'ed (Appendix C.3).  A channel is only skipped when that
        // row is entirely <= floor - 10 < 0, and dbfs() of a negative number is DB_MIN: when the row in HBM is not current
        // (bars-only ticks) any negative stand-in gives the reference's result.  Mono mixdown adds the stale row to the
        // partner's magnitudes first, so its rows are always stored and loaded.
        // (read from a.stale_row, a row of DB_MIN in HBM: filling mag[] in place here made ROCm 7.2's clang sink the
        // store into a pointer phi over scratch and global memory, which its backend cannot select)
        // (underflow in mono mixdown: both channels take row 0, which no tick has filled yet -- DB_MIN, as m_decibels[1] is
        // in the reference at that point; the mean of two negative rows is negative: DB_MIN again)
        // (mono mixdown, channel 1: the reference adds m_decibels[1] -- channel 1's last smoothed magnitudes, left linear by
        // :150-154 -- to row 0's stale dB values.  With temporal smoothing on those magnitudes are this channel's smoothing
        // state; with smoothing off nothing holds them and row 0 stands in: a stated deviation for a starved tick in the middle
        // of a stream, where the sum is garbage either way)
        const float *stale = ((WF_TRACK && a.bars_only != nullptr) && !mono_mix) ? a.stale_row
                             : (mono_mix && ch == 1 && (a.mode & WF_MODE_TSMOOTH)) ? ts
                                                                                    : rows + (size_t)(mono_mix ? 0u : ch) * MO;
        load_row<RG, BLU>(stale, t, mag, NB);
        // Waited for here, inside the rare branch.  Left to the compiler the wait lands where the branches meet, in front of
        // the dB math of every workgroup -- and as the loads are the youngest operations of this path, it is a wait for
        // everything (vector-memory operations complete in order): the common path then sat out the acknowledgement of its
        // smoothing-state stores, a full round trip to HBM, before it took its first logarithm.
        wait_vmem_all();
    }

    // ---- hidden / capture timeout: reset branch (reference :34-48), complete in itself --------------------------
    if(active && hidden && !was_silent) {
        if((a.mode & WF_MODE_TSMOOTH) && row_thread)
            fill_row<RG, BLU>(ts, t, 0.0f, NB);
        if(ch < (stereo ? 2u : 1u)) {
            const bool dup = a.out_ch > a.cap_ch; // one captured channel shown as two rows
            if(!a.skip_decibels && row_thread) {
                fill_row<RG, BLU>(rows + (size_t)ch * MO, t, a.db_min, NB);
                if(dup)
                    fill_row<RG, BLU>(rows + (size_t)MO, t, a.db_min, NB);
            }
            if(has_display) {
                // what render_bars makes of rows of DB_MIN: every bar at border_bottom
                float *bo = a.bar.out + ((size_t)stream * a.bar.disp_ch + ch) * a.bar.num_bars;
                if(a.bar.pre_out != nullptr && t < (dup ? 2 : 1))
                    a.bar.pre_out[(size_t)stream * a.bar.disp_ch + ch + t] = a.bar.border_bottom;
                for(int i = t; i < a.bar.num_bars * (dup ? 2 : 1); i += T) {
                    bo[i] = a.bar.border_bottom;
                    if constexpr(!BLU && MIRROR) // (the Bluestein / mixed-radix instantiations, at their register caps, do not serve wf_hip_set_bars_mirrors)
#pragma unroll
                        for(int j = 0; j < 8; ++j)
                            if(j < a.bar.out2_n)
                                bo[(long long)i + a.bar.out2_delta[j]] = a.bar.border_bottom;
                }
            }
        }
    }

    // ---- end-of-tick dB pass (reference :141-179) ------------------------------------------------------------------
    if(mono_mix && SPLIT) {
        // the channels run in different launches, channel 1 first: its magnitudes wait in m_decibels[1]
        if(ch == 1) {
            if(process && row_thread)
                store_row<RG, BLU>(rows + (size_t)MO, t, mag, NB);
        } else if(do_db && row_thread) {
            float o[RP];
            load_row<RG, BLU>(rows + (size_t)MO, t, o, NB);
#pragma unroll
            for(int i = 0; i < RP; ++i)
                mag[i] = (mag[i] + o[i]) * 0.5f;
        }
    } else if(mono_mix) {
        // dB[0][i] = dbfs((dB[0][i] + dB[1][i]) * 0.5f): channel 1 hands its magnitudes to channel 0 through LDS
        float *xch = reinterpret_cast<float *>(lds);
        __syncthreads();
        if(do_db && ch == 1 && row_thread) {
#pragma unroll
            for(int u = 0; u < RP / 4; ++u)
                *reinterpret_cast<f4 *>(xch + 4 * (t + RG::T * u)) = f4{mag[4 * u], mag[4 * u + 1], mag[4 * u + 2], mag[4 * u + 3]};
        }
        __syncthreads();
        if(do_db && ch == 0 && row_thread) {
            const float *other = reinterpret_cast<const float *>(lds + lds_cf);
#pragma unroll
            for(int u = 0; u < RP / 4; ++u) {
                const f4 o = *reinterpret_cast<const f4 *>(other + 4 * (t + RG::T * u));
                mag[4 * u] = (mag[4 * u] + o.x) * 0.5f;
                mag[4 * u + 1] = (mag[4 * u + 1] + o.y) * 0.5f;
                mag[4 * u + 2] = (mag[4 * u + 2] + o.z) * 0.5f;
                mag[4 * u + 3] = (mag[4 * u + 3] + o.w) * 0.5f;
            }
        }
    }
    WF_STAMP(9);
    // The row of a bars / curve display is parked in the exchange buffer once every wavefront of the spectrum has read its
    // last points from it.  That was a workgroup barrier in front of the row's LDS stores, a few hundred cycles before the
    // one behind them; instead every wavefront counts itself in here, right behind its reads (LDS operations of a wave
    // execute in order), and checks the count before it parks its part of the row: by then the dB math has passed and the
    // others have long arrived.  (Mono mixdown inside one workgroup has barriers of its own between
    // the last reads and here.)
    const bool count_arrivals = T > 64 && has_display && (SPLIT || !mono_mix) && !BOTH;
    if(count_arrivals) {
        asm volatile("" ::: "memory");
        if(lane == 0)
            __hip_atomic_fetch_add(arrivals, 1, WF_ARRIVE_ORDER_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the bar tables of this thread: requested here so that their L2 latency runs under the dB math and the row stores
    if constexpr(!COEF_EARLY && DISP != 2)
        bars_fetch_entries<G, PS_OK, DISP == 1>(a.bar, t, bar_entries, WF_PS_FINISHER);
    if constexpr(WF_DEFER_STATE && !BLU && DEC == 0) {
        if(process && !mono_mix)
            p4_store_state<G>(a, t, ts, mag); // behind the table requests (p4_split_smooth<.., DEFER>)
    }
    const bool have_row = do_db && !(mono_mix && ch == 1); // this subgroup produces row `ch` (and row 1 too when one
                                                            // captured channel is shown as stereo, reference :141-142)
    const bool dup_row = have_row && (a.out_ch > a.cap_ch);
    float d[RP];
    bool row_exceeds = false; // bars-only handles: this thread's part of the row has a value > floor - 10
    if(have_row && row_thread) {
        p4_db<RG, BLU>(a, t, mag, d, vol_comp, NB);
        if(!SPLIT && (WF_TRACK && a.bars_only != nullptr)) {
            // taken here, where d[] is produced: reading the array again under a later branch makes ROCm 7.2's clang merge
            // that read with a global load through a pointer phi, i.e. a flat pointer into scratch, and its backend aborts
            // ("Illegal instruction detected: Operand has incorrect register class", V_CMP_NE_U32 0, src_private_base)
#pragma unroll
            for(int i = 0; i < RP; ++i)
                row_exceeds = row_exceeds || ((!BLU || 4 * (t + RG::T * (i / 4)) < NB) && d[i] > a.silent_floor);
        }
        // (Storing the rows behind the bars / curve points instead -- so that the wait for their table loads, vector memory
        // completing in order, is not a wait for these stores' acknowledgement -- measured +-0 for both, -10 % for bars at
        // N = 2048, and is no longer in the tree.)
        if(!a.skip_decibels) {
            store_row<RG, BLU, WF_NT_ROWS>(rows + (size_t)ch * MO, t, d, NB);
            if(dup_row)
                store_row<RG, BLU, WF_NT_ROWS>(rows + (size_t)MO, t, d, NB);
        }
    }
    WF_STAMP(10);
    if(active && ch == 0 && t == 0)
        (SPLIT ? a.flags_out : a.stream_flags)[stream] =
            paused ? sflags
                   : ((sflags & (WF_STREAM_HIDDEN | WF_STREAM_TIMEOUT | WF_STREAM_WRAPPED | WF_STREAM_STARVED)) | ((hidden || plan.last_silent) ? WF_STREAM_LAST_SILENT : 0u));
    if(!SPLIT && (WF_TRACK && a.bars_only != nullptr) && active) {
        // bars-only handles: what the next tick's silence test would find in the row this spectrum owns (see TickArgs)
        bool exceeds = false, write = true;
        if(have_row) {
            exceeds = row_exceeds;
        } else if(hidden && !was_silent) {
            exceeds = false; // the reset branch: rows of DB_MIN
        } else if(!a.bars_only->use_verdict && !(mono_mix && ch == 1)) {
            // first tracked tick, row untouched by it: the row in HBM is still current
            exceeds = row_thread && !row_all_below<RG, BLU>(rows + (size_t)ch * MO, t, a.silent_floor, NB);
        } else
            write = false; // untouched: the word stays
        const bool any_exceeds = __any(exceeds) != 0;
        if(write && lane == 0)
            a.bars_only->row_verdict[(size_t)spec * WPS + (wave_in_block - sub * WPS)] = any_exceeds ? 1u : 0u;
    }
    if constexpr(SPLIT) {
        // what the next tick's silence test will find in this channel's row (reference :78-86: any value > floor - 10?)
        bool exceeds = false;
        if(have_row) {
#pragma unroll
            for(int i = 0; i < RP; ++i)
                exceeds = exceeds || ((!BLU || 4 * (t + T * (i / 4)) < NB) && d[i] > a.silent_floor);
        } else if(!(hidden && !was_silent)) // row untouched this tick (the reset branch leaves DB_MIN everywhere: below)
            exceeds = (ch == 0 ? vin0 : vin1) != 0u;
        if(__any(exceeds) && lane == 0)
            atomicOr(a.verdict_out + spec, 1u);
        if(tid == 0)
            a.verdict_clear[spec] = 0u;
    }

    // ---- bars: what render_bars derives from the rows just written (reference src/source.cpp:1500-1557) ---------------
#ifndef WF_TAIL_PRIO
#define WF_TAIL_PRIO 0
#endif
    // the further bars buffers (BarArgs::out2_delta) of a spectrum whose displayed row(s) this tick leaves as they are: copied over
    if(!BLU && MIRROR && has_display && a.bar.out2_n > 0 && active && ch < a.bar.disp_ch) {
        const bool have_row_ = do_db && !(mono_mix && ch == 1);
        const bool reset_ = hidden && !was_silent && ch < (stereo ? 2u : 1u);
        if(!have_row_ && !reset_) { // (uniform over the spectrum)
            const float *bo = a.bar.out + ((size_t)stream * a.bar.disp_ch + ch) * a.bar.num_bars;
            const int n = a.bar.num_bars * ((a.out_ch > a.cap_ch) ? 2 : 1);
            for(int i = t; i < n; i += T) {
                const float v = bo[i];
#pragma unroll
                for(int j = 0; j < 8; ++j)
                    if(j < a.bar.out2_n)
                        const_cast<float *>(bo)[(long long)i + a.bar.out2_delta[j]] = v;
            }
        }
    }
    if(has_display && !WF_EXP_NO_TAIL) {
        // The display phase runs with raised issue priority: what is left of the workgroup's life is a short serial stretch (one
        // wavefront per spectrum at the end) that holds the workgroup's LDS, and on a SIMD shared with three wavefronts in their
        // transform passes every instruction of it otherwise waits its turn
        if(WF_TAIL_PRIO > 0)
            __builtin_amdgcn_s_setprio(WF_TAIL_PRIO);
        // mono mixdown displays one row per stream: its curve points are shared by the threads of both spectra of the workgroup
        // (the plugin's default configuration: 800 points, 4 steps of 256 threads instead of 7 of 128)
        constexpr bool both = BOTH;
        float *dbl = both ? reinterpret_cast<float *>(smem_raw) : reinterpret_cast<float *>(lds);
        auto row_sync = [&] {
            if(both)
                __syncthreads();
            else
                spectrum_sync<G>();
        };
        // every thread of the spectrum is done reading its exchange buffer
        if(count_arrivals) {
            while(__hip_atomic_load(arrivals, WF_ARRIVE_ORDER_ACQ, __HIP_MEMORY_SCOPE_WORKGROUP) < WPS)
                __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        } else if(!(mono_mix && !SPLIT))
            row_sync();
        if constexpr(PS_OK) {
            if(WF_PS_MODE) {
                const bool ps_finisher = (t >> 6) == 0;
                // Every wavefront leaves its part of the row and its group sums, then counts itself in (release: its LDS stores are
                // ordered in front of the count); the spectrum's first wavefront waits for the others (acquire) and finishes.  One
                // wavefront per spectrum: program order is all it takes.
                WF_STAMP(12);
                if(have_row)
                    ps_park<RG>(dbl, MO, t, d);
                WF_STAMP(14);
                int *parked = facts + 2 * SPW * WPS + 2 + sub;
                if constexpr(T > 64) {
                    if(lane == 0)
                        __hip_atomic_fetch_add(parked, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if(ps_finisher) {
                        while(__hip_atomic_load(parked, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < WPS)
                            __builtin_amdgcn_s_sleep(1);
                    }
                } else
                    __builtin_amdgcn_wave_barrier();
                WF_STAMP(15);
                if(ps_finisher) {
                    float *out0 = a.bar.out + ((size_t)stream * a.bar.disp_ch + ch) * a.bar.num_bars;
                    const BarArgs bar_row = [&] { BarArgs b = a.bar; if(!MIRROR) b.out2_n = 0; if(b.pre_out) b.pre_out += (size_t)stream * a.bar.disp_ch + ch; return b; }();
                    ps_finish<G, RG>(bar_row, bar_entries, dbl, MO, lane, have_row, out0, dup_row ? out0 + a.bar.num_bars : nullptr);
                }
                WF_STAMP(13);
                return;
            }
        }
        if constexpr(DISP == 1)
            return; // (unreachable: WF_PS_MODE is a constant here -- the other layouts are not compiled in)
        if(have_row && row_thread)
            store_row<RG, BLU>(dbl, t, d, NB);
        if(a.bar.curve == 2 && have_row && t == 0) // the Catmull-Rom taps of the last points reach bins M and M + 1 (dropped by the reference)
            dbl[MO] = dbl[MO + 1] = 0.0f;
        if((BLU && !MR) || !a.bar.piece_mode) // (wave-private bar layout: every wavefront reads back only what it has parked itself)
            row_sync();
        float *out0 = a.bar.out + ((size_t)stream * a.bar.disp_ch + ch) * a.bar.num_bars;
        float *out1 = dup_row ? out0 + a.bar.num_bars : nullptr;
#ifdef WF_PHASE_TIMING
        BarArgs bar_args = a.bar;
        if(bar_args.pre_out)
            bar_args.pre_out += (size_t)stream * a.bar.disp_ch + (BOTH ? 0u : ch);
        bar_args.clk = (a.phase_clock && threadIdx.x == 0) ? a.phase_clock + (size_t)blockIdx.x * 16 : nullptr;
        WF_STAMP(12);
#else
        // (the Bluestein / mixed-radix instantiations do not serve wf_hip_set_bars_mirror -- the host refuses it for their sizes --:
        // with the count a constant 0 the further stores fold away; the 96-register instantiation spilled on them)
        // (and BarArgs::pre_out points at the entry of the row being finished: the stream's only row when the threads of both spectra share it)
        const BarArgs bar_row = [&] {
            BarArgs b = a.bar;
            if(BLU || !MIRROR)
                b.out2_n = 0;
            if(b.pre_out)
                b.pre_out += (size_t)stream * a.bar.disp_ch + (BOTH ? 0u : ch);
            return b;
        }();
        const BarArgs &bar_args = bar_row;
#endif
        OutVals<G> ov;
        bool pending = true;
        if constexpr(BOTH) {
            {
                constexpr int TT = 2 * T;
                float *row0 = a.bar.out + (size_t)stream * a.bar.disp_ch * a.bar.num_bars; // the stream's only displayed row
                if(a.bar.stream_steps) {
                    curve_row_stream<G, TT>(bar_args, do_db, dbl, dbl, tid, row0, nullptr, [] { __syncthreads(); });
                } else {
                    if(a.bar.curve == 2)
                        curve_row_catrom<G, TT>(bar_args, do_db, dbl, tid, ov);
                    else
                        curve_row<G, TT>(bar_args, do_db, dbl, tid, ov);
                    outputs_finish<G, TT>(bar_args, do_db, ov, dbl, tid, row0, nullptr, [] { __syncthreads(); });
                }
                pending = false;
            }
        }
        if(pending && !both) {
            if(a.bar.stream_steps) {
                curve_row_stream<G>(bar_args, have_row, dbl, dbl, t, out0, out1, [] { spectrum_sync<G>(); });
                pending = false;
            } else if(a.bar.curve == 2)
                curve_row_catrom<G>(bar_args, have_row, dbl, t, ov);
            else if(a.bar.curve)
                curve_row<G>(bar_args, have_row, dbl, t, ov);
            else
                pending = bars_reduce_row<G, !(BLU && !MR)>(
                    bar_args, (BLU && !MR) ? bars_preload<G, false, false>(a.bar, t) : bar_pre_early, bar_entries, have_row, dbl, dbl + MO, t, out0, out1, ov, [] { spectrum_sync<G>(); },
                    [](float v, int m) { return v + __shfl_xor(v, m, 64); }, arrivals, count_arrivals ? 2 * WPS : WPS);
            if(pending)
                outputs_finish<G>(bar_args, have_row, ov, dbl, t, out0, out1, [] { spectrum_sync<G>(); });
        }
    }
    WF_STAMP(13);
#undef WF_PS_MODE
#undef WF_PS_FINISHER
}

} // namespace wf
