// wf_big.hpp -- gfx950 kernels of the FFT sizes whose complex transform does not fit a CU's LDS (device code only; hipcc).
//
// The reference accepts every multiple of 16 up to 65536 samples with "enable large FFT" (src/source.cpp:349, :359-363,
// :562-565).  Up to 32768 samples (16384 complex points, 132 KB) a spectrum lives in one workgroup's LDS
// (spectrum_tick_kernel).  Beyond that:
//   65536 samples                    -> 32768 complex points (the packed real transform)
//   n = 16400 .. 65520, not 2^k      -> Bluestein (direct form) over L = 32768 / 65536 / 131072 complex points (two transforms)
// A transform of L = L1 * 16384 points (L1 = 2, 4, 8) is done in two steps through a scratch buffer in device memory
// (decimation in frequency, n = n1 * 16384 + n2, k = k1 + L1 * k2):
//   big_columns_kernel   v[k1][n2] = W_L^(n2 k1) * sum_n1 u[n1 * 16384 + n2] W_L1^(n1 k1)      (u = the windowed samples, see MODE)
//   big_rows_kernel      U[k1 + L1 k2] = sum_n2 v[k1][n2] W_16384^(n2 k2)   -- the 32768-sample geometry's three LDS passes
// then big_epilogue_kernel does what P4 and the end of spectrum_tick_kernel do (real split or |c_k|, slope, temporal
// smoothing, silence state machine, dBFS, volume normalisation, roll-off; reference src/source_generic.cpp:63-179) on
// 16384 bins per workgroup, and big_outputs_kernel what render_bars / render_curve derive from the finished rows
// (src/source.cpp:1360-1425, :1500-1564).  The channels of a stream are coupled as in split mode: "has a non-zero sample"
// through a word the columns kernel ORs into, "previous row entirely <= floor - 10" through the rotating verdict words.
//
// This is the compatibility path (every transform moves its data through L2 / Infinity Cache three times); the sizes
// the slider produces by default never reach it.
#pragma once
#include <hip/hip_runtime.h>
#include "wf_geometry.hpp"
#include "wf_tick_phases.hpp"
#include "wf_mixed.hpp"

namespace wf {

// (GBig, BIG_L2, BIG_TP: wf_geometry.hpp)

struct BigArgs {
    const float *ring;
    const uint32_t *wpos;
    const uint32_t *delay_stream;
    uint32_t ring_mask, ring_stride, delay, cap_ch;
    uint32_t n;              // samples per window (m_fft_size)
    uint32_t L;              // complex points per transform (L1 * 16384)
    const float *window;     // [n] (MODE 0)
    const cf *blu_a;         // [L] window_j * conj(w_j), zero from n on (MODE 1)
    const cf *blu_b;         // [L] FFT_L of the chirp (MODE 2)
    const cf *tw_big;        // [L1][16384] W_L^(n2 k1)
    const cf *tw1, *tw2;     // the row transform's tables
    cf *v;                   // [n_spec][L1][16384] columns' output
    cf *z;                   // [n_spec][L] rows' output, natural order
    uint32_t *nz;            // [n_spec] != 0: the window has a non-zero sample
    uint32_t spec_base;      // first spectrum of this launch
};

// MODE 0: u[j] = (x[2j] w[2j], x[2j+1] w[2j+1])  (packed real transform of n = 2L samples, reference :97-106)
// MODE 1: u[j] = x[j] * blu_a[j]                  (Bluestein, first transform: the chirped window)
// MODE 2: u[j] = conj(z[j] * blu_b[j])            (Bluestein, second transform)
template<int L1, int MODE> __global__ __launch_bounds__(256) void big_columns_kernel(const BigArgs a)
{
    const uint32_t spec = a.spec_base + blockIdx.y;
    const uint32_t n2 = 2u * (blockIdx.x * 256u + threadIdx.x); // this thread's pair of columns
    cf u0[L1], u1[L1];
    uint32_t acc = 0;
    if constexpr(MODE == 2) {
        const cf *zin = a.z + (size_t)spec * a.L;
#pragma unroll
        for(int n1 = 0; n1 < L1; ++n1) {
            const uint32_t j = (uint32_t)n1 * BIG_L2 + n2;
            const f4 s = ld4(reinterpret_cast<const float *>(zin + j));
            const f4 b = ld4(reinterpret_cast<const float *>(a.blu_b + j));
            const cf p0 = cmul(cf{s.x, s.y}, cf{b.x, b.y}), p1 = cmul(cf{s.z, s.w}, cf{b.z, b.w});
            u0[n1] = cf{p0.x, -p0.y};
            u1[n1] = cf{p1.x, -p1.y};
        }
    } else {
        const uint32_t stream = spec >> (a.cap_ch - 1u);
        const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
        const uint32_t start = (a.wpos[stream] - delay - a.n) & a.ring_mask;
        const float *x = a.ring + (size_t)spec * a.ring_stride;
#pragma unroll
        for(int n1 = 0; n1 < L1; ++n1) {
            const uint32_t j = (uint32_t)n1 * BIG_L2 + n2;
            if constexpr(MODE == 0) {
                float s[4];
#pragma unroll
                for(uint32_t e = 0; e < 4; ++e) {
                    s[e] = x[(start + 2u * j + e) & a.ring_mask];
                    acc |= f32_bits(s[e]);
                }
                const f4 w = ld4(a.window + 2u * j);
                u0[n1] = cf{s[0] * w.x, s[1] * w.y};
                u1[n1] = cf{s[2] * w.z, s[3] * w.w};
            } else {
                const bool in0 = j < a.n, in1 = j + 1u < a.n;
                const float v0 = x[(start + (in0 ? j : 0u)) & a.ring_mask], v1 = x[(start + (in1 ? j + 1u : 0u)) & a.ring_mask];
                const float s0 = in0 ? v0 : 0.0f, s1 = in1 ? v1 : 0.0f;
                acc |= f32_bits(s0) | f32_bits(s1);
                const f4 q = ld4(reinterpret_cast<const float *>(a.blu_a + j));
                u0[n1] = cf{s0 * q.x, s0 * q.y};
                u1[n1] = cf{s1 * q.z, s1 * q.w};
            }
        }
    }
    dft_dif<L1>(u0);
    dft_dif<L1>(u1);
    cf *v = a.v + (size_t)spec * a.L;
    constexpr int LB = ilog2(L1);
#pragma unroll
    for(int k1 = 0; k1 < L1; ++k1) {
        cf c0 = u0[brev(k1, LB)], c1 = u1[brev(k1, LB)];
        if(k1 > 0) {
            const f4 w = ld4(reinterpret_cast<const float *>(a.tw_big + (size_t)k1 * BIG_L2 + n2));
            c0 = cmul(c0, cf{w.x, w.y});
            c1 = cmul(c1, cf{w.z, w.w});
        }
        st4(reinterpret_cast<float *>(v + (size_t)k1 * BIG_L2 + n2), f4{c0.x, c0.y, c1.x, c1.y});
    }
    if constexpr(MODE != 2) {
        // x != 0.0f for any sample of the window (reference :63-72): -0.0f is zero, NaNs are not
        if(__any((acc & 0x7fffffffu) != 0u) && (threadIdx.x & 63u) == 0u)
            atomicOr(a.nz + spec, 1u);
    }
}

// one row of 16384 points per workgroup: the three LDS passes of the 32768-sample geometry, then out in natural order
template<int L1> __global__ __launch_bounds__(GBig::T, 4) void big_rows_kernel(const BigArgs a)
{
    using G = GBig;
    constexpr int T = G::T, P = G::P, R1 = G::R1, M1 = G::M1;
    static_assert(G::B1 == 1, "the row loader fetches one point per pass-1 row");
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    cf *lds = reinterpret_cast<cf *>(big_smem);
    cf *tw2_lds = lds + G::LDS_CF;
    const int t = (int)threadIdx.x;
    const uint32_t k1 = blockIdx.x, spec = a.spec_base + blockIdx.y;
    const cf *v = a.v + (size_t)spec * a.L + (size_t)k1 * BIG_L2;
    TickArgs ta{};
    ta.tw1 = a.tw1;
    P1Regs<G> r;
#pragma unroll
    for(int j = 0; j < R1; ++j) {
        const f2 q = ld2(reinterpret_cast<const float *>(v + j * M1 + t));
        r.smp[j][0] = q.x;
        r.smp[j][1] = q.y;
        r.win[j][0] = r.win[j][1] = 1.0f;
        if(j >= 1 && tw1_row_loaded(j))
            p1_load_tw1<G>(ta, t, j, r.tw1[j]);
    }
    // the pass-2 twiddles by LDS-DMA, as in spectrum_tick_kernel
    {
        constexpr int BYTES = G::R2 * G::R3 * (int)sizeof(cf), PER = 64 * 16;
        const int wave = t >> 6, lane = t & 63;
#pragma unroll
        for(int c = 0; c < BYTES / PER; ++c)
            if((c % (T / 64)) == wave) {
                const char *g = reinterpret_cast<const char *>(a.tw2) + c * PER + lane * 16;
                char *l = reinterpret_cast<char *>(tw2_lds) + c * PER;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16,
                                                 0, 0);
            }
    }
    p1_window_pass1<G>(ta, t, r, lds);
    __syncthreads();
    cf pts[P];
    p2_read<G>(t, lds, pts);
    __syncthreads();
    p2_pass2_write<G>(tw2_lds, t, lds, pts);
    __syncthreads();
    p3_read<G>(t, lds, pts);
    __syncthreads();
    p3_pass3_write<G>(t, lds, pts);
    __syncthreads();
    cf *z = a.z + (size_t)spec * a.L;
#pragma unroll
    for(int i = 0; i < P; ++i) {
        const int k2 = t + T * i;
        const cf c = lds_ld2(lds, ex3_addr<G>(k2));
        *reinterpret_cast<f2 *>(z + (size_t)k1 + (size_t)L1 * k2) = f2{c.x, c.y};
    }
}
template<int L1> constexpr size_t big_rows_lds_bytes() { return (size_t)GBig::LDS_CF * sizeof(cf) + (size_t)GBig::R2 * GBig::R3 * sizeof(cf) + 16; }

// ---- epilogue ------------------------------------------------------------------------------------------------------
// MODE 1: bins from the packed transform Z of big_m points: 2X[k] = (Z[k] + conj Z[m-k]) - i W_2m^k (Z[k] - conj Z[m-k])
// MODE 2: bins are |c_k| of the Bluestein convolution
template<int MODE, bool TS, bool FPK>
WF_DEV void p4_big_impl(const TickArgs &a, int t, int kbase, int nb, const cf *z, float *ts, float (&mag)[GBig::P])
{
    using G = GBig;
    constexpr int T = G::T, P = G::P;
    const int m = (int)a.big_m;
#pragma unroll
    for(int u = 0; u < P / 4; ++u) {
        const int k0 = 4 * (t + T * u);
#pragma unroll
        for(int i = 0; i < 4; ++i)
            mag[4 * u + i] = 0.0f;
        if(k0 >= nb)
            continue;
        const int kk = kbase + k0;
        const f4 sv = ld4(a.slope + kk);
        f4 st = f4{0.0f, 0.0f, 0.0f, 0.0f};
        if(TS)
            st = ld_state(ts + k0);
        cf A[4];
        // MODE 3: MODE 1 on rows that lie where big_br_rows_kernel left them: Z[k] at z[(k % big_c) big_rs + k / big_c], big_c a power of two
        const uint32_t cs = (uint32_t)__builtin_ctz(a.big_c), cm = a.big_c - 1u;
        auto zt = [&](int k) { return z + (size_t)((uint32_t)k & cm) * a.big_rs + ((uint32_t)k >> cs); };
        if constexpr(MODE == 3) {
#pragma unroll
            for(int i = 0; i < 4; ++i) {
                const f2 q = ld2(reinterpret_cast<const float *>(zt(kk + i)));
                A[i] = cf{q.x, q.y};
            }
        } else {
            const f4 za = ld4(reinterpret_cast<const float *>(z + kk)), zb = ld4(reinterpret_cast<const float *>(z + kk + 2));
            A[0] = cf{za.x, za.y}; A[1] = cf{za.z, za.w}; A[2] = cf{zb.x, zb.y}; A[3] = cf{zb.z, zb.w};
        }
        if constexpr(MODE == 1 || MODE == 3) {
            const f4 wa = ld4(reinterpret_cast<const float *>(a.big_tws + kk)), wb = ld4(reinterpret_cast<const float *>(a.big_tws + kk + 2));
            const cf W[4] = {cf{wa.x, wa.y}, cf{wa.z, wa.w}, cf{wb.x, wb.y}, cf{wb.z, wb.w}};
#pragma unroll
            for(int i = 0; i < 4; ++i) {
                const int km = (kk + i) == 0 ? 0 : m - kk - i; // Z[m] is Z[0]; m need not be a power of two
                const f2 bq = ld2(reinterpret_cast<const float *>(MODE == 3 ? zt(km) : z + km));
                const float er = A[i].x + bq.x, ei = A[i].y - bq.y;
                const float dr = A[i].x - bq.x, di = A[i].y + bq.y;
                const float pr = fmaf(W[i].x, dr, -(W[i].y * di));
                const float pi = fmaf(W[i].x, di, W[i].y * dr);
                mag[4 * u + i] = mag2(er + pi, ei - pr) * a.half_coef;
            }
        } else {
#pragma unroll
            for(int i = 0; i < 4; ++i)
                mag[4 * u + i] = mag2(A[i].x, A[i].y) * a.half_coef;
        }
        const float sl4[4] = {sv.x, sv.y, sv.z, sv.w}, st4v[4] = {st.x, st.y, st.z, st.w};
        p4_slope_smooth_group<G, TS, FPK>(a, t, u, ts, st4v, sl4, mag);
    }
}
template<int MODE> WF_DEV void p4_big(const TickArgs &a, int t, int kbase, int nb, const cf *z, float *ts, float (&mag)[GBig::P])
{
    if(a.mode & WF_MODE_TSMOOTH) {
        if(a.mode & WF_MODE_FAST_PEAKS)
            p4_big_impl<MODE, true, true>(a, t, kbase, nb, z, ts, mag);
        else
            p4_big_impl<MODE, true, false>(a, t, kbase, nb, z, ts, mag);
    } else
        p4_big_impl<MODE, false, false>(a, t, kbase, nb, z, ts, mag);
}

// What follows a spectrum's magnitudes (the end of spectrum_tick_kernel, on rows of RG::T * RG::P bins starting at bin kbase): the
// silence state machine from the per-channel facts, slope / smoothing / state through fill_mag(ts, mag) when the channel is
// processed, the reset branch, the mono mixdown, dB and the rows, the verdict words of the next tick's silence test.
//   nz_ch0 / nz_ch1: the channel's window holds a non-zero sample (reference :63-72)
template<class RG, class FillMag>
WF_DEV void big_finish(const TickArgs &a, int t, int kbase, uint32_t spec, bool nz_ch0, bool nz_ch1, FillMag &&fill_mag)
{
    constexpr int RP = RG::P;
    const int lane = t & 63;
    const uint32_t cap_shift = a.cap_ch - 1;
    const uint32_t stream = spec >> cap_shift, ch = spec & cap_shift;
    const bool stereo = (a.mode & WF_MODE_STEREO) != 0;
    const bool mono_mix = (a.mode & WF_MODE_MONO_MIX) != 0;
    const uint32_t wpos = a.wpos[stream];
    const uint32_t sflags = a.stream_flags[stream];
    const int MO = (int)a.row_bins;
    const int NB = MO - kbase; // bins of this part (the row helpers skip groups at or beyond it)
    float *ts = a.tsmooth + (size_t)spec * MO + kbase;
    float *rows = a.decibels + (size_t)stream * a.out_ch * MO + kbase; // row r of this stream at rows + r * MO
    const bool paused = (sflags & WF_STREAM_PAUSED) != 0;
    const bool hidden = (sflags & (WF_STREAM_HIDDEN | WF_STREAM_PAUSED)) != 0;
    const bool was_silent = (sflags & WF_STREAM_LAST_SILENT) != 0 || paused;
    // the facts of the silence state machine (reference :63-95), as in split mode
    const uint32_t s0 = stream * a.cap_ch;
    const uint32_t vin0 = a.verdict_in[s0], vin1 = a.cap_ch > 1 ? a.verdict_in[s0 + 1u] : 0u;
    const bool nz0 = !hidden && nz_ch0;
    const bool nz1 = a.cap_ch > 1 && !hidden && nz_ch1;
    const bool below0 = vin0 == 0u;
    const bool below1 = stereo ? (vin1 == 0u) : (vin0 == 0u); // mono display: channel 1 inspects row 0 too (reference :81)
    StreamPlan plan = plan_stream(was_silent, a.cap_ch, stereo, nz0, nz1, below0, below1);
    const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const bool underflow = (!(sflags & WF_STREAM_WRAPPED) && (wpos - a.blu_n) < delay) || (sflags & WF_STREAM_STARVED) != 0; // reference :55-61
    if(underflow) {
        plan.process0 = plan.process1 = false;
        plan.last_silent = was_silent;
    }
    const bool process = !hidden && (ch == 0 ? plan.process0 : plan.process1);
    const bool do_db = !hidden && !plan.last_silent; // reference :138-139

    float mag[RP];
#pragma unroll
    for(int i = 0; i < RP; ++i)
        mag[i] = 0.0f;
    if(process)
        fill_mag(ts, mag);
    else if(do_db && (!(mono_mix && ch == 1) || underflow)) // skipped channel of a live stream: its stale row is re-dBFS'ed
        load_row<RG, true>(rows + (size_t)(mono_mix ? 0u : ch) * MO, t, mag, NB);

    // hidden / capture timeout: reset branch (reference :34-48)
    if(hidden && !was_silent) {
        if(a.mode & WF_MODE_TSMOOTH)
            fill_row<RG, true>(ts, t, 0.0f, NB);
        if(ch < (stereo ? 2u : 1u)) {
            fill_row<RG, true>(rows + (size_t)ch * MO, t, a.db_min, NB);
            if(a.out_ch > a.cap_ch)
                fill_row<RG, true>(rows + (size_t)MO, t, a.db_min, NB);
        }
    }
    // mono mixdown (reference :150-154): channel 1 ran a launch ahead and left its magnitudes in m_decibels[1]
    if(mono_mix) {
        if(ch == 1) {
            if(process)
                store_row<RG, true>(rows + (size_t)MO, t, mag, NB);
        } else if(do_db) {
            float o[RP];
#pragma unroll
            for(int i = 0; i < RP; ++i)
                o[i] = 0.0f;
            load_row<RG, true>(rows + (size_t)MO, t, o, NB);
#pragma unroll
            for(int i = 0; i < RP; ++i)
                mag[i] = (mag[i] + o[i]) * 0.5f;
        }
    }
    const bool have_row = do_db && !(mono_mix && ch == 1);
    const bool dup_row = have_row && (a.out_ch > a.cap_ch);
    float d[RP];
    bool exceeds = false;
    if(have_row) {
        p4_db<RG, true>(a, t, mag, d, a.vol_comp_stream ? a.vol_comp_stream[stream] : a.vol_comp, NB, kbase);
#pragma unroll
        for(int i = 0; i < RP; ++i)
            exceeds = exceeds || (4 * (t + RG::T * (i / 4)) < NB && d[i] > a.silent_floor);
        store_row<RG, true, WF_NT_ROWS>(rows + (size_t)ch * MO, t, d, NB);
        if(dup_row)
            store_row<RG, true, WF_NT_ROWS>(rows + (size_t)MO, t, d, NB);
    } else if(!(hidden && !was_silent)) // row untouched this tick (the reset branch leaves DB_MIN everywhere: below)
        exceeds = (ch == 0 ? vin0 : vin1) != 0u;
    if(ch == 0 && t == 0)
        a.flags_out[stream] = paused ? sflags
                                     : ((sflags & (WF_STREAM_HIDDEN | WF_STREAM_TIMEOUT | WF_STREAM_WRAPPED | WF_STREAM_STARVED)) |
                                        ((hidden || plan.last_silent) ? WF_STREAM_LAST_SILENT : 0u));
    // what the next tick's silence test will find in this channel's row (reference :78-86: any value > floor - 10?)
    if(__any(exceeds) && lane == 0)
        atomicOr(a.verdict_out + spec, 1u);
    if(t == 0)
        a.verdict_clear[spec] = 0u;
}

// grid (parts, spectra): workgroup (part, spec) finishes bins [part * 16384, +16384) of the spectrum's row
template<int MODE> __global__ __launch_bounds__(GBig::T, 4) void big_epilogue_kernel(const TickArgs a)
{
    using G = GBig;
    const int t = (int)threadIdx.x;
    const int kbase = (int)blockIdx.x * BIG_TP;
    uint32_t spec = a.stream_base * a.cap_ch + blockIdx.y;
    if(a.split_ch != 0xffffffffu) // mono mixdown: one channel of every stream per launch, channel 1 first
        spec = 2u * (a.stream_base + blockIdx.y) + a.split_ch;
    const uint32_t s0 = (spec >> (a.cap_ch - 1u)) * a.cap_ch;
    const int NB = (int)a.row_bins - kbase;
    big_finish<RowG<G::T, G::P>>(a, t, kbase, spec, a.big_nz[s0] != 0u, a.cap_ch > 1 && a.big_nz[s0 + 1u] != 0u, [&](float *ts, float (&mag)[G::P]) {
        p4_big<MODE>(a, t, kbase, NB, a.big_z + (size_t)spec * a.big_l, ts, mag);
    });
}

// ---- fft_size 65536: one kernel ---------------------------------------------------------------------------------------------
// The packed real transform of 65536 samples is L = 2 x 16384 complex points.  Decimation in frequency over n1 (two columns)
// leaves two independent 16384-point rows, and row k1 delivers exactly the bins of parity k1: U[k1 + 2 k2].  The real split
// pairs bin k with bin m - k (m = 32768) -- the SAME parity -- so a row needs nothing from the other one.  big_whole_kernel
// therefore (1) forms the rows' inputs itself: windowed sample pairs straight from the ring, u0 + u1 for row 0 and
// (u0 - u1) W_L^n2 for row 1 (what big_columns_kernel writes to a scratch buffer for the other sizes), (2) transforms row 0 and
// then row 1 in the exchange buffer, each followed by its real split into registers, and (3) finishes the tick from those
// registers.  Device-memory traffic per transform: 26 N bytes (columns, rows, epilogue: round 2) -> 18 N (round 3: rows with the
// column step and the split folded in, magnitudes through memory, epilogue) -> 10 N, what the tick itself needs; three kernels ->
// two -> one.

// the 16 points per row this thread feeds into pass 1: u0 +- u1 (times W_L^n2 for row 1), u = windowed sample pairs.
// Eight to ten registers per point are in flight until its sums are formed, so the burst goes out in groups of rows.
#ifndef BIG_FETCH_ROWS
#define BIG_FETCH_ROWS 4
#endif
// (32-bit byte offsets from a uniform base: the loads take the SGPR-base + VGPR-offset form instead of a 64-bit address pair
// per request -- sixty-four of those do not fit the register file)
#ifdef BIG_PLAIN_INDEX
WF_DEV float big_ring1(const float *x, uint32_t i) { return x[i]; }
WF_DEV f2 big_ring2(const float *x, uint32_t i) { return ld2(x + i); }
WF_DEV f4 big_ring4(const float *x, uint32_t i) { return ld4(x + i); }
#else
WF_DEV float big_ring1(const float *x, uint32_t i) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(x) + (size_t)(i * 4u)); }
WF_DEV f2 big_ring2(const float *x, uint32_t i) { return *reinterpret_cast<const f2 *>(reinterpret_cast<const char *>(x) + (size_t)(i * 4u)); }
WF_DEV f4 big_ring4(const float *x, uint32_t i) { return *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(x) + (size_t)(i * 4u)); }
#endif
// The row transform of the fused path runs on 512 threads of 32 points (two waves per SIMD, 256 registers each) rather than
// GBig's 1024 of 16: at 128 registers the fused fetch and split spilled 196 B per lane, and with one workgroup per CU the
// spill traffic of 1024 workgroups went all the way to device memory -- 420 MB of the 770 MB a launch moved for 201 MB of
// payload (profiles/r03a_n65536_pmc.json).
#ifndef WF_FOLD_T
#define WF_FOLD_T 512
#endif
using GFold = Geom<32768, WF_FOLD_T, 16, 32, 32>;
static_assert(GFold::LDS_CF == GBig::LDS_CF && GFold::R2 == GBig::R2 && GFold::R3 == GBig::R3 && GFold::R1 == GBig::R1, "the fused rows use GBig's tables and LDS budget");

// Both rows from ONE pass over the window: row 0's sums u0 + u1 are parked in the exchange buffer, in the very slots this thread's
// pass-1 outputs will take, and come back as r.smp (all sixteen in registers during the burst do not fit); row 1's
// (u0 - u1) W_L^n2 stay in r1.smp -- 64 registers that wait while row 0 is transformed.
template<class G, bool ALIGNED, int BURST = BIG_FETCH_ROWS> WF_DEV uint32_t big_fused_fetch2(const TickArgs &a, int t, const float *x, uint32_t start, P1Regs<G> &r, P1Regs<G> &r1, cf *lds)
{
    constexpr int R1 = G::R1, M1 = G::M1, B1 = G::B1;
    constexpr int ROWS = (ALIGNED ? BURST : BURST / 2) / B1 > 0 ? (ALIGNED ? BURST : BURST / 2) / B1 : 1;
    uint32_t acc = 0;
    const float *tw_row = reinterpret_cast<const float *>(a.big_tw + (size_t)BIG_L2); // row 1 of the table
#pragma unroll
    for(int j0 = 0; j0 < R1; j0 += ROWS) {
        f2 s0[ROWS][B1], s1[ROWS][B1], w0[ROWS][B1], w1[ROWS][B1], tw[ROWS][B1];
#pragma unroll
        for(int jj = 0; jj < ROWS; ++jj) {
            const int j = j0 + jj;
            const uint32_t n2 = (uint32_t)(j * M1 + B1 * t); // this thread's B1 consecutive points of pass-1 row j
            const uint32_t i0 = start + 2u * n2, i1 = start + 2u * (n2 + BIG_L2);
            if(ALIGNED && B1 == 2) { // start is a multiple of 4: a 16-byte group never straddles the ring's wrap
                const f4 q0 = big_ring4(x, i0 & a.ring_mask), q1 = big_ring4(x, i1 & a.ring_mask);
                s0[jj][0] = f2{q0.x, q0.y}; s0[jj][B1 - 1] = f2{q0.z, q0.w};
                s1[jj][0] = f2{q1.x, q1.y}; s1[jj][B1 - 1] = f2{q1.z, q1.w};
            } else {
#pragma unroll
                for(int b = 0; b < B1; ++b) {
                    const uint32_t e0 = i0 + 2u * b, e1 = i1 + 2u * b;
                    if(ALIGNED) {
                        s0[jj][b] = big_ring2(x, e0 & a.ring_mask);
                        s1[jj][b] = big_ring2(x, e1 & a.ring_mask);
                    } else {
                        s0[jj][b] = f2{big_ring1(x, e0 & a.ring_mask), big_ring1(x, (e0 + 1u) & a.ring_mask)};
                        s1[jj][b] = f2{big_ring1(x, e1 & a.ring_mask), big_ring1(x, (e1 + 1u) & a.ring_mask)};
                    }
                }
            }
            if(B1 == 2) {
                const f4 a0 = big_ring4(a.window, 2u * n2), a1 = big_ring4(a.window, 2u * (n2 + BIG_L2)), tq = big_ring4(tw_row, 2u * n2);
                w0[jj][0] = f2{a0.x, a0.y}; w0[jj][B1 - 1] = f2{a0.z, a0.w};
                w1[jj][0] = f2{a1.x, a1.y}; w1[jj][B1 - 1] = f2{a1.z, a1.w};
                tw[jj][0] = f2{tq.x, tq.y}; tw[jj][B1 - 1] = f2{tq.z, tq.w};
            } else {
                w0[jj][0] = big_ring2(a.window, 2u * n2);
                w1[jj][0] = big_ring2(a.window, 2u * (n2 + BIG_L2));
                tw[jj][0] = big_ring2(tw_row, 2u * n2);
            }
            if(j >= 1 && tw1_row_loaded(j))
                p1_load_tw1<G>(a, t, j, r.tw1[j]);
        }
#pragma unroll
        for(int jj = 0; jj < ROWS; ++jj) {
            const int j = j0 + jj;
            cf pt[B1];
#pragma unroll
            for(int b = 0; b < B1; ++b) {
                const f2 p0 = s0[jj][b], p1 = s1[jj][b], v0 = w0[jj][b], v1 = w1[jj][b], q = tw[jj][b];
                acc |= f32_bits(p0.x) | f32_bits(p0.y) | f32_bits(p1.x) | f32_bits(p1.y);
                const cf u0 = cf{p0.x * v0.x, p0.y * v0.y}, u1 = cf{p1.x * v1.x, p1.y * v1.y};
                pt[b] = cf{u0.x + u1.x, u0.y + u1.y};
                const cf d = cmul(cf{u0.x - u1.x, u0.y - u1.y}, cf{q.x, q.y});
                r1.smp[j][2 * b] = d.x;
                r1.smp[j][2 * b + 1] = d.y;
            }
            // parked in the exchange buffer, in the very slots this thread's pass-1 outputs will take (p1_store): the sixteen
            // sums of a thread would otherwise occupy 2 R1 B1 registers for the whole burst
            if(B1 == 2)
                lds_st4(lds, ex1_addr<G>(j, B1 * t), pt[0], pt[B1 - 1]);
            else
                lds_st2(lds, ex1_addr<G>(j, t), pt[0]);
        }
#ifndef BIG_NO_FENCE
        if(j0 + ROWS < R1)
            __builtin_amdgcn_sched_barrier(0); // the next group's requests stay behind this group's sums
#endif
    }
    // ... and back: the thread's own slots, read through an index the compiler cannot match with the stores above (it
    // would forward the stored values and keep them in registers after all)
    int tr = t;
    asm volatile("" : "+v"(tr));
#pragma unroll
    for(int j = 0; j < R1; ++j) {
        if(B1 == 2) {
            const f4 q = lds_ld4(lds, ex1_addr<G>(j, B1 * tr));
            r.smp[j][0] = q.x; r.smp[j][1] = q.y; r.smp[j][2 * B1 - 2] = q.z; r.smp[j][2 * B1 - 1] = q.w;
        } else {
            const cf q = lds_ld2(lds, ex1_addr<G>(j, tr));
            r.smp[j][0] = q.x; r.smp[j][1] = q.y;
        }
#pragma unroll
        for(int b = 0; b < 2 * B1; ++b)
            r.win[j][b] = r1.win[j][b] = 1.0f;
    }
    return acc;
}


// bins of parity K1 from row K1's transform (natural order in LDS): mag[4 u + 2 h + K1] = |2 X[k]| coef / 2 for
// k = 4 (t + T u) + 2 h + K1, i.e. Z[k2] with k2 = 2 (t + T u) + h and its mirror image m - k, which is row K1's
// k2' = (16384 - K1 - k2) mod 16384
template<class G, int K1> WF_DEV void big_fused_split(const TickArgs &a, int t, const cf *lds, float (&out)[2 * G::P])
{
    constexpr int T = G::T, P = G::P;
    constexpr int WSTEP = 32 / ((2 * 2 * (int)BIG_L2) / (4 * T)); // W_65536^(4 T u) = W_32^(WSTEP u)
    static_assert(WSTEP >= 1 && WSTEP * (P / 2) <= 16 && 4 * T * 32 == 2 * 2 * (int)BIG_L2 * WSTEP, "the bins of a thread are W_32 steps apart");
    // W_65536^k for k = 4 t + 2 h + K1; the bins 4 T u further on are that times W_32^(WSTEP u) (compile-time constants),
    // as p4_split_smooth forms its twiddles.  out[4 u + 2 h + K1] is bin 4 (t + T u) + 2 h + K1.
    cf wh[2];
#pragma unroll
    for(int h = 0; h < 2; ++h) {
        const f2 w = ld2(reinterpret_cast<const float *>(a.big_tws + 4 * t + 2 * h + K1));
        wh[h] = cf{w.x, w.y};
    }
#pragma unroll
    for(int u = 0; u < P / 2; ++u) {
#pragma unroll
        for(int h = 0; h < 2; ++h) {
            const int k2 = 2 * (t + T * u) + h;
            const int km = ((int)BIG_L2 - K1 - k2) & ((int)BIG_L2 - 1);
            const cf A = lds_ld2(lds, ex3_addr<G>(k2)), B = lds_ld2(lds, ex3_addr<G>(km));
            const cf w = mul_w32(wh[h], WSTEP * u);
            const float er = A.x + B.x, ei = A.y - B.y;
            const float dr = A.x - B.x, di = A.y + B.y;
            const float pr = fmaf(w.x, dr, -(w.y * di));
            const float pi = fmaf(w.x, di, w.y * dr);
            out[2 * (2 * u + h) + K1] = mag2(er + pi, ei - pr) * a.half_coef;
        }
    }
}

// One workgroup of GFold's 512 threads runs row 0 and then row 1 of its spectrum through the exchange buffer and keeps each row's
// 16384 magnitudes in registers -- 32 per thread and row, and a thread's magnitudes of the two rows are exactly the four-bin
// groups of the row layout the end of the tick works on: row k1's k2 = 2 (t + T u) + h is bin 4 (t + T u) + 2 h + k1.  State,
// slope table, roll-off and the dB rows are then read and written as whole 16-byte groups, once.  The window is read once as
// well (big_fused_fetch2); the silence facts come from the fetch itself as in spectrum_tick_kernel's split mode.
// Against round 3's two kernels (magnitudes through device memory, 1.45 x the algorithmic bytes, profiles/r04f_n65536_pmc.json):
// 256 stereo streams 0.336 -> 0.445 of the HBM peak on one lane, 0.557 on two (profiles/r04g_n65536_whole.txt).
#ifndef WF_WHOLE_BURST
#define WF_WHOLE_BURST 2 // (rows per burst of the joint fetch: 2 is +3 % over BIG_FETCH_ROWS' 4 here -- row 1's sums fill up the registers)
#endif
// the three LDS passes of the 32768-sample geometry on the row whose pass-1 inputs are in r, then its real split into mag
template<class G, int K1>
WF_DEV void big_whole_row(const TickArgs &a, int t, P1Regs<G> &r, cf *lds, const cf *tw2_lds, float (&mag)[2 * G::P])
{
    constexpr int P = G::P;
    p1_window_pass1<G>(a, t, r, lds);
    __syncthreads();
    cf pts[P];
    p2_read<G>(t, lds, pts);
    __syncthreads();
    p2_pass2_write<G>(tw2_lds, t, lds, pts);
    __syncthreads();
    p3_read<G>(t, lds, pts);
    __syncthreads();
    p3_pass3_write<G>(t, lds, pts);
    __syncthreads();
    big_fused_split<G, K1>(a, t, lds, mag);
}

// slope, smoothing and the state store of the whole row.  The operands of WF_WHOLE_BATCH groups of four bins are requested
// together and consumed behind them: written group by group (load, load, wait, store) the state stores fence the next group's
// loads -- the compiler cannot tell the rows apart -- and a thread pays sixteen trips to device memory one after the other
// (20 of a workgroup's 46 us, profiles/r04g_n65536_whole_cuts.txt).
#ifndef WF_WHOLE_BATCH
#define WF_WHOLE_BATCH 8
#endif
// (measured and dropped, profiles/r04g_n65536_whole.txt: the state row's lines touched ahead of the fetch -- 0.557 -> 0.452 with two
// lanes --; the first batch requested in front of row 1's real split -- the fetch's register allocation falls over: 800 B of scratch)
template<class RG, bool TS, bool FPK> WF_DEV void p4_whole_impl(const TickArgs &a, int t, float *ts, float (&mag)[RG::P])
{
    constexpr int NB = WF_WHOLE_BATCH;
    static_assert((RG::P / 4) % NB == 0);
#pragma unroll
    for(int u0 = 0; u0 < RG::P / 4; u0 += NB) {
        f4 sv[NB], st[NB];
#pragma unroll
        for(int i = 0; i < NB; ++i) {
            const int k0 = 4 * (t + RG::T * (u0 + i));
            sv[i] = ld4(a.slope + k0);
            st[i] = TS ? ld_state(ts + k0) : f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for(int i = 0; i < NB; ++i) {
            const float sl4[4] = {sv[i].x, sv[i].y, sv[i].z, sv[i].w}, st4v[4] = {st[i].x, st[i].y, st[i].z, st[i].w};
            p4_slope_smooth_group<RG, TS, FPK>(a, t, u0 + i, ts, st4v, sl4, mag);
        }
    }
}

// one spectrum per workgroup (grid: spectra; mono mixdown: one channel of every stream per launch, TickArgs::split_ch)
template<bool ALIGNED> __global__ __launch_bounds__(GFold::T, GFold::T / 256) void big_whole_kernel(const TickArgs a)
{
    using G = GFold;
    using RG = RowG<G::T, 2 * G::P>;
    constexpr int T = G::T;
    static_assert(RG::T * RG::P == 2 * (int)BIG_L2, "a workgroup finishes the whole row of 32768 bins");
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    cf *lds = reinterpret_cast<cf *>(big_smem);
    cf *tw2_lds = lds + G::LDS_CF;
    const int t = (int)threadIdx.x;
    uint32_t spec = a.stream_base * a.cap_ch + blockIdx.x;
    if(a.split_ch != 0xffffffffu)
        spec = 2u * (a.stream_base + blockIdx.x) + a.split_ch;
    const uint32_t cap_shift = a.cap_ch - 1u;
    const uint32_t stream = spec >> cap_shift, ch = spec & cap_shift;
    const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const uint32_t start = (a.wpos[stream] - delay - (uint32_t)(2 * G::N)) & a.ring_mask;
    const float *x = a.ring + (size_t)spec * a.ring_stride;
    {   // the pass-2 twiddles by LDS-DMA, as in spectrum_tick_kernel (their own region: both rows use them)
        constexpr int BYTES = G::R2 * G::R3 * (int)sizeof(cf), PER = 64 * 16;
        const int wave = t >> 6, lane = t & 63;
#pragma unroll
        for(int c = 0; c < BYTES / PER; ++c)
            if((c % (T / 64)) == wave) {
                const char *g = reinterpret_cast<const char *>(a.tw2) + c * PER + lane * 16;
                char *l = reinterpret_cast<char *>(tw2_lds) + c * PER;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16,
                                                 0, 0);
            }
    }
    float mag[RG::P];
    P1Regs<G> r0, r1;
    const uint32_t acc = big_fused_fetch2<G, ALIGNED, WF_WHOLE_BURST>(a, t, x, start, r0, r1, lds);
    big_whole_row<G, 0>(a, t, r0, lds, tw2_lds, mag);
#pragma unroll
    for(int j = 1; j < G::R1; ++j) // (the pass-1 twiddles again: row 0's copies are not kept across its transform)
        if(tw1_row_loaded(j))
            p1_load_tw1<G>(a, t, j, r1.tw1[j]);
    __syncthreads(); // row 1's pass 1 writes where row 0's transform has just been read
    big_whole_row<G, 1>(a, t, r1, lds, tw2_lds, mag);
    // x != 0.0f for any sample of the window (reference :63-72): a row's fetch sees all of it; the partner channel's window is
    // looked at only when this one is digital silence (workgroup-uniform and rare), as in spectrum_tick_kernel's split mode
    const bool nz_own = __syncthreads_or((acc & 0x7fffffffu) != 0u) != 0;
    bool nz_other = false;
    if(a.cap_ch > 1 && !nz_own) {
        const float *xo = a.ring + (size_t)(spec ^ 1u) * a.ring_stride;
        uint32_t o = 0;
        for(uint32_t i = (uint32_t)t; i < (uint32_t)(2 * G::N); i += (uint32_t)T)
            o |= f32_bits(xo[(start + i) & a.ring_mask]);
        nz_other = __syncthreads_or((o & 0x7fffffffu) != 0u) != 0;
    }
    big_finish<RG>(a, t, 0, spec, ch == 0 ? nz_own : nz_other, ch == 0 ? nz_other : nz_own, [&](float *ts, float (&m)[RG::P]) {
#pragma unroll
        for(int i = 0; i < RG::P; ++i)
            m[i] = mag[i];
        if(a.mode & WF_MODE_TSMOOTH) {
            if(a.mode & WF_MODE_FAST_PEAKS)
                p4_whole_impl<RG, true, true>(a, t, ts, m);
            else
                p4_whole_impl<RG, true, false>(a, t, ts, m);
        } else
            p4_whole_impl<RG, false, false>(a, t, ts, m);
    });
}

// ---- fft sizes above 16384 with small prime factors: big_c rows of a mixed-radix transform ------------------------------------
// n/2 = C R complex points, R <= 8192 with a mixed-radix plan (wf_mixed.hpp), C <= 8.  Decimation in frequency over the C
// columns: row k1 transforms a[n2] = (sum_c z[n2 + R c] W_C^(c k1)) W_(n/2)^(n2 k1), n2 < R, and delivers the bins
// Z[k1 + C k2].  As in big_whole_kernel the column step is folded into the fetch -- every row reads the whole window, the
// rows of a spectrum sit eight workgroup indices apart (same XCD, one trip to device memory) --, the R points go through the
// passes between the two halves of the exchange buffer, and the last pass stores Z in natural order; big_epilogue_kernel<1>
// does the real split (it pairs k with n/2 - k: another row) and everything behind it.  Replaces Bluestein through device
// memory (four kernels, L >= 3 n / 2 points three times through memory) for these sizes: 48000 = 3 x 8000, 32000 = 2 x 8000, ...
__global__ __launch_bounds__(GBig::T, 4) void big_mr_rows_kernel(const TickArgs a)
{
    using G = GBig;
    constexpr int T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    cf *lds = reinterpret_cast<cf *>(big_smem);
    const int t = (int)threadIdx.x;
    const uint32_t C = a.big_c, R = a.big_r;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t k1 = slot % C, rel = (slot / C) * 8u + xcd;
    if(rel >= a.stream_count * a.cap_ch)
        return;
    const uint32_t spec = a.stream_base * a.cap_ch + rel;
    const uint32_t stream = spec >> (a.cap_ch - 1u);
    const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const uint32_t start = (a.wpos[stream] - delay - a.blu_n) & a.ring_mask;
    const float *x = a.ring + (size_t)spec * a.ring_stride;
    const cf *wc = a.big_wc + k1 * 8u;
    const cf *twc = a.big_tw + (size_t)k1 * R;
    uint32_t acc = 0;
    if(((start | R) & 1u) == 0u && (start & 3u) == 0u) {
        // (uniform) the window starts on a 16-byte boundary of the ring and the rows have an even length: two points per
        // request -- samples, window coefficients and column twiddles as 16-byte vectors
        for(uint32_t n2 = 2u * (uint32_t)t; n2 < R; n2 += 2u * (uint32_t)T) {
            cf s0 = cf{0.0f, 0.0f}, s1 = s0;
            for(uint32_t c = 0; c < C; ++c) {
                const uint32_t idx = n2 + R * c;
                const f4 xs = ld4(x + ((start + 2u * idx) & a.ring_mask)), w = ld4(a.window + 2u * idx);
                acc |= f32_bits(xs.x) | f32_bits(xs.y) | f32_bits(xs.z) | f32_bits(xs.w);
                const cf u0 = cf{xs.x * w.x, xs.y * w.y}, u1 = cf{xs.z * w.z, xs.w * w.w};
                if(k1 == 0) {
                    s0 = cadd(s0, u0);
                    s1 = cadd(s1, u1);
                } else {
                    s0 = cadd(s0, cmul(u0, wc[c]));
                    s1 = cadd(s1, cmul(u1, wc[c]));
                }
            }
            if(k1 != 0) {
                const f4 q = ld4(reinterpret_cast<const float *>(twc + n2));
                s0 = cmul(s0, cf{q.x, q.y});
                s1 = cmul(s1, cf{q.z, q.w});
            }
            lds_st4(lds, (int)n2, s0, s1);
        }
    } else {
        for(uint32_t n2 = (uint32_t)t; n2 < R; n2 += (uint32_t)T) {
            cf s = cf{0.0f, 0.0f};
            for(uint32_t c = 0; c < C; ++c) {
                const uint32_t idx = n2 + R * c, si = start + 2u * idx;
                const float x0 = x[si & a.ring_mask], x1 = x[(si + 1u) & a.ring_mask];
                const f2 w = ld2(a.window + 2u * idx);
                acc |= f32_bits(x0) | f32_bits(x1);
                const cf u = cf{x0 * w.x, x1 * w.y};
                s = (k1 == 0) ? cadd(s, u) : cadd(s, cmul(u, wc[c]));
            }
            if(k1 != 0) {
                const f2 q = ld2(reinterpret_cast<const float *>(twc + n2));
                s = cmul(s, cf{q.x, q.y});
            }
            lds_st2(lds, (int)n2, s);
        }
    }
    // x != 0.0f for any sample of the window (reference :63-72): row 0 has seen all of it
    if(k1 == 0 && __any((acc & 0x7fffffffu) != 0u) && (t & 63) == 0)
        atomicOr(a.big_nz_out + spec, 1u);
    cf *wp_lds = lds + G::LDS_CF; // behind the exchange buffer: the prime pass's W_p^m (128 entries at most)
    if(a.mr.radix[0] > 25 && t < a.mr.radix[0])
        wp_lds[t] = a.mr.wp[t];
    cf *z = const_cast<cf *>(a.big_z) + (size_t)spec * a.big_l; // (the epilogue's input; this kernel is its producer)
    mr_transform_to<G>(a.mr, true, (int)R, t, lds, wp_lds, [] { __syncthreads(); }, [=](int k2, cf v) {
        *reinterpret_cast<f2 *>(z + (size_t)k2 * C + k1) = f2{v.x, v.y};
    });
}

// ---- fft sizes above 16384 whose n/2 is TWO rows of a mixed-radix transform: one kernel, as for 65536 --------------------------------
// n/2 = 2 R, R <= 8192 with a plan on 512 threads.  What big_whole_kernel does with two 16384-point power-of-two rows, with the rows
// transformed by wf_mixed.hpp's passes: ONE pass over the window forms row 0's inputs u0 + u1 (into the exchange buffer, natural
// order) and row 1's (u0 - u1) W_(n/2)^n2 (16 complex per thread, in registers while row 0 is transformed); row k1 delivers the bins
// of parity k1 and the real split pairs bin k with bin n/2 - k -- the same parity --, so each row is split right behind its transform
// into the four-bin groups of the row layout (row k1's k2 = 2 (t + T u) + h is bin 4 (t + T u) + 2 h + k1); big_finish ends the
// tick from those registers.  No scratch in device memory, one kernel instead of rows + epilogue (85 of the slider's positions:
// 16448 ... 32704).
template<class RG, bool TS, bool FPK> WF_DEV void p4_part_impl(const TickArgs &a, int t, float *ts, int nb, float (&mag)[RG::P])
{
    constexpr int NBT = WF_WHOLE_BATCH; // (as p4_whole_impl, on a row that need not fill the layout: groups at or beyond nb sit out)
    static_assert((RG::P / 4) % NBT == 0);
#pragma unroll
    for(int u0 = 0; u0 < RG::P / 4; u0 += NBT) {
        f4 sv[NBT], st[NBT];
#pragma unroll
        for(int i = 0; i < NBT; ++i) {
            const int k0 = 4 * (t + RG::T * (u0 + i)), at = k0 < nb ? k0 : 0;
            sv[i] = ld4(a.slope + at);
            st[i] = TS ? ld_state(ts + at) : f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for(int i = 0; i < NBT; ++i) {
            const float sl4[4] = {sv[i].x, sv[i].y, sv[i].z, sv[i].w}, st4v[4] = {st[i].x, st[i].y, st[i].z, st[i].w};
            if(4 * (t + RG::T * (u0 + i)) < nb)
                p4_slope_smooth_group<RG, TS, FPK>(a, t, u0 + i, ts, st4v, sl4, mag);
        }
    }
}
// bins of parity K1 from row K1's transform (Z_row[k2] at mr_z_addr): out[4 u + 2 h + K1] for k2 = 2 (t + T u) + h < R
template<int T, int K1, int RP> WF_DEV void big_mrw_split(const TickArgs &a, int t, int R, const cf *lds, float (&out)[RP])
{
#pragma unroll
    for(int u = 0; u < RP / 4; ++u) {
#pragma unroll
        for(int h = 0; h < 2; ++h) {
            const int k2 = 2 * (t + T * u) + h;
            if(k2 < R) {
                const int k = 2 * k2 + K1;
                const int km = K1 ? R - 1 - k2 : (k2 == 0 ? 0 : R - k2); // (n/2 - k) / 2 within the row; Z[n/2] is Z[0]
                const cf A = lds_ld2(lds, mr_z_addr(a.mr, k2)), B = lds_ld2(lds, mr_z_addr(a.mr, km));
                const f2 w = ld2(reinterpret_cast<const float *>(a.big_tws + k));
                const float er = A.x + B.x, ei = A.y - B.y;
                const float dr = A.x - B.x, di = A.y + B.y;
                const float pr = fmaf(w.x, dr, -(w.y * di));
                const float pi = fmaf(w.x, di, w.y * dr);
                out[4 * u + 2 * h + K1] = mag2(er + pi, ei - pr) * a.half_coef;
            }
        }
    }
}
__global__ __launch_bounds__(GFold::T, GFold::T / 256) void big_mr_whole_kernel(const TickArgs a)
{
    using G = GFold;                 // (1024 threads of 8 points: 46 registers spilled at the 128 four waves per SIMD leave, 0.074 -> 0.090 ms at 32000)
    constexpr int T = G::T, PT = 16; // 16 points per thread and row: R <= 8192
    using RG = RowG<T, 32>;          // 16384 bins: n / 2 <= 16384
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    cf *lds = reinterpret_cast<cf *>(big_smem);
    const int t = (int)threadIdx.x;
    uint32_t spec = a.stream_base * a.cap_ch + blockIdx.x;
    if(a.split_ch != 0xffffffffu)
        spec = 2u * (a.stream_base + blockIdx.x) + a.split_ch;
    const uint32_t cap_shift = a.cap_ch - 1u;
    const uint32_t stream = spec >> cap_shift, ch = spec & cap_shift;
    const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const uint32_t start = (a.wpos[stream] - delay - a.blu_n) & a.ring_mask;
    const float *x = a.ring + (size_t)spec * a.ring_stride;
    const int R = (int)a.big_r;
    cf *wp_lds = lds + 2 * a.mr.half; // behind the exchange buffer: the prime pass's W_p^m
    if(a.mr.radix[0] > 25 && t < a.mr.radix[0])
        wp_lds[t] = a.mr.wp[t];
    cf r1[PT];
    uint32_t acc = 0;
#pragma unroll
    for(int i = 0; i < PT; ++i) {
        const int n2 = t + T * i;
        const bool in = n2 < R;
        const uint32_t j0 = in ? (uint32_t)n2 : 0u, j1 = j0 + (uint32_t)R, s0 = start + 2u * j0, s1 = start + 2u * j1;
        // (the pairs as 8-byte requests where the window starts on an even sample: 0.074 -> 0.077 ms at 32000 -- measured, left out)
        const float x00 = x[s0 & a.ring_mask], x01 = x[(s0 + 1u) & a.ring_mask], x10 = x[s1 & a.ring_mask], x11 = x[(s1 + 1u) & a.ring_mask];
        const f2 w0 = ld2(a.window + 2u * j0), w1 = ld2(a.window + 2u * j1);
        const f2 q = ld2(reinterpret_cast<const float *>(a.big_tw + (size_t)R + j0)); // W_(n/2)^n2: row 1 of the column twiddles
        acc |= f32_bits(x00) | f32_bits(x01) | f32_bits(x10) | f32_bits(x11);
        const cf u0 = cf{x00 * w0.x, x01 * w0.y}, u1 = cf{x10 * w1.x, x11 * w1.y};
        r1[i] = cmul(csub(u0, u1), cf{q.x, q.y});
        if(in)
            lds_st2(lds, n2, cadd(u0, u1));
    }
    float mag[RG::P];
#pragma unroll
    for(int i = 0; i < RG::P; ++i)
        mag[i] = 0.0f;
    mr_transform<G>(a.mr, true, R, t, lds, wp_lds, [] { __syncthreads(); });
    big_mrw_split<T, 0>(a, t, R, lds, mag);
    __syncthreads(); // every thread has read row 0's Z: row 1's inputs may go in
#pragma unroll
    for(int i = 0; i < PT; ++i)
        if(t + T * i < R)
            lds_st2(lds, t + T * i, r1[i]);
    mr_transform<G>(a.mr, true, R, t, lds, wp_lds, [] { __syncthreads(); });
    big_mrw_split<T, 1>(a, t, R, lds, mag);
    // x != 0.0f for any sample of the window (reference :63-72), the partner channel as in big_whole_kernel
    const bool nz_own = __syncthreads_or((acc & 0x7fffffffu) != 0u) != 0;
    bool nz_other = false;
    if(a.cap_ch > 1 && !nz_own) {
        const float *xo = a.ring + (size_t)(spec ^ 1u) * a.ring_stride;
        uint32_t o = 0;
        for(uint32_t i = (uint32_t)t; i < a.blu_n; i += (uint32_t)T)
            o |= f32_bits(xo[(start + i) & a.ring_mask]);
        nz_other = __syncthreads_or((o & 0x7fffffffu) != 0u) != 0;
    }
    const int nb = (int)a.row_bins;
    big_finish<RG>(a, t, 0, spec, ch == 0 ? nz_own : nz_other, ch == 0 ? nz_other : nz_own, [&](float *ts, float (&m)[RG::P]) {
#pragma unroll
        for(int i = 0; i < RG::P; ++i)
            m[i] = mag[i];
        if(a.mode & WF_MODE_TSMOOTH) {
            if(a.mode & WF_MODE_FAST_PEAKS)
                p4_part_impl<RG, true, true>(a, t, ts, nb, m);
            else
                p4_part_impl<RG, true, false>(a, t, ts, nb, m);
        } else
            p4_part_impl<RG, false, false>(a, t, ts, nb, m);
    });
}

// ---- fft sizes above 16384 with no mixed-radix plan (or one that opens with a prime pass): big_c rows, each by Bluestein INSIDE LDS ---
// (round 5; until then the sizes without a plan -- about 300 of the 768 positions of the reference's FFT-size slider above 16384,
// src/source.cpp:359-363 -- ran Bluestein over 3n/2 .. points through device memory: five kernels, 0.01-0.03 of the roofline.)
// n/2 = C R, C = 16 for the multiples of 32 (every slider position), 8 for the other multiples of 16: decimation in frequency over
// the C columns,
//   a[k1][n2] = (sum_c z[n2 + R c] W_C^(c k1)) W_(n/2)^(n2 k1) conj(w_n2)      (big_br_columns_kernel; w_m = exp(i pi m^2 / R):
//                                                                             column twiddle and opening chirp are ONE table, big_tw)
// and the R-point DFT of row k1 -- it delivers Z[k1 + C k2], k2 < R -- by Bluestein over the container geometry G of L = G::M >= 2 R - 1
// complex points (big_br_rows_kernel: the phase functions of the fused kernel's Bluestein instantiation): FFT_L, times FFT_L(chirp)
// (blu_b), conjugate, FFT_L again, Z_row[k2] = blu_q[k2] conj(R_k2), written over the row's input.  The epilogue of the packed real
// transform (big_epilogue_kernel<3>: <1> reading the rows where they lie) takes it from there.
// Scratch: [n_spec][C][big_rs] complex, big_rs = R rounded up to even (16-byte rows); a.big_l = C big_rs, the spectrum's stride.
// First form (one kernel: every row summing the columns in its own fetch, as big_mr_rows_kernel does): every row workgroup reads the
// WHOLE window -- C x the requests, 2 C x 16 dependent round trips in front of two transforms: 425 us of 494 at 48016 x 256 streams.
template<int C> __global__ __launch_bounds__(256) void big_br_columns_kernel(const TickArgs a)
{
    const uint32_t R = a.big_r;
    const uint32_t spec = a.stream_base * a.cap_ch + blockIdx.y;
    const uint32_t stream = spec >> (a.cap_ch - 1u);
    const uint32_t delay = a.delay + (a.delay_stream ? a.delay_stream[stream] : 0u);
    const uint32_t start = (a.wpos[stream] - delay - a.blu_n) & a.ring_mask;
    const float *x = a.ring + (size_t)spec * a.ring_stride;
    const uint32_t n2 = blockIdx.x * 256u + threadIdx.x;
    const bool in = n2 < R;
    const uint32_t n2c = in ? n2 : 0u;
    cf v[C];
    uint32_t acc = 0;
    if((start & 1u) == 0u) { // (uniform) the window starts on an even sample: a pair is one aligned 8-byte request (it cannot straddle the ring's wrap)
#pragma unroll
        for(int c = 0; c < C; ++c) {
            const uint32_t i = n2c + R * (uint32_t)c;
            const f2 xs = ld2(x + ((start + 2u * i) & a.ring_mask)), w = ld2(a.window + 2u * i);
            acc |= f32_bits(xs.x) | f32_bits(xs.y);
            v[c] = cf{xs.x * w.x, xs.y * w.y};
        }
    } else {
#pragma unroll
        for(int c = 0; c < C; ++c) {
            const uint32_t i = n2c + R * (uint32_t)c, si = start + 2u * i;
            const float x0 = x[si & a.ring_mask], x1 = x[(si + 1u) & a.ring_mask];
            const f2 w = ld2(a.window + 2u * i);
            acc |= f32_bits(x0) | f32_bits(x1);
            v[c] = cf{x0 * w.x, x1 * w.y};
        }
    }
    dft_dif<C>(v); // X[k1] in v[brev(k1)]
    if(in) {
        cf *out = const_cast<cf *>(a.big_z) + (size_t)spec * a.big_l + n2;
#pragma unroll
        for(int k1 = 0; k1 < C; ++k1) {
            const f2 q = ld2(reinterpret_cast<const float *>(a.big_tw + (size_t)k1 * R + n2));
            const cf o = cmul(v[brev(k1, ilog2(C))], cf{q.x, q.y});
            *reinterpret_cast<f2 *>(out + (size_t)k1 * a.big_rs) = f2{o.x, o.y};
        }
    }
    // x != 0.0f for any sample of the window (reference :63-72)
    if(__any(in && (acc & 0x7fffffffu) != 0u) && (threadIdx.x & 63u) == 0u)
        atomicOr(a.big_nz_out + spec, 1u);
}

template<class G> constexpr size_t big_br_lds_bytes() { return (size_t)G::LDS_CF * sizeof(cf) + (size_t)G::R2 * G::R3 * sizeof(cf) + 16; }
template<class G> __global__ __launch_bounds__(G::T, 4) void big_br_rows_kernel(const TickArgs a)
{
    constexpr int T = G::T, P = G::P, R1 = G::R1, B1 = G::B1, M1 = G::M1;
    static_assert(B1 == 2, "two points -- one 16-byte request -- per thread and pass-1 row");
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    cf *lds = reinterpret_cast<cf *>(big_smem);
    cf *tw2_lds = lds + G::LDS_CF;
    const int t = (int)threadIdx.x;
    const uint32_t C = a.big_c, R = a.big_r;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3; // (the rows of a spectrum on one XCD: the epilogue reads them together)
    const uint32_t k1 = slot % C, rel = (slot / C) * 8u + xcd;
    if(rel >= a.stream_count * a.cap_ch)
        return;
    const uint32_t spec = a.stream_base * a.cap_ch + rel;
    cf *row = const_cast<cf *>(a.big_z) + (size_t)spec * a.big_l + (size_t)k1 * a.big_rs; // (in: the columns' output; out: Z of this row)
    P1Regs<G> r;
#pragma unroll
    for(int j = 0; j < R1; ++j) {
        const uint32_t idx = (uint32_t)(j * M1 + B1 * t);
        const bool in0 = idx < R, in1 = idx + 1u < R;
        const f4 q = ld4(reinterpret_cast<const float *>(row + (in0 ? idx : 0u)));
        r.smp[j][0] = in0 ? q.x : 0.0f;
        r.smp[j][1] = in0 ? q.y : 0.0f;
        r.smp[j][2] = in1 ? q.z : 0.0f;
        r.smp[j][3] = in1 ? q.w : 0.0f;
#pragma unroll
        for(int e = 0; e < 4; ++e)
            r.win[j][e] = 1.0f;
        if(j >= 1 && tw1_row_loaded(j))
            p1_load_tw1<G>(a, t, j, r.tw1[j]);
    }
    { // the pass-2 twiddles by LDS-DMA, as in spectrum_tick_kernel
        constexpr int BYTES = G::R2 * G::R3 * (int)sizeof(cf), PER = 64 * 16;
        static_assert(BYTES % PER == 0, "pass-2 twiddle table in whole wave-wide requests");
        const int wave = t >> 6, lane = t & 63;
#pragma unroll
        for(int c = 0; c < BYTES / PER; ++c)
            if((c % (T / 64)) == wave) {
                const char *g = reinterpret_cast<const char *>(a.tw2) + c * PER + lane * 16;
                char *l = reinterpret_cast<char *>(tw2_lds) + c * PER;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16,
                                                 0, 0);
            }
    }
    cf pts[P];
    auto transform = [&] {
        p1_window_pass1<G>(a, t, r, lds);
        __syncthreads();
        p2_read<G>(t, lds, pts);
        __syncthreads();
        p2_pass2_write<G>(tw2_lds, t, lds, pts);
        __syncthreads();
        p3_read<G>(t, lds, pts);
        __syncthreads();
        p3_pass3_write<G>(t, lds, pts);
        __syncthreads();
    };
    transform();
    blu_mid<G>(a, t, lds, r); // conj(FFT(y) . FFT(chirp)) back into the fetch registers
    __syncthreads();          // every thread has read its points: pass 1 may overwrite the buffer
    transform();
    for(uint32_t k2 = (uint32_t)t; k2 < R; k2 += (uint32_t)T) {
        const cf v = lds_ld2(lds, ex3_addr<G>((int)k2));
        const f2 q = ld2(reinterpret_cast<const float *>(a.blu_q + k2));
        const cf o = cmul(cf{v.x, -v.y}, cf{q.x, q.y});
        *reinterpret_cast<f2 *>(row + k2) = f2{o.x, o.y};
    }
}

// ---- render-time outputs from the finished rows --------------------------------------------------------------------
// One workgroup per displayed row: bars (one wavefront per bar over the flat coefficient tables, BarArgs, the bins read from the
// row in device memory) or curve points (the row parked in LDS, curve_row_stream); the Gaussian filter stages its inputs in LDS
// (bars: at its start; curve: behind the row).
__global__ __launch_bounds__(GBig::T) void big_outputs_kernel(const TickArgs a)
{
    using G = GBig;
    constexpr int T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    float *dbl = reinterpret_cast<float *>(big_smem);
    const int t = (int)threadIdx.x;
    const uint32_t stream = a.stream_base + blockIdx.x / a.bar.disp_ch, r = blockIdx.x % a.bar.disp_ch;
    // (BarArgs::pre_out points at the entry of the row being finished)
    const BarArgs b = [&] { BarArgs c = a.bar; if(c.pre_out) c.pre_out += (size_t)stream * a.bar.disp_ch + r; return c; }();
    const int MO = (int)a.row_bins;
    const float *row = a.decibels + ((size_t)stream * a.out_ch + r) * MO;
    float *out_row = b.out + ((size_t)stream * b.disp_ch + r) * b.num_bars;
    if(b.curve) {
        // a curve point takes its taps from anywhere in the row: the row is parked in LDS first
        for(int i = 4 * t; i < MO; i += 4 * T)
            st4(dbl + i, ld4(row + i));
        if(t == 0)
            dbl[MO] = dbl[MO + 1] = 0.0f; // the Catmull-Rom taps of the last points (curve_row_stream)
        __syncthreads();
        curve_row_stream<G>(b, true, dbl, dbl, t, out_row, nullptr, [] { __syncthreads(); });
        return;
    }
    // bars: every bin is read once, by the wavefront of its bar, straight from the row (the tick kernel or the epilogue stored it
    // a moment ago: L2 / Infinity Cache) -- no parked row, so the workgroup needs LDS only for the filter's staging and many of
    // them share a CU (parked, a 32768-bin row was 128 KB: one workgroup per CU, 67 us for 512 rows)
    const int n = b.num_bars;
    const bool filtered = b.gauss_radius > 0;
    const int pad = b.gauss_radius - 1, size = 2 * b.gauss_radius - 1;
    float *vp = dbl + b.stage_off;  // [pad | n | pad]
    float *wl = vp + n + 2 * pad;   // [size]
    if(filtered) {
        for(int i = t; i < pad; i += T) {
            vp[i] = 0.0f;
            vp[pad + n + i] = 0.0f;
        }
        for(int i = t; i < size; i += T)
            wl[i] = b.gauss[i];
    }
    const int wave = t >> 6, lane = t & 63;
    float *part = filtered ? wl + size : dbl; // [num_tasks] partial sums, behind the filter's staging
    for(int task = wave; task < b.big_num_tasks; task += T / 64) {
        const int e0 = b.big_task[4 * task + 1], e1 = b.big_task[4 * task + 2], bin0 = b.big_task[4 * task + 3];
        float acc = 0.0f;
        int e = e0 + lane;
        if(bin0 >= 0) {
            // (wave-uniform) consecutive bins: coefficient and bin of an entry are requested together -- one trip to the L2 per batch
            // instead of two, sixteen entries of a lane in flight; the sums are formed in the same order as below
            const int shift = bin0 - e0; // entry e multiplies bin e + shift
            for(; e + 15 * 64 < e1; e += 16 * 64) {
                float cv[16], rv[16];
#pragma unroll
                for(int i = 0; i < 16; ++i) {
                    cv[i] = b.coef[e + 64 * i];
                    rv[i] = row[e + shift + 64 * i];
                }
#pragma unroll
                for(int i = 0; i < 16; ++i)
                    acc = fmaf(rv[i], cv[i], acc);
            }
            for(; e + 3 * 64 < e1; e += 4 * 64) {
                float cv[4], rv[4];
#pragma unroll
                for(int i = 0; i < 4; ++i) {
                    cv[i] = b.coef[e + 64 * i];
                    rv[i] = row[e + shift + 64 * i];
                }
#pragma unroll
                for(int i = 0; i < 4; ++i)
                    acc = fmaf(rv[i], cv[i], acc);
            }
            for(; e < e1; e += 64)
                acc = fmaf(row[e + shift], b.coef[e], acc);
        }
        // (else) eight of a lane's (index, coefficient, bin) triples in flight at a time instead of one dependent chain per entry
        for(; e + 7 * 64 < e1; e += 8 * 64) {
            int bi[8];
            float cv[8], rv[8];
#pragma unroll
            for(int i = 0; i < 8; ++i) {
                bi[i] = b.bin[e + 64 * i];
                cv[i] = b.coef[e + 64 * i];
            }
#pragma unroll
            for(int i = 0; i < 8; ++i)
                rv[i] = row[bi[i]];
#pragma unroll
            for(int i = 0; i < 8; ++i)
                acc = fmaf(rv[i], cv[i], acc);
        }
        for(; e < e1; e += 64)
            acc = fmaf(row[b.bin[e]], b.coef[e], acc);
#pragma unroll
        for(int m = 32; m >= 1; m >>= 1)
            acc += __shfl_xor(acc, m, 64);
        if(lane == 0)
            part[task] = acc;
    }
    __syncthreads();
    for(int bar = t; bar < n; bar += T) { // the tasks of a bar, added in their order
        float acc = 0.0f;
        for(int task = b.big_bar_task[bar]; task < b.big_bar_task[bar + 1]; ++task)
            acc += part[task];
        const float v = acc / (float)b.count[bar];
        if(filtered)
            vp[pad + bar] = v;
        else
            emit_output(b, bar, v, out_row, nullptr);
    }
    if(filtered) {
        __syncthreads();
        for(int o = t; o < n; o += T) {
            float sum = 0.0f;
            for(int tap = 0; tap < size; ++tap)
                sum = fmaf(vp[o + tap], wl[tap], sum);
            emit_output(b, o, sum / b.gauss_wsum[o], out_row, nullptr);
        }
    }
}

} // namespace wf
