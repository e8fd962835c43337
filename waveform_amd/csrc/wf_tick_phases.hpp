// wf_tick_phases.hpp -- thread-level phases of the fused spectrum tick.
//
// One spectrum (= one channel of one stream, one call of the per-channel loop at
// reference src/source_generic.cpp:53-136) is processed by T threads in phases; a
// phase boundary is a point where threads exchange data through LDS:
//
//   P1  fetch the latest N samples of the channel ring (reference :55-59), multiply by
//       the window (:97-103), radix-R1 butterflies, twiddle            -> write ex1
//   P2  read ex1, radix-R2 butterflies, twiddle                         -> write ex2
//   P3  read ex2, radix-R3 butterflies                                  -> write ex3 (Z, four planes by k mod 4)
//   P4  read Z[k], Z[M-k]; real split; |X|*2/sum(w) (:110-119); slope (:121-122);
//       temporal smoothing incl. fast peaks (:124-132); dBFS (:144-159, src/source.hpp:293-299);
//       volume normalisation (:161-167); roll-off (:169-179)            -> HBM
//   then, for configurations that display bars or a curve, the render-time reduction of the row just produced
//   (render_bars / render_curve, src/source.cpp:1360-1425, 1500-1564): bars_reduce_row / curve_row, outputs_finish.
//
// Each P*_read / P*_write split below marks where a block barrier is needed when
// T > 64 (several wavefronts share the spectrum); with T == 64 program order inside
// the wavefront is enough.
#pragma once
#include "wf_fft_core.hpp"

namespace wf {

enum : uint32_t {
    WF_MODE_TSMOOTH = 1u << 0,      // m_tsmoothing != NONE
    WF_MODE_FAST_PEAKS = 1u << 1,   // m_fast_peaks
    WF_MODE_STEREO = 1u << 2,       // m_stereo
    WF_MODE_NORMALIZE = 1u << 3,    // m_normalize_volume
    WF_MODE_ROLLOFF = 1u << 4,      // m_rolloff_q > 0 && m_rolloff_rate > 0
    WF_MODE_SLOPE = 1u << 5,        // m_slope > 0
    WF_MODE_WINDOW = 1u << 6,       // m_window_func != NONE
    WF_MODE_MONO_MIX = 1u << 7,     // !m_stereo && m_capture_channels > 1
};

// per-stream state bits kept in HBM between ticks
enum : uint32_t {
    WF_STREAM_LAST_SILENT = 1u << 0, // m_last_silent
    WF_STREAM_HIDDEN = 1u << 1,      // !m_show or capture timed out (host sets it)
    WF_STREAM_TIMEOUT = 1u << 2,     // set with HIDDEN when the cause is the capture timeout: tick_meter treats the two differently
    WF_STREAM_PAUSED = 1u << 4,      // the host did not tick this source in this video frame (WF_HIP_PAUSED): the stream is left exactly as it is
    WF_STREAM_STARVED = 1u << 5,     // the host found fewer samples than window + A/V-sync delay in the source's own buffer (WF_HIP_STARVED):
                                     // the tick takes the kernel's underflow path (no channel processed, the end-of-tick dB pass still runs)
    WF_STREAM_WRAPPED = 1u << 3,     // wpos has wrapped past 2^32 since the last reset: "fewer samples than the A/V-sync delay yet" is over for good
};

// bars or curve (render_bars / render_curve interpolation, filter, dB -> pixel mapping); out == nullptr: neither
struct BarArgs {
    // Per bar b the reference computes (1/count_b) * sum over the band's samples k of sum_t dB[ix_k - r + 1 + t] * W[k][t]
    // (src/filter.hpp:194-211; POINT mode: plain band mean, src/source.cpp:1525-1532).  The samples of a band sit on
    // consecutive bins, so the double sum collapses into ONE dot product per bar over a contiguous bin range, with
    // coefficient = the sum of all tap weights that land on that bin (built on the host in double from the reference's
    // own weight table, taps outside [0, M) dropped as kernel_convolve does): 8x fewer multiplies, 7x smaller table.
    // Flattened over all bars: entry e multiplies dB[bin[e]] by coef[e]; bar b owns entries [off[b], off[b+1]).
    const float *coef;         // [entries]
    const int *bin;            // [entries]
    const int *off;            // [num_bars + 1]
    const int *count;          // [num_bars] m_band_widths
    const int *chunk;          // [num_chunks + 1] bar ranges whose entries fit the LDS scratch together
    // big_outputs_kernel (wf_big.hpp): the entries cut into tasks of at most 2048, one wavefront each -- a bar at the top of a log
    // axis owns half the row, and one wavefront walking it alone was the kernel's whole duration
    const int *big_task;       // [num_tasks][4] bar, first entry, one past the last entry, the first entry's bin where the entries walk consecutive bins (else -1)
    const int *big_bar_task;   // [num_bars + 1] the tasks of bar b: [big_bar_task[b], big_bar_task[b + 1])
    int big_num_tasks;
    // Usual case (bars <= threads per spectrum): every thread owns one segment of the entries -- near-equal lengths,
    // never straddling a bar (the bars' own lengths differ by two orders of magnitude on a log axis) -- with its
    // coefficients laid lane-major: block c of thread s at [(c*T + s)*4, +4), zero-padded, for 4 * lane_blocks consecutive
    // bins from lane_base[s].  Bar b owns
    // segments [bar_seg[b], bar_seg[b+1]).  num_segs == 0: not built, the flat tables above are used chunk by chunk.
    const float *lane_coef;    // [lane_blocks][T][4]
    const int *lane_base;      // [T] first of the segment's 4 * lane_blocks consecutive bins (a multiple of 4: 16-byte LDS reads)
    const int *bar_seg;        // [num_bars + 1]
    const int *seg_group;      // [T] > 0: this segment starts a group of that many (<= 8) consecutive segments of one bar
    // wave-local layout (T > 64, no filter): every bar's segments lie inside one wavefront, so the partials are added with
    // wave-level ordering only -- no workgroup barrier between the dot products and the stores
    const int *lead_bar;       // [T] the bar whose first segment this thread owns, or -1
    const int *lead_end;       // [T] one past that bar's last segment
    int wave_local;
    // wave-private layout (BarPieceTables, wf_host_tables.hpp): seg_group = its `info` words, bar_seg = its `bar_piece`; the
    // segments of a thread read bins its own wavefront parked -- no barrier between parking the row and reading it --, pieces
    // are summed by a DPP prefix scan and the last wavefront to arrive adds the pieces of every bar
    int piece_mode;
    // prefix-sum layout (BarPsTables, wf_host_tables.hpp): every wavefront leaves its part of the row and its four-bin group sums;
    // the first wavefront of the spectrum forms a float64 prefix sum over 16-bin quads and evaluates the sub-bands, two lanes each
    // (one look-up and one 7-tap edge window per lane).  ps_tab: [3][64][4]; ps_lanes: 64, or 0: off
    const float *ps_tab;
    int ps_lanes;
    int num_segs;
    int lane_blocks;
    // Curve display (render_curve, reference src/source.cpp:1360-1425): num_bars = m_width points per row, point
    // o = k*T + s is the dot product of 8 consecutive dB bins starting at cur_base[o] with cur_coef[o][0..8) (the
    // Lanczos / Catmull-Rom taps of that point, clipped to [0, M) as kernel_convolve does; POINT mode: one tap).
    const float *cur_coef;     // [out_steps][T][8]
    const int *cur_base;       // [out_steps][T]
    // Catmull-Rom curve (the plugin's default interpolation, reference src/source.cpp:141): the four weights of a point are
    // a cubic in u = x - floor(x) (make_catrom_kernel, src/filter.hpp:67-104), so the table is the point's position alone --
    // 4 bytes per point instead of 36 -- and the weights are evaluated on the fly in the reference's own operation order.
    const float *cur_x;        // [out_steps rounded up to 4][T] m_interp_indices, lane-major; padding entries hold 1.0f
    int curve;                 // 2: Catmull-Rom curve from cur_x; 1: curve tables above; 0: bar tables
    int out_steps;             // ceil(num_bars / T) when the outputs are finished one per thread and step, else 0
    int both_subs;             // != 0 (curve, mono mixdown, two spectra per workgroup): the threads of both spectra finish the
                               // one displayed row; the tables are laid out for 2 * T threads
    int stream_steps;          // != 0: more steps than a thread's registers hold (wide curves): points are finished as they
                               // are produced -- mapped and stored at once, or staged behind the dB row for the filter
    // Gaussian filter across the outputs before the dB -> pixel mapping (apply_filter / weighted_avg,
    // reference src/filter.hpp:133-157,171-180; kernel make_gauss_kernel :40-65); gauss_radius == 0: off
    const float *gauss;        // [2 * gauss_radius - 1]
    const float *gauss_wsum;   // [num_bars] the sum of the weights whose taps fall inside the row (weighted_avg's divisor)
    int gauss_radius;
    int stage_off;             // chunked bars + filter: float offset (from the dB row in LDS) of the staging area [pad | bars | pad | weights]
#ifdef WF_PHASE_TIMING
    unsigned long long *clk;   // development aid: this workgroup's stamp slots
#endif
    float *out;                // [n_streams][disp_ch][num_bars]
    // mirrored frequency axis (reference src/source.cpp:1559-1564): outputs above the middle are replaced by images of the lower ones
    // AFTER render_bars / render_curve have taken the row's smallest y for the shader (miny / minpos, :1548-1557).  Above the middle
    // every output sits on the clamped top position and has the same value: output num_bars / 2 + 1's goes here, one float per
    // displayed row, so that the host finds the reference's miny without drawing the row itself.  nullptr: not kept
    float *pre_out;            // [n_streams][disp_ch]; inside the kernel's display phase: the entry of the row being finished
    // > 0 (wf_hip_set_bars_mirror / _mirrors): every tick also leaves the batch's bars -- the ones it finishes and, copied over, the
    // ones it does not touch (paused, hidden or silent streams) -- in out2_n more buffers of the same shape, buffer j starting
    // out2_delta[j] floats behind `out`: the send buffer of the all-gather of BASELINE configs[4] (or, with peer access, this shard's
    // slice of every device's gathered result) written by the kernel instead of by copies behind it
    int out2_n;
    long long out2_delta[8];
    int num_bars;
    int num_chunks;
    int entries;               // total number of entries (= off[num_bars])
    int lanes_per_bar;         // power of two <= 64: threads that share one bar in the segmented sum
    int mirror;
    float border_top, border_bottom;
    float ceiling, dbrange;    // m_ceiling, m_ceiling - m_floor
    float inv_dbrange;         // 1 / dbrange
    int lerp_mixed;            // border_top and border_bottom have opposite signs or one is 0: std::lerp's first form
    uint32_t disp_ch;
};

#ifdef WF_PHASE_TIMING
#define WF_BAR_STAMP(i) do { if(t == 0 && b.clk) b.clk[i] = __builtin_readcyclecounter(); } while(0)
#else
#define WF_BAR_STAMP(i)
#endif

// State of a handle that runs bars-only ticks, in device memory.  From the first such tick on every wavefront leaves "my
// slice of the row I produced has a value > floor - 10" in row_verdict[spec * (T/64) + wave] and the silence test reads those
// words instead of the rows (use_verdict; 0 on the first such tick, whose rows are still current).  Split mode has its own
// rotating verdict words (TickArgs::verdict_*) and leaves row_verdict null.  stale_row: M values of DB_MIN that stand in for
// the stale row of a skipped channel (wf_kernels.hpp).
struct BarsOnlyState {
    uint32_t *row_verdict;
    const float *stale_row;
    uint32_t use_verdict;
};

// FFT sizes with no prime factor above 5 (wf_mixed.hpp): the passes of the direct n/2-point transform
constexpr int MR_MAX_PASSES = 4;
struct MrPlan {
    int passes;                 // 0: no plan (Bluestein runs)
    int radix[MR_MAX_PASSES];
    int tw_off[MR_MAX_PASSES];  // pass s >= 1: its twiddles start at tw + tw_off[s]
    const cf *wp;               // radix[0] a prime above 25 (mr_pass_prime): [radix[0]] W_p^m in device memory (the tick kernel finds it in
                                // its LDS twiddle area, staged with the other tables; the large-FFT rows kernel copies it there itself)
    const cf *tw;               // per pass [R][Ns]: W_(Ns R)^(k jm), Ns = product of the radices before it (coalesced across a wavefront's butterflies)
    // The spectrum's exchange buffer, sized by the TRANSFORM rather than by the container geometry (wf::mr_exchange_cf): the passes
    // work between its two halves of `half` >= n/2 points, the finished Z[k] sits at (k & 3) * s3 + (k >> 2) -- ex3_addr's
    // four planes with a plane stride that fits n/2 points -- and spectrum s of a workgroup starts at s * lds_cf.  N = 800 on
    // the 2048-sample container: 6.5 instead of 8.7 KB per spectrum.
    int half, s3, lds_cf;
};
WF_DEV int mr_z_addr(const MrPlan &p, int k) { return (k & 3) * p.s3 + (k >> 2); }

struct TickArgs {
    // audio rings: one per (stream, captured channel), ring_cap samples each (power of two)
    const float *ring;
    const uint32_t *wpos;      // [n_streams] samples written so far, modulo 2^32
    uint32_t ring_mask;        // ring_cap - 1
    uint32_t ring_cap;
    uint32_t ring_stride;      // floats between the rings of consecutive (stream, channel) rows (>= ring_cap)
    uint32_t delay;            // frames between the end of the window and wpos (A/V sync, reference :50-51)
    const uint32_t *delay_stream; // [n_streams] added to `delay` per stream (wf_hip_set_stream_delay), or nullptr
    // per-configuration tables (read-only, shared by every stream)
    const float *window;       // [N]; all ones when FFTWindow::NONE
    const cf *tw1;             // [R1][M/R1]   W_M^(n' k1)
    const cf *tw2;             // [R2][R3]     W_(R2 R3)^(n3 k2)
    const cf *tws;             // [M]          W_N^k
    const float *slope;        // [M]; all ones when m_slope <= 0
    const float *rolloff;      // [M]
    // per-spectrum state and outputs
    float *tsmooth;            // [n_streams * cap_ch][M]   m_tsmooth_buf
    float *decibels;           // [n_streams][out_ch][M]    m_decibels
    uint32_t *stream_flags;    // [n_streams] read at the start of the tick; written at its end unless flags_out is set
    // split mode (the channels of a stereo stream in different workgroups, see spectrum_tick_kernel<.., SPLIT>): the words a
    // workgroup writes must not be the ones its partner reads in the same tick, so they rotate through three buffers
    uint32_t *flags_out;       // [n_streams] the flags the next tick reads
    const uint32_t *verdict_in; // [n_streams * cap_ch] != 0: the row left by the previous tick has a value > floor - 10
    uint32_t *verdict_out;     // the same for the rows as this tick leaves them (zero on entry; waves OR into it)
    uint32_t *verdict_clear;   // the buffer the next tick ORs into
    // scalars
    float half_coef;           // 0.5f * (2.0f / m_window_sum)
    float slope_step;          // 3 * m_slope / (M - 1): the slope factor of bin k is 1 + k * slope_step (0: slope off), Policy<G>::SLOPE_LINEAR
    float g, g2;               // get_gravity(seconds), 1 - g
    float vol_comp;            // min(m_volume_target - dbfs(m_input_rms), m_max_gain)
    const float *vol_comp_stream; // [n_streams] the same per stream (wf_hip_set_input_rms), or nullptr: vol_comp for all
    float db_min;              // DB_MIN
    float silent_floor;        // (float)(m_floor - 10)
    uint32_t n_streams;
    uint32_t stream_base, stream_count; // the slice of the batch this launch runs (wf_hip_tick issues the batch as lanes)
    uint32_t cap_ch;           // m_capture_channels (1 or 2)
    uint32_t out_ch;           // m_output_channels
    uint32_t mode;
    uint32_t skip_decibels;    // WF_HIP_TICK_NO_DECIBELS: bars-only batch mode
    // Mono mixdown on a geometry that holds one spectrum per workgroup (N = 32768 and its Bluestein sizes): the tick is two
    // launches of the split kernel -- channel 1 of every stream first (its smoothed magnitudes go to m_decibels[1], where
    // the reference keeps them too, src/source_generic.cpp:134), then channel 0, which mixes them in.  split_ch = the
    // channel this launch runs; 0xffffffff: all channels in one launch.
    uint32_t split_ch;
    // Once a tick of a handle has skipped the row store (WF_HIP_TICK_NO_DECIBELS), m_decibels in HBM is no longer what the
    // silence state machine must inspect (reference :78-86): see BarsOnlyState.  nullptr: rows are always stored (the usual
    // case; one pointer here instead of its fields keeps the kernel's scalar registers free).
    const BarsOnlyState *bars_only;
    const float *stale_row;   // BarsOnlyState::stale_row again, as a kernel argument: a pointer read from memory is a generic one,
                              // and a FLAT load anywhere on a path makes every later wait a wait for everything (see wf_kernels.hpp)
    // FFT sizes that are not powers of two (Bluestein, spectrum_tick_kernel<.., BLU>): the geometry's M is the padded
    // convolution length L, the transform the host asked for has blu_n points and row_bins = blu_n / 2 output bins
    const cf *blu_q;           // [blu_n / 2] conj(w_k) / L: Z_k = blu_q[k] * conj(R_k) for the twice-transformed R
    const cf *blu_qr;          // [blu_n / 2] blu_q[(blu_n / 2 - k) mod (blu_n / 2)]
    const cf *blu_w;           // [blu_n / 2] W_blu_n^k, the real-split twiddles
    MrPlan mr;                 // blu_n = 2^a 3^b 5^c: the transform is computed directly (blu_a, blu_b, blu_q, blu_qr unused)
    uint32_t big_c, big_r;     // big_mr_rows_kernel (wf_big.hpp): the n/2 points as big_c rows of big_r (mr plans a row)
    uint32_t big_rs;           // big_br_*_kernel: the stride of a row in the scratch buffer (big_r rounded up to even)
    const cf *big_wc;          // [8][8] W_big_c^(c k1): the column step of row k1
    const cf *blu_a;           // [2 M] the factors of x_2j and x_2j+1 in point j of the chirped, windowed, packed input; zero from point blu_n / 2 on
    const cf *blu_b;           // [M] FFT_M of the chirp
    uint32_t blu_n;
    uint32_t row_bins;         // bins per output / state row (M >> DEC for the power-of-two paths)
    // transforms beyond a CU's LDS (wf_big.hpp): the finished transform in device memory and what the epilogue needs with it
    const cf *big_z;           // [n_spec][big_l] rows' output, natural order
    const cf *big_tws;         // [big_m] W_(2 big_m)^k (real split of the 65536-sample transform)
    const cf *big_tw;          // [L1][16384] W_L^(n2 k1): the column twiddles (big_whole_kernel / big_mr_rows_kernel fold the column step into their fetch)
    uint32_t *big_nz_out;      // big_nz, writable (row 0 of big_mr_rows_kernel ORs "the window has a non-zero sample" into it)
    const uint32_t *big_nz;    // [n_spec] != 0: the window has a non-zero sample
    uint32_t big_m, big_l;     // complex points of the packed real transform; complex points per transform (scratch stride)
    BarArgs bar;
    unsigned long long *phase_clock; // development aid (builds with -DWF_PHASE_TIMING): s_memtime stamps per workgroup
};

// ---- small helpers ------------------------------------------------------------------------
WF_DEV f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
WF_DEV f2 ld2(const float *p) { return *reinterpret_cast<const f2 *>(p); }
WF_DEV void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
// Streaming accesses.  Measured on MI355X (cfg3, interleaved A/B on one box): the m_decibels rows -- output nothing on the
// device reads again -- stored with the non-temporal hint +1.8 % (they stop displacing the rings from the 256 MB Infinity
// Cache).  The smoothing state is read once and written once per tick: the hint on its stores alone +-0, on its loads
// alone +-0, on both +3.7 % (0.701 -> 0.727; 16384 streams 0.69 -> 0.72-0.75) -- then only the rings, whose consecutive
// windows overlap by 80 %, compete for the cache.  On the window loads themselves the hint costs 2 % (WF_NT_SMP).
#ifndef WF_SLOPE_LINEAR
#define WF_SLOPE_LINEAR 1 // Policy<G>::SLOPE_LINEAR
#endif
#ifndef WF_BAR_COEF_EARLY
#define WF_BAR_COEF_EARLY 1 // Policy<G>::BAR_COEF_EARLY
#endif
#ifndef WF_NT_ROWS
#define WF_NT_ROWS true // m_decibels rows stored with the non-temporal hint
#endif
#ifndef WF_NT_STATE
#define WF_NT_STATE 1
#endif
#ifndef WF_NT_SMP
#define WF_NT_SMP 0
#endif
#ifndef WF_NT_STATE_LD
#define WF_NT_STATE_LD 1
#endif
#if defined(__HIPCC__)
WF_DEV f4 ld4_nt(const float *p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return f4{v.x, v.y, v.z, v.w};
}
WF_DEV void st4_nt(float *p, f4 v)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(p));
}
#else // the g++ wavefront emulator (tests/emu): plain accesses
WF_DEV f4 ld4_nt(const float *p) { return ld4(p); }
WF_DEV void st4_nt(float *p, f4 v) { st4(p, v); }
#endif

// m_tsmooth_buf is read once and written once per tick: both with the hint (WF_NT_STATE / WF_NT_STATE_LD)
WF_DEV f4 ld_state(const float *p) { return WF_NT_STATE_LD ? ld4_nt(p) : ld4(p); }
WF_DEV void st_state(float *p, f4 v)
{
    if(WF_NT_STATE)
        st4_nt(p, v);
    else
        st4(p, v);
}

// Products and sums that must be rounded on their own, as the reference's scalar code rounds them.  HIP compiles device code
// with -ffp-contract=fast, and __fmul_rn / __fadd_rn are plain operators to the optimiser: `__fadd_rn(__fmul_rn(a, b), c)`
// becomes one fma.  Harmless within a tolerance -- except where the sum cancels: the level meter's smoothing starts from
// m_meter_buf = DB_MIN (a quirk of the reference: -758.6 taken as a linear level), and on the tick where g * old + g2 * new
// first turns positive a fused product moved the level by 7e-4 dB (fuzz seed 14562 of an extended sweep).
WF_DEV float mul_unfused(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
WF_DEV float add_unfused(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
// (g * old) + (g2 * cur), three roundings (reference src/source_generic.cpp:255)
WF_DEV float meter_ema(float g, float old, float g2, float cur)
{
#pragma clang fp contract(off)
    const float x = g * old;
    const float y = g2 * cur;
    return x + y;
}

// The spectrum's temporal smoothing, mag = g * old + (1 - g) * mag (reference src/source_generic.cpp:124-132).  WF_EMA_GENERIC = 1: the
// generic class's three roundings (two products, one sum); 0: fma(g, old, g2 * mag), two roundings, as the reference's own AVX2 class
// contracts it (src/source_avx2.cpp:154)
#ifndef WF_EMA_GENERIC
#define WF_EMA_GENERIC 1
#endif
WF_DEV float spectrum_ema(float g, float old, float g2, float cur)
{
#if WF_EMA_GENERIC
    return meter_ema(g, old, g2, cur);
#else
    return fmaf(g, old, g2 * cur);
#endif
}

// ordering point between LDS operations of one wavefront (they execute in program order: a scheduling fence suffices)
#if defined(__HIPCC__)
WF_DEV void wait_vmem_all() { __builtin_amdgcn_s_waitcnt(0x0F70); } // vmcnt(0), the other counters untouched (gfx9 encoding)
WF_DEV void wave_fence() { __builtin_amdgcn_wave_barrier(); }
WF_DEV float wave_shfl_down(float v, int d) { return __shfl_down(v, d, 64); } // lane l gets lane l + d's value (its own past the wavefront)
// Segmented inclusive prefix sum over the lanes of a wavefront: lane l ends up with the sum of the lanes of its segment up to
// and including itself.  `info` bits 0..5 say which of the six steps of a wave-wide scan this lane takes (BarPieceTables::info:
// the lanes 1, 2, 4, 8 below it inside its row of 16, the last lane of the row before, lane 31).  Six DPP moves at VALU speed
// instead of six ds_bpermute round trips; a step not taken adds an exact 0 (the moved value is masked, not multiplied: whatever
// an unused lane holds stays out).
WF_DEV float seg_prefix_scan(float v, uint32_t info)
{
#define WF_SCAN_STEP(CTRL, ROWS, BIT)                                                                                      \
    {                                                                                                                      \
        const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, true);                            \
        v += __int_as_float(o & __builtin_amdgcn_sbfe((int)info, BIT, 1));                                                 \
    }
    WF_SCAN_STEP(0x111, 0xf, 0) // row_shr:1
    WF_SCAN_STEP(0x112, 0xf, 1) // row_shr:2
    WF_SCAN_STEP(0x114, 0xf, 2) // row_shr:4
    WF_SCAN_STEP(0x118, 0xf, 3) // row_shr:8
    WF_SCAN_STEP(0x142, 0xa, 4) // row_bcast:15 into rows 1 and 3
    WF_SCAN_STEP(0x143, 0xc, 5) // row_bcast:31 into rows 2 and 3
#undef WF_SCAN_STEP
    return v;
}
// The arrival counters of the display phase (a wavefront has made its last reads of the exchange buffer / has left its part of the
// row): release on the count, acquire on the wait, at workgroup scope -- what the memory model asks for.  (Until round 5: relaxed
// operations between compiler barriers, leaning on gfx9 executing a wavefront's LDS operations in order; on LDS the ordered forms cost
// one s_waitcnt lgkmcnt(0) in front of the count.  -DWF_ARRIVE_RELAXED=1 builds the old form for an A/B.)
#ifndef WF_ARRIVE_RELAXED
#define WF_ARRIVE_RELAXED 0
#endif
#define WF_ARRIVE_ORDER_REL (WF_ARRIVE_RELAXED ? __ATOMIC_RELAXED : __ATOMIC_RELEASE)
#define WF_ARRIVE_ORDER_ACQ (WF_ARRIVE_RELAXED ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE)
// a wavefront counts itself in (LDS atomic, workgroup scope); every lane gets the count before it
WF_DEV int wave_arrive(int *counter, int lane)
{
    int old = 0;
    asm volatile("" ::: "memory");
    if(lane == 0)
        old = __hip_atomic_fetch_add(counter, 1, WF_ARRIVE_RELAXED ? __ATOMIC_RELAXED : __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    asm volatile("" ::: "memory");
    return old;
}
#else
WF_DEV void wait_vmem_all() {}
WF_DEV void wave_fence() {}
WF_DEV float wave_shfl_down(float v, int) { return v; } // (the emulator does not run the bar reduction)
WF_DEV float seg_prefix_scan(float v, uint32_t) { return v; }
WF_DEV int wave_arrive(int *counter, int) { return __atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL); }
#endif

// All LDS traffic goes through these four helpers (ds_read_b64/b128, ds_write_b64/b128).
// WF_LDS_TRACE is defined only by the g++ wavefront emulator in tests/emu to record
// (address, width, direction) per access for the bank-conflict census; it is empty here.
#ifndef WF_LDS_TRACE
#define WF_LDS_TRACE(idx, bytes, is_write)
#endif
WF_DEV cf lds_ld2(const cf *lds, int idx) { WF_LDS_TRACE(idx, 8, 0); return lds[idx]; }
WF_DEV void lds_st2(cf *lds, int idx, cf a) { WF_LDS_TRACE(idx, 8, 1); lds[idx] = a; }
WF_DEV f4 lds_ld4(const cf *lds, int idx) { WF_LDS_TRACE(idx, 16, 0); return *reinterpret_cast<const f4 *>(lds + idx); }
WF_DEV void lds_st4(cf *lds, int idx, cf a, cf b) { WF_LDS_TRACE(idx, 16, 1); *reinterpret_cast<f4 *>(lds + idx) = f4{a.x, a.y, b.x, b.y}; }


WF_DEV uint32_t f32_bits(float v)
{
#if defined(__HIPCC__)
    return __float_as_uint(v);
#else
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    return u;
#endif
}

// dbfs(), reference src/source.hpp:293-299: mag > 0 ? 20*log10f(mag) : DB_MIN.
// Device form: max(log2(m) * 20*log10(2), DB_MIN) on the hardware log2 (v_log_f32, <= 1 ulp of its result; relative
// error of the dB value <= ~2e-7).  log2(0) = -inf and log2(negative) = NaN both end at DB_MIN through the max, as the
// reference's test does; for every normal m > 0 the max is a no-op because 20*log10(FLT_MIN) == DB_MIN.  One stated
// deviation: v_log_f32 flushes denormal inputs, so magnitudes below FLT_MIN (1.2e-38, i.e. under -758 dBFS) read
// DB_MIN where the reference would read -759..-897 dB -- far below any displayable floor (>= -120 dB).
#ifndef WF_FAST_DB
#define WF_FAST_DB 1
#endif
WF_DEV float dbfs(float mag, float db_min)
{
#if defined(__HIPCC__) && WF_FAST_DB
    return fmaxf(__builtin_amdgcn_logf(mag) * 6.02059991327962390f, db_min);
#else
    return (mag > 0.0f) ? 20.0f * log10f(mag) : db_min;
#endif
}

// |2X| from its parts.  hypotf in the reference (:119); here sqrt(fma) on the hardware square root (1 ulp).
#ifndef WF_FAST_SQRT
#define WF_FAST_SQRT 1
#endif
WF_DEV float mag2(float xr, float xi)
{
    const float s = fmaf(xi, xi, xr * xr);
#if defined(__HIPCC__) && WF_FAST_SQRT
#if defined(WF_MAG_SAFE)
    // squares of parts below ~1e-19 underflow (the first ticks behind a reset through a narrow window: a few samples under
    // sin^16 tails give |X| ~ 1e-26 where hypotf still answers): rare branch with the parts scaled by 2^64
    if(__builtin_expect(s < 0x1p-100f, 0)) {
        const float a = xr * 0x1p64f, b = xi * 0x1p64f;
        return __builtin_amdgcn_sqrtf(fmaf(b, b, a * a)) * 0x1p-64f;
    }
#endif
    return __builtin_amdgcn_sqrtf(s);
#else
    return sqrtf(s);
#endif
}

// ---- P1: fetch + window + pass 1 ---------------------------------------------------------------
// x      : base of this spectrum's ring
// start  : ring index of the first sample of the window
// ALIGNED: start % 4 == 0 (vector loads never straddle the ring wrap)
// p1_fetch copies this thread's share of the window into registers (the reference's peek_front into
// m_fft_input, :55-59) as smp[j][e] = x[start + 2*(j*M1 + B1*t) + e] and reports whether any of them is
// non-zero (the reference's silence scan, :63-72).
// Everything pass 1 and the real split need from HBM/L2, issued back to back so the latencies overlap:
//   smp   this thread's share of the window (the reference's peek_front into m_fft_input, :55-59):
//         smp[j][e] = x[start + 2*(j*M1 + B1*t) + e]
//   win   the window coefficients of the same samples (:97-103)
//   tw1   W_M^(n' k1) for this thread's n' (k1 = 1..R1-1)
//   wb    W_N^(4t+i), i = 0..3: the real-split twiddles of this thread's first bin group; the other groups
//         are wb * W_N^(4Tu) = wb * W_32^(u*64/P) (exact multiples of 1/32 turn)
// Prefetch policy.  Every thread owns <= 16 points, holds the window / twiddle operands from the start and prefetches the
// smoothing state while passes 2-3 run, in one of two ways (WF_PREFETCH_STATE):
//   1 = into registers (16 VGPRs at P = 16), together with the slope table;
//   2 = "touch": one dword per 64-byte line of the state row (and of the slope table) is requested right after pass 1 so
//       that P4's real loads are served from L2; costs 2 VGPRs instead of 32, which keeps the multi-wavefront
//       geometries at 4 waves per SIMD;  0 = load in P4.
//  -1 (default) = the measured choice per geometry on MI355X (interleaved A/B, same box): registers for the one-wavefront
//       geometries (N = 1024: 68 vs 75 us, N = 2048: 72.3 vs 75 us), touch for N >= 4096 (77.1-79.1 vs 79.3-80.4 us then;
//       122 instead of 156 VGPRs).
#ifndef WF_PREFETCH_STATE
#define WF_PREFETCH_STATE -1
#endif
template<class G> struct Policy {
    static_assert(G::P <= 32, "the phase functions keep a thread's operands in registers: at most 32 points per thread (two waves per SIMD)");
    // 1: state + slope into registers right after pass 1; 2: their lines only "touched" there (into L2), loaded in P4;
    // 0: everything requested at the start of P4, consumed after the real split.  Round 1 found 1 / 2 best (one wavefront /
    // several); with the lanes and the non-temporal row stores of round 2 the early requests only lengthen the fetch
    // burst's queue: mode 0 is +3 % at N = 4096, +1.7 % at 2048, +1..3 % at 8192 / 16384, +-0 at 32768 (interleaved A/B).
    // The 8-point geometry keeps 1 (its decimated epilogue takes the operands from registers).
    static constexpr int MODE = (WF_PREFETCH_STATE >= 0) ? WF_PREFETCH_STATE : (G::P <= 8 ? 1 : 0);
    static constexpr bool PREFETCH_STATE = (MODE == 1 || MODE == 3);
    static constexpr bool PREFETCH_SLOPE = (MODE == 1 || MODE == 3);
    static constexpr bool TOUCH_STATE = (MODE == 2);
    static constexpr bool PREFETCH_LATE = (MODE == 3); // 3: as 1, but requested behind pass 3's stores, under the barrier in front of P4
    // The slope factors m_slope_modifiers[k] = log10f(10 * powf(1000, k * slope / (M - 1))) (src/source.cpp:1283-1290) are, but for
    // the roundings of powf and log10f, 1 + 3 k slope / (M - 1): the geometries that request all of P4's operands first (MODE 0)
    // form them with one fma per bin instead of loading them -- four of the twelve requests of P4's burst and 16 registers across
    // the real split gone (N = 4096: 0.791 -> 0.803 of the HBM peak, interleaved, profiles/r04a_coef_early.txt); the factors
    // differ from the reference's table by at most 4e-7 relative (tests/test_cpu_units.py pins it), DESIGN.md section 5.
    static constexpr bool SLOPE_LINEAR = (WF_SLOPE_LINEAR != 0) && MODE == 0;
    // The display's per-thread tables (BarEntries) requested in front of P4 instead of behind its state stores: a load at the end
    // of the kernel waits out the whole loaded memory pipeline once more -- that wait, not the barrier or the shuffles, was the
    // bars tail (profiles/r04a_tail_cuts.txt).  Needs the registers SLOPE_LINEAR frees (without them: 128 VGPRs + 36 B of scratch).
    static constexpr bool BAR_COEF_EARLY = (WF_BAR_COEF_EARLY != 0) && SLOPE_LINEAR && G::T >= 128;
};
// Pass-1 twiddle rows W_M^(n' k1), k1 = 1..R1-1: only the rows whose k1 is a power of two come from the table; the others
// are products of two of those (or of one and an earlier product): k1 = hi + lo with lo the lowest set bit.  At R1 = 8 that
// is 3 loaded rows and 4 complex multiplications per point instead of 7 loaded rows: 16 fewer VGPRs held through the fetch
// burst and 4 of its 23 sixteen-byte requests gone.  Accuracy: inputs are correctly rounded table entries (1/2 ulp each), a
// product adds one rounding: the worst row (k1 = 7 = (4 + 2) + 1) carries ~3.5 rounding units (2e-7 relative) instead of 1/2
// -- the size of the FFT's own round-off, an order of magnitude under the parity tolerance.  (Round 1 tried powers of the
// single row k1 = 1: up to 8e-7, rejected then.)  WF_TW1_POW2=0 loads every row.
#ifndef WF_TW1_POW2
#define WF_TW1_POW2 1
#endif
constexpr bool tw1_row_loaded(int k1) { return !WF_TW1_POW2 || (k1 & (k1 - 1)) == 0; }

template<class G> struct P1Regs {
    float smp[G::R1][2 * G::B1];
    float win[G::R1][2 * G::B1];
    float tw1[G::R1][2 * G::B1];
    cf wb[4];
};

// window coefficients / pass-1 twiddles of row j of this thread
template<class G> WF_DEV void p1_load_window(const TickArgs &a, int t, int j, float (&win)[2 * G::B1])
{
    constexpr int B1 = G::B1, M1 = G::M1;
    const uint32_t s0 = 2u * (uint32_t)(j * M1 + B1 * t);
    if(B1 == 2) {
        const f4 w = ld4(a.window + s0);
        win[0] = w.x; win[1] = w.y; win[2 * B1 - 2] = w.z; win[2 * B1 - 1] = w.w;
    } else {
        const f2 w = ld2(a.window + s0);
        win[0] = w.x; win[1] = w.y;
    }
}
template<class G> WF_DEV void p1_load_tw1(const TickArgs &a, int t, int k1, float (&tw1)[2 * G::B1])
{
    constexpr int B1 = G::B1, M1 = G::M1;
    const float *tw = reinterpret_cast<const float *>(a.tw1 + k1 * M1 + B1 * t);
    if(B1 == 2) {
        const f4 w = ld4(tw);
        tw1[0] = w.x; tw1[1] = w.y; tw1[2 * B1 - 2] = w.z; tw1[2 * B1 - 1] = w.w;
    } else {
        const f2 w = ld2(tw);
        tw1[0] = w.x; tw1[1] = w.y;
    }
}

// returns whether any sample of this thread is non-zero (the reference's silence scan, :63-72)
// DEC > 0 (FFT sizes below the smallest geometry): the window is the first N >> DEC samples of a zero-padded N-point
// transform -- rows j >= R1 >> DEC of every thread are zeros and are not loaded -- and the real-split twiddles are those of
// the bins the small transform keeps, k = (4t + i) << DEC.
// TLDS: the window and pass-1 twiddle tables are not loaded here -- the workgroup stages them in LDS (p1_tables_to_lds).
template<class G, bool ALIGNED, int DEC = 0, bool TLDS = false>
WF_DEV bool p1_fetch(const TickArgs &a, int t, const float *x, uint32_t start, P1Regs<G> &r)
{
    constexpr int R1 = G::R1, B1 = G::B1, M1 = G::M1;
    constexpr int RV = R1 >> DEC; // rows that hold samples
    static_assert(RV >= 1, "zero-padding leaves at least one row of samples per thread");
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        const uint32_t s0 = 2u * (uint32_t)(j * M1 + B1 * t);
        if(j >= RV) {
            WF_UNROLL
            for(int e = 0; e < 2 * B1; ++e) {
                r.smp[j][e] = 0.0f;
                r.win[j][e] = 0.0f;
            }
        } else if(ALIGNED) {
            if(B1 == 2) {
                const f4 q = WF_NT_SMP ? ld4_nt(x + ((start + s0) & a.ring_mask)) : ld4(x + ((start + s0) & a.ring_mask));
                r.smp[j][0] = q.x; r.smp[j][1] = q.y; r.smp[j][2 * B1 - 2] = q.z; r.smp[j][2 * B1 - 1] = q.w;
            } else {
                const f2 q = ld2(x + ((start + s0) & a.ring_mask));
                r.smp[j][0] = q.x; r.smp[j][1] = q.y;
            }
        } else {
            WF_UNROLL
            for(int e = 0; e < 2 * B1; ++e)
                r.smp[j][e] = *(x + ((start + s0 + (uint32_t)e) & a.ring_mask));
        }
    }
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        if(j < RV && !TLDS)
            p1_load_window<G>(a, t, j, r.win[j]);
        if(j >= 1 && !TLDS && tw1_row_loaded(j))
            p1_load_tw1<G>(a, t, j, r.tw1[j]);
    }
    if(DEC == 0) {
        const f4 wa = ld4(reinterpret_cast<const float *>(a.tws + 4 * t));
        const f4 wc = ld4(reinterpret_cast<const float *>(a.tws + 4 * t + 2));
        r.wb[0] = cf{wa.x, wa.y}; r.wb[1] = cf{wa.z, wa.w}; r.wb[2] = cf{wc.x, wc.y}; r.wb[3] = cf{wc.z, wc.w};
    } else {
        WF_UNROLL
        for(int i = 0; i < 4; ++i) {
            const int k = ((4 * t + i) << DEC) & (G::M - 1); // threads beyond the kept bins read a valid entry they never use
            const f2 w = ld2(reinterpret_cast<const float *>(a.tws + k));
            r.wb[i] = cf{w.x, w.y};
        }
    }
    // x != 0.0f for any sample: OR the bit patterns, drop the sign bit (-0.0f == 0.0f); NaNs have non-zero bits
    uint32_t acc = 0;
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        WF_UNROLL
        for(int e = 0; e < 2 * B1; ++e)
            acc |= f32_bits(r.smp[j][e]);
    }
    return (acc & 0x7fffffffu) != 0;
}

// window multiply (reference :97-103; the table is all ones for FFTWindow::NONE: x * 1.0f == x), butterflies over n1 (= j),
// twiddle by W_M^(n' k1): o[k1][b] = A'[k1][n' + b]
template<class G>
WF_DEV void p1_window_dft(P1Regs<G> &r, cf (&o)[G::R1][G::B1])
{
    constexpr int R1 = G::R1, B1 = G::B1;
    cf u[B1][R1];
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        WF_UNROLL
        for(int e = 0; e < 2 * B1; ++e)
            r.smp[j][e] *= r.win[j][e];
        WF_UNROLL
        for(int b = 0; b < B1; ++b)
            u[b][j] = cf{r.smp[j][2 * b], r.smp[j][2 * b + 1]};
    }
    WF_UNROLL
    for(int b = 0; b < B1; ++b)
        dft_dif<R1>(u[b]);
    constexpr int LB = ilog2(R1);
    WF_UNROLL
    for(int k1 = 0; k1 < R1; ++k1) {
        if(k1 >= 1 && !tw1_row_loaded(k1)) { // row k1 = row (k1 minus its lowest set bit) times row (lowest set bit)
            const int lo = k1 & -k1, hi = k1 - lo;
            WF_UNROLL
            for(int b = 0; b < B1; ++b) {
                const cf w = cmul(cf{r.tw1[hi][2 * b], r.tw1[hi][2 * b + 1]}, cf{r.tw1[lo][2 * b], r.tw1[lo][2 * b + 1]});
                r.tw1[k1][2 * b] = w.x;
                r.tw1[k1][2 * b + 1] = w.y;
            }
        }
        WF_UNROLL
        for(int b = 0; b < B1; ++b)
            o[k1][b] = (k1 == 0) ? u[b][0] : cmul(u[b][brev(k1, LB)], cf{r.tw1[k1][2 * b], r.tw1[k1][2 * b + 1]});
    }
}
// store A'[k1][n'] into the exchange buffer
template<class G> WF_DEV void p1_store(int t, cf *lds, const cf (&o)[G::R1][G::B1])
{
    constexpr int R1 = G::R1, B1 = G::B1;
    const int np = B1 * t;
    WF_UNROLL
    for(int k1 = 0; k1 < R1; ++k1) {
        if(B1 == 2)
            lds_st4(lds, ex1_addr<G>(k1, np), o[k1][0], o[k1][B1 - 1]);
        else
            lds_st2(lds, ex1_addr<G>(k1, np), o[k1][0]);
    }
}
template<class G>
WF_DEV void p1_window_pass1(const TickArgs &a, int t, P1Regs<G> &r, cf *lds)
{
    (void)a;
    cf o[G::R1][G::B1];
    p1_window_dft<G>(r, o);
    p1_store<G>(t, lds, o);
}
// TLDS: the window (N floats) and the pass-1 twiddles (M complex) staged by the workgroup in LDS, in memory order:
// tab[0 .. M) = window as float pairs, tab[M .. 2M) = tw1.  This thread's operands, as p1_fetch would have loaded them.
template<class G> WF_DEV void p1_tables_from_lds(int t, const cf *tab, P1Regs<G> &r)
{
    constexpr int R1 = G::R1, B1 = G::B1, M1 = G::M1, M = G::M;
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        if(B1 == 2) {
            const f4 w = lds_ld4(tab, j * M1 + B1 * t);
            r.win[j][0] = w.x; r.win[j][1] = w.y; r.win[j][2 * B1 - 2] = w.z; r.win[j][2 * B1 - 1] = w.w;
            if(j >= 1) {
                const f4 q = lds_ld4(tab, M + j * M1 + B1 * t);
                r.tw1[j][0] = q.x; r.tw1[j][1] = q.y; r.tw1[j][2 * B1 - 2] = q.z; r.tw1[j][2 * B1 - 1] = q.w;
            }
        } else {
            const cf w = lds_ld2(tab, j * M1 + t);
            r.win[j][0] = w.x; r.win[j][1] = w.y;
            if(j >= 1) {
                const cf q = lds_ld2(tab, M + j * M1 + t);
                r.tw1[j][0] = q.x; r.tw1[j][1] = q.y;
            }
        }
    }
}

// Issued as soon as pass 1 has freed its registers so that the HBM latency of the smoothing state (and the
// L2 latency of the slope table) is covered by passes 2 and 3 instead of being paid in P4.
template<class G> struct P4Regs {
    float st[Policy<G>::PREFETCH_STATE ? G::P : 1]; // m_tsmooth_buf of this thread's bins
    float sl[Policy<G>::PREFETCH_SLOPE ? G::P : 1]; // m_slope_modifiers of this thread's bins
    float touch[2];                                 // TOUCH_STATE: the two requested dwords (kept only to be waited for)
};
template<class G> WF_DEV void p4_prefetch(const TickArgs &a, int t, const float *ts, P4Regs<G> &q)
{
    constexpr int T = G::T, P = G::P;
    if(Policy<G>::TOUCH_STATE) {
        // M floats per row = M/16 lines of 64 bytes; T threads cover them in ceil(M/16/T) = 1 step for every shipped geometry
        static_assert(G::M / 16 <= G::T || !Policy<G>::TOUCH_STATE, "one touch per thread covers the row");
        const int line = (t < G::M / 16) ? t : 0;
        q.touch[0] = (a.mode & WF_MODE_TSMOOTH) ? ts[16 * line] : 0.0f;
        q.touch[1] = a.slope[16 * line];
        return;
    }
    if(Policy<G>::PREFETCH_STATE && (a.mode & WF_MODE_TSMOOTH)) { // one scalar branch around all of the state loads
        constexpr int S = Policy<G>::PREFETCH_STATE ? 1 : 0;
        WF_UNROLL
        for(int u = 0; u < P / 4; ++u) {
            const f4 o = ld_state(ts + 4 * (t + T * u));
            q.st[S * (4 * u)] = o.x; q.st[S * (4 * u + 1)] = o.y; q.st[S * (4 * u + 2)] = o.z; q.st[S * (4 * u + 3)] = o.w;
        }
    }
    WF_UNROLL
    for(int u = 0; u < P / 4; ++u) {
        const int k0 = 4 * (t + T * u);
        if(Policy<G>::PREFETCH_SLOPE) {
            constexpr int S = Policy<G>::PREFETCH_SLOPE ? 1 : 0;
            const f4 o = ld4(a.slope + k0);
            q.sl[S * (4 * u)] = o.x; q.sl[S * (4 * u + 1)] = o.y; q.sl[S * (4 * u + 2)] = o.z; q.sl[S * (4 * u + 3)] = o.w;
        }
    }
}

// ---- Bluestein (FFT sizes that are not powers of two) ---------------------------------------------------------------------
// X_k = conj(w_k) sum_j (x_j conj(w_j)) w_(k-j), w_m = exp(i pi m^2 / n): a circular convolution of length M >= 3n/2 done
// with two M-point complex transforms of the same core.  The product a_j = x_j * (window_j conj(w_j)) is the first
// transform's input, already complex: it goes through p1_window_dft with a window of ones.
// The chirped window (two complex factors per point, 16 bytes -- twice the samples' own size) does not come in with the samples:
// six registers per point in flight were 96 and more for the burst, and the Bluestein kernels spilled (48-108 B per lane,
// rounds 1-2).  The workgroup stages the table in its exchange buffers by LDS-DMA instead (blu_table_to_lds: np * 16 bytes
// <= one buffer, free until pass 1 stores into it), p1_fetch_blu fetches the raw sample pairs only, and the products are
// formed behind the workgroup's first barrier from LDS (blu_products_from_lds) -- the structure of the TLDS path.
template<class G> WF_DEV bool p1_fetch_blu(const TickArgs &a, int t, const float *x, uint32_t start, P1Regs<G> &r)
{
    // packed form (wf_host_tables.hpp): point j of the transform is x[2j] * blu_a[2j] + x[2j+1] * blu_a[2j+1], j < blu_n / 2;
    // here: the pair (x[2j], x[2j+1]) of every point of this thread, zero beyond the window
    constexpr int R1 = G::R1, B1 = G::B1, M1 = G::M1;
    const uint32_t np = a.blu_n >> 1;
    uint32_t acc = 0;
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        WF_UNROLL
        for(int b = 0; b < B1; ++b) {
            const uint32_t idx = (uint32_t)(j * M1 + B1 * t + b);
            const bool in = idx < np;
            const uint32_t s = start + 2u * (in ? idx : 0u);
            const float v0 = x[s & a.ring_mask], v1 = x[(s + 1u) & a.ring_mask];
            const float x0 = in ? v0 : 0.0f, x1 = in ? v1 : 0.0f;
            acc |= f32_bits(x0) | f32_bits(x1);
            r.smp[j][2 * b] = x0;
            r.smp[j][2 * b + 1] = x1;
        }
        if(j >= 1 && tw1_row_loaded(j))
            p1_load_tw1<G>(a, t, j, r.tw1[j]);
    }
    return (acc & 0x7fffffffu) != 0;
}
// the one- and two-round-trip form for the smallest geometries (eight points per thread and fewer: 48 registers in flight fit,
// and the staging barrier costs more than it saves there: N = 496 0.283 -> 0.277 with the table through LDS, measured):
// table entries fetched with the samples, products formed on the spot
template<class G> WF_DEV bool p1_fetch_blu_direct(const TickArgs &a, int t, const float *x, uint32_t start, P1Regs<G> &r)
{
    constexpr int R1 = G::R1, B1 = G::B1, M1 = G::M1;
    const uint32_t np = a.blu_n >> 1;
    uint32_t acc = 0;
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        WF_UNROLL
        for(int b = 0; b < B1; ++b) {
            const uint32_t idx = (uint32_t)(j * M1 + B1 * t + b);
            const bool in = idx < np;
            const uint32_t s = start + 2u * (in ? idx : 0u);
            const float v0 = x[s & a.ring_mask], v1 = x[(s + 1u) & a.ring_mask];
            const float x0 = in ? v0 : 0.0f, x1 = in ? v1 : 0.0f;
            acc |= f32_bits(x0) | f32_bits(x1);
            const f4 q = ld4(reinterpret_cast<const float *>(a.blu_a + 2u * idx));
            r.smp[j][2 * b] = fmaf(x1, q.z, x0 * q.x);
            r.smp[j][2 * b + 1] = fmaf(x1, q.w, x0 * q.y);
        }
        WF_UNROLL
        for(int e = 0; e < 2 * B1; ++e)
            r.win[j][e] = 1.0f;
        if(j >= 1 && tw1_row_loaded(j))
            p1_load_tw1<G>(a, t, j, r.tw1[j]);
    }
    return (acc & 0x7fffffffu) != 0;
}
template<class G> constexpr bool blu_table_via_lds() { return G::P > 8; }
#if defined(__HIPCC__)
// blu_a[0 .. 2 np) -> the start of the workgroup's LDS, 1 KB per wave-wide request, the waves sharing the requests
WF_DEV void blu_table_to_lds(const TickArgs &a, void *lds_dst, int wave, int n_waves, int lane)
{
    const uint32_t bytes = (a.blu_n >> 1) * 16u;
    for(uint32_t c = (uint32_t)wave; c * 1024u < bytes; c += (uint32_t)n_waves) { // wave-uniform
        const uint32_t off = c * 1024u + (uint32_t)lane * 16u;
        if(off < bytes) {
            const char *g = reinterpret_cast<const char *>(a.blu_a) + off;
            char *l = static_cast<char *>(lds_dst) + c * 1024u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
        }
    }
}
#else
WF_DEV void blu_table_to_lds(const TickArgs &a, void *lds_dst, int wave, int, int lane)
{
    if(wave == 0 && lane == 0)
        for(uint32_t i = 0; i < (a.blu_n >> 1) * 2u; ++i)
            static_cast<cf *>(lds_dst)[i] = a.blu_a[i];
}
#endif
// a_j = x[2j] T1_j + x[2j+1] T2_j with (T1_j, T2_j) = tab[2j], tab[2j+1] from LDS; r.smp holds the sample pairs on entry
template<class G> WF_DEV void blu_products_from_lds(const TickArgs &a, int t, const cf *tab, P1Regs<G> &r)
{
    constexpr int R1 = G::R1, B1 = G::B1, M1 = G::M1;
    const uint32_t np = a.blu_n >> 1;
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        WF_UNROLL
        for(int b = 0; b < B1; ++b) {
            const uint32_t idx = (uint32_t)(j * M1 + B1 * t + b);
            const f4 q = lds_ld4(tab, (int)(2u * (idx < np ? idx : 0u))); // (beyond the window the samples are zero: any finite entry does)
            const float x0 = r.smp[j][2 * b], x1 = r.smp[j][2 * b + 1];
            r.smp[j][2 * b] = fmaf(x1, q.z, x0 * q.x);
            r.smp[j][2 * b + 1] = fmaf(x1, q.w, x0 * q.y);
        }
        WF_UNROLL
        for(int e = 0; e < 2 * B1; ++e)
            r.win[j][e] = 1.0f;
    }
}
template<class G> WF_DEV void blu_mid(const TickArgs &a, int t, const cf *lds, P1Regs<G> &r)
{
    constexpr int R1 = G::R1, B1 = G::B1, M1 = G::M1;
    WF_UNROLL
    for(int j = 0; j < R1; ++j) {
        const int idx = j * M1 + B1 * t;
        cf bv[B1];
        const float *tab = reinterpret_cast<const float *>(a.blu_b + idx);
        if(B1 == 2) {
            const f4 q = ld4(tab);
            bv[0] = cf{q.x, q.y}; bv[B1 - 1] = cf{q.z, q.w};
        } else {
            const f2 q = ld2(tab);
            bv[0] = cf{q.x, q.y};
        }
        WF_UNROLL
        for(int b = 0; b < B1; ++b) {
            const cf c = cmul(lds_ld2(lds, ex3_addr<G>(idx + b)), bv[b]);
            r.smp[j][2 * b] = c.x;
            r.smp[j][2 * b + 1] = -c.y;
        }
        WF_UNROLL
        for(int e = 0; e < 2 * B1; ++e)
            r.win[j][e] = 1.0f;
        if(j >= 1 && tw1_row_loaded(j))
            p1_load_tw1<G>(a, t, j, r.tw1[j]);
    }
}

// Bluestein epilogue (packed form): the real split of the n/2-point transform the convolution delivers; slope, smoothing and
// the state store as in p4_slope_smooth_group, on the groups of four bins that lie inside the row
// MR (wf_mixed.hpp): the buffer holds Z itself -- no chirp factors, no conjugation
template<class G, bool TS, bool FPK, bool MR = false>
WF_DEV void p4_split_blu_impl(const TickArgs &a, int t, const cf *lds, float *ts, float (&mag)[G::P])
{
    // Z_k = blu_q[k] * conj(R_k), R = the twice-transformed buffer in natural order; X_k from Z_k and Z_(n'-k) by the real
    // split with W_n^k, as p4_split_smooth does for the power-of-two sizes: 2X[k] = (A + conj B) - i W (A - conj B)
    constexpr int T = G::T, P = G::P;
    const int np = (int)a.row_bins; // blu_n / 2
    WF_UNROLL
    for(int u = 0; u < P / 4; ++u) {
        const int k0 = 4 * (t + T * u);
        if(k0 < np) {
            const f4 sv = ld4(a.slope + k0);
            f4 st = f4{0.0f, 0.0f, 0.0f, 0.0f};
            if(TS)
                st = ld_state(ts + k0);
            f4 qa = f4{1.0f, 0.0f, 1.0f, 0.0f}, qb = qa, ra = qa, rb = qa;
            if constexpr(!MR) {
                qa = ld4(reinterpret_cast<const float *>(a.blu_q + k0));
                qb = ld4(reinterpret_cast<const float *>(a.blu_q + k0 + 2));
                ra = ld4(reinterpret_cast<const float *>(a.blu_qr + k0));
                rb = ld4(reinterpret_cast<const float *>(a.blu_qr + k0 + 2));
            }
            const f4 wa = ld4(reinterpret_cast<const float *>(a.blu_w + k0)), wb = ld4(reinterpret_cast<const float *>(a.blu_w + k0 + 2));
            const cf Q[4] = {cf{qa.x, qa.y}, cf{qa.z, qa.w}, cf{qb.x, qb.y}, cf{qb.z, qb.w}};
            const cf QR[4] = {cf{ra.x, ra.y}, cf{ra.z, ra.w}, cf{rb.x, rb.y}, cf{rb.z, rb.w}};
            const cf W[4] = {cf{wa.x, wa.y}, cf{wa.z, wa.w}, cf{wb.x, wb.y}, cf{wb.z, wb.w}};
            const float sl4[4] = {sv.x, sv.y, sv.z, sv.w}, st4v[4] = {st.x, st.y, st.z, st.w};
            WF_UNROLL
            for(int i = 0; i < 4; ++i) {
                const int k = k0 + i, km = (k == 0) ? 0 : np - k;
                const cf rk = lds_ld2(lds, MR ? mr_z_addr(a.mr, k) : ex3_addr<G>(k)), rm = lds_ld2(lds, MR ? mr_z_addr(a.mr, km) : ex3_addr<G>(km));
                const cf A = MR ? rk : cmul(cf{rk.x, -rk.y}, Q[i]), B = MR ? rm : cmul(cf{rm.x, -rm.y}, QR[i]);
                const float er = A.x + B.x, ei = A.y - B.y;
                const float dr = A.x - B.x, di = A.y + B.y;
                const float pr = fmaf(W[i].x, dr, -(W[i].y * di)); // Re(W D)
                const float pi = fmaf(W[i].x, di, W[i].y * dr);    // Im(W D)
                float m = mag2(er + pi, ei - pr) * a.half_coef * sl4[i];
                if(TS) {
                    float old = st4v[i];
                    if(FPK)
                        old = fmaxf(m, old);
                    m = spectrum_ema(a.g, old, a.g2, m);
                }
                mag[4 * u + i] = m;
            }
            if(TS)
                st_state(ts + k0, f4{mag[4 * u], mag[4 * u + 1], mag[4 * u + 2], mag[4 * u + 3]});
        }
    }
}
// ---- the mixed-radix sizes' epilogue (wf_mixed.hpp): the same real split, with its operands requested ahead ------------------
// n/2 <= M/2 bins (the transform works between the two halves of the container's exchange buffer): a thread owns at most P/8
// groups of four bins.  Per group: the smoothing state (f4) and W_n^k of its four bins (2 x f4); the slope factors are formed
// (1 + 3 k slope / (n/2 - 1), Policy<G>::SLOPE_LINEAR's reasoning).  One-wavefront containers request them right behind the
// window fetch -- the state only: 8 registers at 16 points per thread; with the twiddles too the 2048-sample container spilled
// 56 B per lane -- so that its trip to HBM runs under the transform instead of behind it (what the power-of-two one-wavefront
// geometries do: Policy MODE 1); everything else at the start of the epilogue, all at once.
#ifndef WF_MR_PREFETCH
#define WF_MR_PREFETCH 1
#endif
template<class G> struct MrOps {
    static constexpr int GR = (G::P / 8) > 0 ? G::P / 8 : 1;
    f4 st[GR], wa[GR], wb[GR];
};
// STATE: the smoothing state (HBM: the long trip); TWIDDLES: W_n^k (a table every stream shares: L2)
template<class G, bool STATE, bool TWIDDLES> WF_DEV void mr_ops_request(const TickArgs &a, int t, const float *ts, MrOps<G> &o)
{
    constexpr int T = G::T;
    const int np = (int)a.row_bins;
    WF_UNROLL
    for(int u = 0; u < MrOps<G>::GR; ++u) {
        const int k0 = 4 * (t + T * u);
        const int at = k0 < np ? k0 : 0; // (threads beyond the row load bin 0's operands and never use them)
        if(STATE)
            o.st[u] = (a.mode & WF_MODE_TSMOOTH) ? ld_state(ts + at) : f4{0.0f, 0.0f, 0.0f, 0.0f};
        if(TWIDDLES) {
            o.wa[u] = ld4(reinterpret_cast<const float *>(a.blu_w + at));
            o.wb[u] = ld4(reinterpret_cast<const float *>(a.blu_w + at + 2));
        }
    }
}
template<class G, bool TS, bool FPK>
WF_DEV void p4_split_mr_impl(const TickArgs &a, int t, const cf *lds, float *ts, const MrOps<G> &o, float (&mag)[G::P])
{
    constexpr int T = G::T;
    const int np = (int)a.row_bins; // blu_n / 2
    WF_UNROLL
    for(int u = 0; u < MrOps<G>::GR; ++u) {
        const int k0 = 4 * (t + T * u);
        if(k0 < np) {
            const cf W[4] = {cf{o.wa[u].x, o.wa[u].y}, cf{o.wa[u].z, o.wa[u].w}, cf{o.wb[u].x, o.wb[u].y}, cf{o.wb[u].z, o.wb[u].w}};
            const float st4v[4] = {o.st[u].x, o.st[u].y, o.st[u].z, o.st[u].w};
            WF_UNROLL
            for(int i = 0; i < 4; ++i) {
                const int k = k0 + i, km = (k == 0) ? 0 : np - k;
                const cf A = lds_ld2(lds, mr_z_addr(a.mr, k)), B = lds_ld2(lds, mr_z_addr(a.mr, km));
                const float er = A.x + B.x, ei = A.y - B.y;
                const float dr = A.x - B.x, di = A.y + B.y;
                const float pr = fmaf(W[i].x, dr, -(W[i].y * di)); // Re(W D)
                const float pi = fmaf(W[i].x, di, W[i].y * dr);    // Im(W D)
                float m = mag2(er + pi, ei - pr) * a.half_coef * fmaf((float)k, a.slope_step, 1.0f);
                if(TS) {
                    float old = st4v[i];
                    if(FPK)
                        old = fmaxf(m, old);
                    m = spectrum_ema(a.g, old, a.g2, m);
                }
                mag[4 * u + i] = m;
            }
            if(TS)
                st_state(ts + k0, f4{mag[4 * u], mag[4 * u + 1], mag[4 * u + 2], mag[4 * u + 3]});
        }
    }
}
template<class G> WF_DEV void p4_mr(const TickArgs &a, int t, const cf *lds, float *ts, const MrOps<G> &o, float (&mag)[G::P])
{
    if(a.mode & WF_MODE_TSMOOTH) {
        if(a.mode & WF_MODE_FAST_PEAKS)
            p4_split_mr_impl<G, true, true>(a, t, lds, ts, o, mag);
        else
            p4_split_mr_impl<G, true, false>(a, t, lds, ts, o, mag);
    } else
        p4_split_mr_impl<G, false, false>(a, t, lds, ts, o, mag);
}

template<class G, bool MR = false> WF_DEV void p4_direct(const TickArgs &a, int t, const cf *lds, float *ts, float (&mag)[G::P])
{
    if(a.mode & WF_MODE_TSMOOTH) {
        if(a.mode & WF_MODE_FAST_PEAKS)
            p4_split_blu_impl<G, true, true, MR>(a, t, lds, ts, mag);
        else
            p4_split_blu_impl<G, true, false, MR>(a, t, lds, ts, mag);
    } else
        p4_split_blu_impl<G, false, false, MR>(a, t, lds, ts, mag);
}

// ---- P2: pass 2 ---------------------------------------------------------------------------------
template<class G> WF_DEV void p2_read(int t, const cf *lds, cf (&v)[G::P])
{
    constexpr int R2 = G::R2, R3 = G::R3, B2 = G::B2;
    if constexpr(G::H2 == 2) {
        // radix 32 with 16 points per thread (N = 32768): threads c and c + R1*R3 share column c = k1*R3 + n3.  Both read
        // the whole column; thread h keeps the half of the first radix-2 stage that feeds the outputs k2 = 2k' + h, as
        // pass 3 does: u[j] = (v[j] + sg * v[j+16]) * (h ? W_32^j : 1).
        constexpr int HALF = R2 / 2;
        static_assert(R2 == 32 || R2 == 16 || R2 == 8, "the shared second pass: radix 32 / 16 / 8 with 16 / 8 / 4 points per thread");
        const int c = t & (G::R1 * R3 - 1), h = t / (G::R1 * R3);
        const int k1 = c / R3, n3 = c % R3;
        const float hf = (float)h;
        const float sg = 1.0f - 2.0f * hf;
        WF_UNROLL
        for(int j = 0; j < HALF; ++j) {
            const cf lo = lds_ld2(lds, ex1_addr<G>(k1, j * R3 + n3));
            const cf hi = lds_ld2(lds, ex1_addr<G>(k1, (j + HALF) * R3 + n3));
            const cf e = cf{fmaf(sg, hi.x, lo.x), fmaf(sg, hi.y, lo.y)};
            v[j] = cmul(e, half_twiddle32(j * (32 / R2), hf)); // W_R2^j
        }
        return;
    }
    const int q0 = B2 * t;
    const int k1 = q0 / R3, n30 = q0 % R3;
    WF_UNROLL
    for(int n2 = 0; n2 < R2; ++n2) {
        const int base = ex1_addr<G>(k1, n2 * R3 + n30);
        if(B2 == 1) {
            v[n2] = lds_ld2(lds, base);
        } else {
            WF_UNROLL
            for(int b = 0; b < B2; b += 2) {
                const f4 q = lds_ld4(lds, base + b);
                v[b * R2 + n2] = cf{q.x, q.y};
                v[(b + 1) * R2 + n2] = cf{q.z, q.w};
            }
        }
    }
}

// tw2: the workgroup's LDS copy of the pass-2 twiddle table [R2][R3] (no vector-memory wait in the middle of the FFT,
// which would also wait for the state prefetch issued before it: loads return in order)
template<class G> WF_DEV void p2_pass2_write(const cf *tw2, int t, cf *lds, cf (&v)[G::P])
{
    constexpr int R1 = G::R1, R2 = G::R2, R3 = G::R3, B2 = G::B2;
    if constexpr(G::H2 == 2) {
        constexpr int HALF = R2 / 2, LBH = ilog2(HALF);
        const int c = t & (R1 * R3 - 1), h = t / (R1 * R3);
        const int k1 = c / R3, n3 = c % R3;
        cf u[HALF];
        WF_UNROLL
        for(int j = 0; j < HALF; ++j)
            u[j] = v[j];
        dft_dif<HALF>(u);
        WF_UNROLL
        for(int kk = 0; kk < HALF; ++kk) {
            const int k2 = 2 * kk + h;                 // this thread's outputs
            const cf w = lds_ld2(tw2, k2 * R3 + n3);   // W_(R2 R3)^(n3 k2); row 0 is all ones
            lds_st2(lds, ex2_addr<G>(k1 + R1 * k2, n3), cmul(u[brev(kk, LBH)], w));
        }
        return;
    }
    constexpr int LB = ilog2(R2);
    const int q0 = B2 * t;
    const int k1 = q0 / R3, n30 = q0 % R3;
    const int a20 = ex2_addr<G>(k1, n30);
    cf u[B2][R2];
    WF_UNROLL
    for(int b = 0; b < B2; ++b) {
        WF_UNROLL
        for(int n2 = 0; n2 < R2; ++n2)
            u[b][n2] = v[b * R2 + n2];
        dft_dif<R2>(u[b]);
    }
    WF_UNROLL
    for(int k2 = 0; k2 < R2; ++k2) {
        cf o[B2];
        if(k2 == 0) {
            WF_UNROLL
            for(int b = 0; b < B2; ++b)
                o[b] = u[b][0];
        } else if(B2 == 1) {
            const cf w = lds_ld2(tw2, k2 * R3 + n30);
            o[0] = cmul(u[0][brev(k2, LB)], w);
        } else {
            WF_UNROLL
            for(int b = 0; b < B2; b += 2) {
                const f4 w = lds_ld4(tw2, k2 * R3 + n30 + b);
                o[b] = cmul(u[b][brev(k2, LB)], cf{w.x, w.y});
                o[b + 1] = cmul(u[b + 1][brev(k2, LB)], cf{w.z, w.w});
            }
        }
        const int q = k1 + R1 * k2;
        if(B2 == 1) {
            // == ex2_addr<G>(q, n30): two distinct base registers per thread, the rest is an immediate offset
            lds_st2(lds, (a20 ^ ex2_xor<G>(k2)) + k2 * R1 * R3, o[0]);
        } else {
            WF_UNROLL
            for(int b = 0; b < B2; b += 2)
                lds_st4(lds, ex2_addr<G>(q, n30 + b), o[b], o[b + 1]);
        }
    }
}

// ---- P3: pass 3 ---------------------------------------------------------------------------------
template<class G> WF_DEV void p3_read(int t, const cf *lds, cf (&v)[G::P])
{
    constexpr int R3 = G::R3, B3 = G::B3, T = G::T;
    if constexpr(G::H3 == 2) {
        // radix 32 with 16 points per thread: threads q and q + R1*R2 share row q.  Both read the whole row; thread h
        // keeps the first radix-2 stage's half that feeds the outputs k3 = 2k' + h:
        //   h = 0: s[j] = v[j] + v[j+16]        h = 1: d[j] = (v[j] - v[j+16]) * W_32^j
        // written select-free: u[j] = (v[j] + sg * v[j+16]) * (h ? W_32^j : 1).
        constexpr int HALF = R3 / 2;
        const int q = t & (G::R1 * G::R2 - 1), h = t / (G::R1 * G::R2); // the two threads of row q sit in different wavefronts
        const float hf = (float)h;
        const float sg = 1.0f - 2.0f * hf;
        WF_UNROLL
        for(int j = 0; j < HALF; j += 2) {
            const f4 lo = lds_ld4(lds, ex2_addr<G>(q, j));
            const f4 hi = lds_ld4(lds, ex2_addr<G>(q, j + HALF));
            const cf e0 = cf{fmaf(sg, hi.x, lo.x), fmaf(sg, hi.y, lo.y)};
            const cf e1 = cf{fmaf(sg, hi.z, lo.z), fmaf(sg, hi.w, lo.w)};
            v[j] = cmul(e0, half_twiddle32(j * (32 / R3), hf)); // W_R3^j
            v[j + 1] = cmul(e1, half_twiddle32((j + 1) * (32 / R3), hf));
        }
        return;
    }
    WF_UNROLL
    for(int b = 0; b < B3; ++b) {
        const int q = t + T * b;
        WF_UNROLL
        for(int n3 = 0; n3 < R3; n3 += 2) {
            const f4 w = lds_ld4(lds, ex2_addr<G>(q, n3));
            v[b * R3 + n3] = cf{w.x, w.y};
            v[b * R3 + n3 + 1] = cf{w.z, w.w};
        }
    }
}

template<class G> WF_DEV void p3_pass3_write(int t, cf *lds, cf (&v)[G::P])
{
    constexpr int R1 = G::R1, R2 = G::R2, R3 = G::R3, B3 = G::B3, T = G::T;
    if constexpr(G::H3 == 2) {
        constexpr int HALF = R3 / 2, LBH = ilog2(HALF);
        static_assert((R1 * R2) % 4 == 0, "ex3 stride algebra");
        const int q = t & (R1 * R2 - 1), h = t / (R1 * R2);
        cf u[HALF];
        WF_UNROLL
        for(int j = 0; j < HALF; ++j)
            u[j] = v[j];
        dft_dif<HALF>(u);
        const int a3 = ex3_addr<G>(q + R1 * R2 * h); // == ex3_addr<G>(q + R1*R2*(2*kk + h)) - kk * ex3_step(2*R1*R2)
        WF_UNROLL
        for(int kk = 0; kk < HALF; ++kk)
            lds_st2(lds, a3 + kk * ex3_step<G>(2 * R1 * R2), u[brev(kk, LBH)]);
        return;
    }
    constexpr int LB = ilog2(R3);
    WF_UNROLL
    for(int b = 0; b < B3; ++b) {
        cf u[R3];
        WF_UNROLL
        for(int n3 = 0; n3 < R3; ++n3)
            u[n3] = v[b * R3 + n3];
        dft_dif<R3>(u);
        static_assert((R1 * R2) % 4 == 0, "ex3 stride algebra");
        const int q = t + T * b;
        const int a3 = ex3_addr<G>(q); // == ex3_addr<G>(q + R1*R2*k3) - k3 * ex3_step(R1*R2)
        WF_UNROLL
        for(int k3 = 0; k3 < R3; ++k3)
            lds_st2(lds, a3 + k3 * ex3_step<G>(R1 * R2), u[brev(k3, LB)]);
    }
}

// ---- P4: real split + epilogue ------------------------------------------------------------------
// Produces the smoothed linear magnitudes of bins 4g..4g+3 (g = t + T*u) in mag[u][0..3], updating the temporal-smoothing
// state on the way (reference :110-135).  The smoothing mode is a property of the configuration: the kernel branches on it
// once (a scalar branch) into a body compiled for it -- TS: m_tsmoothing != NONE, FPK: m_fast_peaks -- instead of testing
// mode bits per bin (which also made every conditionally loaded register a zero-initialised one).
//
// slope (reference :121-122) and temporal smoothing incl. fast peaks (:124-132) of bin group u; the slope table is all ones
// when m_slope <= 0.  st4v/sl4 are this group's m_tsmooth_buf / m_slope_modifiers values.
template<class G, bool TS, bool FPK, bool DEFER = false>
WF_DEV void p4_slope_smooth_group(const TickArgs &a, int t, int u, float *ts, const float (&st4v)[4], const float (&sl4)[4],
                                  float (&mag)[G::P])
{
    constexpr int T = G::T;
    const int k0 = 4 * (t + T * u);
    WF_UNROLL
    for(int i = 0; i < 4; ++i)
        mag[4 * u + i] *= sl4[i];
    if(TS) {
        WF_UNROLL
        for(int i = 0; i < 4; ++i) {
            float old = st4v[i];
            if(FPK)
                old = fmaxf(mag[4 * u + i], old);
            // (g * oldval) + (g2 * mag) (reference :130); evaluated as fma(g, old, g2*mag) like the reference's own
            // AVX2 path (src/source_avx2.cpp:154) -- within 1 ulp of the generic path's separately rounded sum
            mag[4 * u + i] = spectrum_ema(a.g, old, a.g2, mag[4 * u + i]);
        }
        if(!DEFER)
            st_state(ts + k0, f4{mag[4 * u], mag[4 * u + 1], mag[4 * u + 2], mag[4 * u + 3]});
    }
}
// the state stores a DEFER-ed P4 left out: m_tsmooth_buf = the smoothed magnitudes (reference :131), for the whole row
template<class G> WF_DEV void p4_store_state(const TickArgs &a, int t, float *ts, const float (&mag)[G::P])
{
    if(!(a.mode & WF_MODE_TSMOOTH))
        return;
    WF_UNROLL
    for(int u = 0; u < G::P / 4; ++u)
        st_state(ts + 4 * (t + G::T * u), f4{mag[4 * u], mag[4 * u + 1], mag[4 * u + 2], mag[4 * u + 3]});
}

struct NoMid { WF_DEV void operator()() const {} };
// mid(): called between the real split and the smoothing (whose state stores follow): where the kernel requests the display's
// tables -- in front of the stores in the memory pipeline, behind the registers' peak
template<class G, bool TS, bool FPK, bool DEFER = false, class Mid = NoMid>
WF_DEV void p4_split_smooth_impl(const TickArgs &a, int t, const cf *lds, float *ts, const cf (&wb)[4], const P4Regs<G> &q,
                                 float (&mag)[G::P], Mid mid = Mid{})
{
    constexpr int M = G::M, T = G::T, P = G::P;
    // Threads that did not prefetch state/slope into registers earlier issue ALL of those loads now and consume them only
    // after the whole real split (two loops), so the split math covers their (L2) latency.
    constexpr bool LOAD_ALL_FIRST = !Policy<G>::PREFETCH_STATE;
    float st_all[LOAD_ALL_FIRST ? P : 4], sl_all[(LOAD_ALL_FIRST && !Policy<G>::SLOPE_LINEAR) ? P : 4];
    if(LOAD_ALL_FIRST) {
        WF_UNROLL
        for(int u = 0; u < P / 4; ++u) {
            const int k0 = 4 * (t + T * u);
            if(TS) {
                const f4 o = ld_state(ts + k0);
                st_all[4 * u] = o.x; st_all[4 * u + 1] = o.y; st_all[4 * u + 2] = o.z; st_all[4 * u + 3] = o.w;
            }
            if(!Policy<G>::SLOPE_LINEAR) {
                constexpr int S = Policy<G>::SLOPE_LINEAR ? 0 : 1;
                const f4 sv = ld4(a.slope + k0);
                sl_all[S * (4 * u)] = sv.x; sl_all[S * (4 * u + 1)] = sv.y; sl_all[S * (4 * u + 2)] = sv.z; sl_all[S * (4 * u + 3)] = sv.w;
            }
        }
    }
    // this group's state / slope operands, from wherever the policy put them
    auto group = [&](int u) {
        const int k0 = 4 * (t + T * u);
        float st4v[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sl4[4];
        if(Policy<G>::PREFETCH_SLOPE) {
            WF_UNROLL
            for(int i = 0; i < 4; ++i)
                sl4[i] = q.sl[(Policy<G>::PREFETCH_SLOPE ? 1 : 0) * (4 * u + i)];
        } else if(LOAD_ALL_FIRST && Policy<G>::SLOPE_LINEAR) {
            WF_UNROLL // (formed, not loaded: Policy<G>::SLOPE_LINEAR)
            for(int i = 0; i < 4; ++i)
                sl4[i] = fmaf((float)(k0 + i), a.slope_step, 1.0f);
        } else if(LOAD_ALL_FIRST) {
            WF_UNROLL
            for(int i = 0; i < 4; ++i)
                sl4[i] = sl_all[(LOAD_ALL_FIRST ? 1 : 0) * (4 * u + i)];
        } else {
            const f4 o = ld4(a.slope + k0);
            sl4[0] = o.x; sl4[1] = o.y; sl4[2] = o.z; sl4[3] = o.w;
        }
        if(TS) {
            if(Policy<G>::PREFETCH_STATE) {
                WF_UNROLL
                for(int i = 0; i < 4; ++i)
                    st4v[i] = q.st[(Policy<G>::PREFETCH_STATE ? 1 : 0) * (4 * u + i)];
            } else if(LOAD_ALL_FIRST) {
                WF_UNROLL
                for(int i = 0; i < 4; ++i)
                    st4v[i] = st_all[(LOAD_ALL_FIRST ? 1 : 0) * (4 * u + i)];
            } else {
                const f4 o = ld_state(ts + k0);
                st4v[0] = o.x; st4v[1] = o.y; st4v[2] = o.z; st4v[3] = o.w;
            }
        }
        p4_slope_smooth_group<G, TS, FPK, DEFER>(a, t, u, ts, st4v, sl4, mag);
    };
    // ---- loop 1: real split -> |2X| * coef/2 -------------------------------------------------------------------------
    // LDS addresses: bins advance by 4T per group, i.e. by T words inside every ex3 plane, so every group is the first
    // one's address plus a compile-time step.  Z[k0 + i] sits at aA[i]; the mirrored bins M - k0 - i at aB[i] - u*step,
    // except bin M - 0 = 0 for the very first bin of thread 0.
    int aA[4], aB[4];
    WF_UNROLL
    for(int i = 0; i < 4; ++i) {
        aA[i] = ex3_addr<G>(4 * t + i);
        aB[i] = ex3_addr<G>(M - 4 * t - i - 4 * T * (P / 4 - 1)); // the LAST group's address: offsets below stay non-negative
    }
    const int aB00 = ex3_addr<G>((M - 4 * t) & (M - 1));
    WF_UNROLL
    for(int u = 0; u < P / 4; ++u) {
        cf A[4];
        WF_UNROLL
        for(int i = 0; i < 4; ++i)
            A[i] = lds_ld2(lds, aA[i] + u * ex3_step<G>(4 * T));
        WF_UNROLL
        for(int i = 0; i < 4; ++i) {
            const cf W = mul_w32(wb[i], u * (64 / P)); // W_N^(k0 + i) = wb[i] * W_N^(4Tu)
            const cf B = lds_ld2(lds, (u == 0 && i == 0) ? aB00 : aB[i] + (P / 4 - 1 - u) * ex3_step<G>(4 * T));
            // 2X[k] = (A + conj B) - i W (A - conj B)
            const float er = A[i].x + B.x, ei = A[i].y - B.y;
            const float dr = A[i].x - B.x, di = A[i].y + B.y;
            const float pr = fmaf(W.x, dr, -(W.y * di)); // Re(W D)
            const float pi = fmaf(W.x, di, W.y * dr);    // Im(W D)
            const float xr = er + pi, xi = ei - pr;
            mag[4 * u + i] = mag2(xr, xi) * a.half_coef;
        }
        if(!LOAD_ALL_FIRST)
            group(u);
    }
    mid();
    // ---- loop 2: slope, temporal smoothing, state store ----------------------------------------------------------------
    if(LOAD_ALL_FIRST) {
        WF_UNROLL
        for(int u = 0; u < P / 4; ++u)
            group(u);
    }
}

// DEFER: the smoothing-state stores are left to the caller (p4_store_state), which issues them behind the requests for the
// display's tables: vector memory completes in order, so tables requested behind the state stores are not in before the
// stores' acknowledgement -- a round trip to HBM the dot products at the end of the kernel then sit out
template<class G, bool DEFER = false, class Mid = NoMid>
WF_DEV void p4_split_smooth(const TickArgs &a, int t, const cf *lds, float *ts, const cf (&wb)[4], const P4Regs<G> &q, float (&mag)[G::P], Mid mid = Mid{})
{
    if(a.mode & WF_MODE_TSMOOTH) {
        if(a.mode & WF_MODE_FAST_PEAKS)
            p4_split_smooth_impl<G, true, true, DEFER>(a, t, lds, ts, wb, q, mag, mid);
        else
            p4_split_smooth_impl<G, true, false, DEFER>(a, t, lds, ts, wb, q, mag, mid);
    } else
        p4_split_smooth_impl<G, false, false, DEFER>(a, t, lds, ts, wb, q, mag, mid);
}

// ---- decimated epilogue (DEC > 0): the N >> DEC point transform's bin o is bin o << DEC of the zero-padded one --------------
// Row geometry of the outputs: MO = M >> DEC bins, the first MO / 4 threads own four consecutive ones each.
template<int TT, int PP> struct RowG { static constexpr int T = TT, P = PP; };

template<class G, int DEC> WF_DEV void p4_prefetch_dec(const TickArgs &a, int t, const float *ts, P4Regs<G> &q)
{
    static_assert(Policy<G>::PREFETCH_STATE && Policy<G>::PREFETCH_SLOPE, "the decimated path runs on a one-wavefront geometry");
    constexpr int TO = (G::M >> DEC) / 4;
    const int tt = t < TO ? t : 0;
    if(a.mode & WF_MODE_TSMOOTH) {
        const f4 o = ld_state(ts + 4 * tt);
        q.st[0] = o.x; q.st[1] = o.y; q.st[2] = o.z; q.st[3] = o.w;
    }
    const f4 sv = ld4(a.slope + 4 * tt);
    q.sl[0] = sv.x; q.sl[1] = sv.y; q.sl[2] = sv.z; q.sl[3] = sv.w;
}

// Returns the four magnitudes by value (not through the caller's array): with the array written inside each of the three
// mode arms, ROCm 7.2's clang sinks the arms' last stores -- one into scratch, two into the state row -- into a single store
// through a pointer phi, a flat pointer its backend then fails to select ("Operand has incorrect register class").
template<class G, int DEC, bool TS, bool FPK>
WF_DEV f4 p4_split_smooth_dec_impl(const TickArgs &a, int t, const cf *lds, float *ts, const cf (&wb)[4], const P4Regs<G> &q)
{
    constexpr int M = G::M;
    float mag[4];
    WF_UNROLL
    for(int i = 0; i < 4; ++i) {
        const int k = (4 * t + i) << DEC;
        const cf A = lds_ld2(lds, ex3_addr<G>(k));
        const cf B = lds_ld2(lds, ex3_addr<G>((M - k) & (M - 1)));
        const cf W = wb[i];
        const float er = A.x + B.x, ei = A.y - B.y;
        const float dr = A.x - B.x, di = A.y + B.y;
        const float pr = fmaf(W.x, dr, -(W.y * di));
        const float pi = fmaf(W.x, di, W.y * dr);
        const float xr = er + pi, xi = ei - pr;
        float m = mag2(xr, xi) * a.half_coef;
        m *= q.sl[i];
        if(TS) {
            float old = q.st[i];
            if(FPK)
                old = fmaxf(m, old);
            m = spectrum_ema(a.g, old, a.g2, m);
        }
        mag[i] = m;
    }
    const f4 r = f4{mag[0], mag[1], mag[2], mag[3]};
    if(TS)
        st_state(ts + 4 * t, r);
    return r;
}
template<class G, int DEC>
WF_DEV void p4_split_smooth_dec(const TickArgs &a, int t, const cf *lds, float *ts, const cf (&wb)[4], const P4Regs<G> &q, float (&mag)[4])
{
    f4 r;
    if(a.mode & WF_MODE_TSMOOTH) {
        if(a.mode & WF_MODE_FAST_PEAKS)
            r = p4_split_smooth_dec_impl<G, DEC, true, true>(a, t, lds, ts, wb, q);
        else
            r = p4_split_smooth_dec_impl<G, DEC, true, false>(a, t, lds, ts, wb, q);
    } else
        r = p4_split_smooth_dec_impl<G, DEC, false, false>(a, t, lds, ts, wb, q);
    mag[0] = r.x; mag[1] = r.y; mag[2] = r.z; mag[3] = r.w;
}

// dB conversion + volume normalisation + roll-off of this thread's bins (reference :144-179); d[] is the
// final m_decibels content, stored by store_row()
// GUARD (rows shorter than T * P bins: the Bluestein path): groups of four bins at or beyond `nb` are skipped
template<class G, bool GUARD = false>
WF_DEV void p4_db(const TickArgs &a, int t, const float (&mag)[G::P], float (&d)[G::P], float vol_comp, int nb = 0, int kbase = 0)
{
    constexpr int T = G::T, P = G::P;
    // roll-off row: all of this thread's vectors requested before the first logarithm -- one round trip to L2 under the dB
    // math instead of one per group of four bins, each waited for on the spot
    f4 roll[P / 4];
    if(a.mode & WF_MODE_ROLLOFF) {
        WF_UNROLL
        for(int u = 0; u < P / 4; ++u) {
            const int k0 = 4 * (t + T * u);
            roll[u] = (GUARD && k0 >= nb) ? f4{0.0f, 0.0f, 0.0f, 0.0f} : ld4(a.rolloff + k0 + kbase);
        }
    }
    WF_UNROLL
    for(int u = 0; u < P / 4; ++u) {
        int k0 = 4 * (t + T * u);
        if(GUARD && k0 >= nb)
            continue;
        k0 += kbase; // rows longer than T * P bins are finished in parts (wf_big.hpp): the bin index of the whole row
        WF_UNROLL
        for(int i = 0; i < 4; ++i)
            d[4 * u + i] = dbfs(mag[4 * u + i], a.db_min);
        if(a.mode & WF_MODE_NORMALIZE) {
            WF_UNROLL
            for(int i = 0; i < 4; ++i)
                if(k0 + i >= 1) // the generic path starts at i = 1 (reference :165)
                    d[4 * u + i] += vol_comp;
        }
    }
    if(a.mode & WF_MODE_ROLLOFF) {
        WF_UNROLL
        for(int u = 0; u < P / 4; ++u) {
            const int k0 = 4 * (t + T * u);
            if(GUARD && k0 >= nb)
                continue;
            const float rr[4] = {roll[u].x, roll[u].y, roll[u].z, roll[u].w};
            WF_UNROLL
            for(int i = 0; i < 4; ++i)
                if(k0 + kbase + i >= 1) // reference :173
                    d[4 * u + i] = fmaxf(d[4 * u + i] - rr[i], a.db_min);
        }
    }
}

template<class G, bool GUARD = false, bool NT = false> WF_DEV void store_row(float *row, int t, const float (&d)[G::P], int nb = 0)
{
    constexpr int T = G::T, P = G::P;
    WF_UNROLL
    for(int u = 0; u < P / 4; ++u)
        if(!GUARD || 4 * (t + T * u) < nb) {
            if constexpr(NT)
                st4_nt(row + 4 * (t + T * u), f4{d[4 * u], d[4 * u + 1], d[4 * u + 2], d[4 * u + 3]});
            else
                st4(row + 4 * (t + T * u), f4{d[4 * u], d[4 * u + 1], d[4 * u + 2], d[4 * u + 3]});
        }
}
// ---- silence state machine helpers (reference :63-95, :138-139) ------------------------------------
// "outsilent": every value of the previously displayed row is <= floor - 10.  Each thread looks at the
// bins it owns in the P4 layout; the caller and-reduces over the spectrum's threads.
template<class G, bool GUARD = false> WF_DEV bool row_all_below(const float *row, int t, float limit, int nb = 0)
{
    constexpr int T = G::T, P = G::P;
    bool below = true;
    WF_UNROLL
    for(int u = 0; u < P / 4; ++u) {
        if(GUARD && 4 * (t + T * u) >= nb)
            continue;
        const f4 d = ld4(row + 4 * (t + T * u));
        below = below && !(d.x > limit) && !(d.y > limit) && !(d.z > limit) && !(d.w > limit);
    }
    return below;
}

// a skipped channel of a stream that is not silent: the reference's end-of-tick dB pass runs over its
// *stale* m_decibels row again; load that row as the "magnitudes"
template<class G, bool GUARD = false> WF_DEV void load_row(const float *row, int t, float (&mag)[G::P], int nb = 0)
{
    constexpr int T = G::T, P = G::P;
    WF_UNROLL
    for(int u = 0; u < P / 4; ++u) {
        if(GUARD && 4 * (t + T * u) >= nb)
            continue;
        const f4 d = ld4(row + 4 * (t + T * u));
        mag[4 * u] = d.x; mag[4 * u + 1] = d.y; mag[4 * u + 2] = d.z; mag[4 * u + 3] = d.w;
    }
}

template<class G, bool GUARD = false> WF_DEV void fill_row(float *row, int t, float v, int nb = 0)
{
    constexpr int T = G::T, P = G::P;
    WF_UNROLL
    for(int u = 0; u < P / 4; ++u)
        if(!GUARD || 4 * (t + T * u) < nb)
            st4(row + 4 * (t + T * u), f4{v, v, v, v});
}

// The per-stream decision of the channel loop, replayed from the per-channel facts.
//   last_silent : m_last_silent on entry
//   nz[c]       : channel c's window has a non-zero sample
//   below[c]    : the row the reference inspects for channel c (m_decibels[stereo ? c : 0] as left by the
//                 previous tick) is entirely <= floor - 10
// Outputs process[c] (run the FFT path for channel c) and the new m_last_silent.
struct StreamPlan { bool process0, process1; bool last_silent; };
WF_DEV StreamPlan plan_stream(bool last_silent, uint32_t cap_ch, bool stereo, bool nz0, bool nz1, bool below0, bool below1)
{
    StreamPlan p;
    p.process0 = p.process1 = false;
    bool ls = last_silent;
    uint32_t silent_channels = 0;
    // channel 0
    if(nz0) {
        ls = false;              // reference :68-69
        p.process0 = true;
    } else if(!ls) {             // :76-77
        if(below0) {             // :78-94
            if(++silent_channels >= cap_ch)
                ls = true;
        } else
            p.process0 = true;
    }
    // channel 1
    if(cap_ch > 1) {
        if(nz1) {
            ls = false;
            p.process1 = true;
        } else if(!ls) {
            // In mono display mode channel 1 inspects row 0, which channel 0 has just overwritten with linear
            // magnitudes (>= 0 > floor - 10) if it was processed in this tick.
            const bool outsilent = below1 && !(!stereo && p.process0);
            if(outsilent) {
                if(++silent_channels >= cap_ch)
                    ls = true;
            } else
                p.process1 = true;
        }
    }
    p.last_silent = ls;
    return p;
}

// ---- bars (render_bars interpolation + dB -> pixel mapping, reference src/source.cpp:1500-1557) -------

// std::lerp as libstdc++ evaluates it (reference src/math_funcs.hpp:31-35)
WF_DEV float lerp_std(float a, float b, float t)
{
    if((a <= 0 && b >= 0) || (a >= 0 && b <= 0))
        return t * b + (1 - t) * a;
    if(t == 1)
        return b;
    const float x = a + t * (b - a);
    return ((t > 1) == (b > a)) ? (b < x ? x : b) : (b > x ? x : b);
}

// All T threads of a spectrum reduce the bars of one displayed row.
//   db   : the row's dB values in LDS (float[M]);  prod: LDS scratch for the products (capacity checked on the host)
//   sync : barrier over the spectrum's threads;  xor_sum(v, m): v + (v of lane ^ m) for m < 64 (wave shuffle)
// What a thread needs to know about "its" bar in the first pass of the first chunk (bar = t / lanes_per_bar).  Fetched at
// the very start of the kernel with the audio window, so that the bars phase at the end does not begin with a chain of
// dependent table loads.
struct BarPre { int off, len, count; int s0, s1; int glen; int lead; };
template<class G, bool PIECES = true, bool PS_OK = true> WF_DEV BarPre bars_preload(const BarArgs &b, int t)
{
    BarPre p{0, 0, 1, 0, 0, 0, -1};
    if(b.out != nullptr && !(PS_OK && b.ps_lanes > 0)) { // (the prefix-sum layout keeps everything in its lane table: bars_fetch_entries)
        if(PIECES && b.num_segs > 0 && b.piece_mode) {
            // glen: the scan flags and 1 + the slot this lane's piece total goes to.  One wavefront per spectrum: the slot is the
            // bar itself (lead / count); several: lane l of whichever wavefront arrives last finishes bar l from slots [s0, s1)
            p.glen = b.seg_group[t];
            if(G::T == 64) {
                p.lead = (int)((uint32_t)p.glen >> 8) - 1;
                if(p.lead >= 0)
                    p.count = b.count[p.lead];
            } else if((t & 63) < b.num_bars) {
                p.lead = t & 63;
                p.s0 = b.bar_seg[p.lead];
                p.s1 = b.bar_seg[p.lead + 1];
                p.count = b.count[p.lead];
            }
        } else if(b.num_segs > 0 && b.wave_local) { // the bar this thread leads, if any
            p.lead = b.lead_bar[t];
            p.s0 = t;
            p.s1 = b.lead_end[t]; // one past the last segment of this thread's bar
            if(p.lead >= 0)
                p.count = b.count[p.lead];
        } else if(b.num_segs > 0) { // bar t's segment range
            p.glen = b.seg_group[t];
            if(t < b.num_bars) {
                p.s0 = b.bar_seg[t];
                p.s1 = b.bar_seg[t + 1];
                p.count = b.count[t];
            }
        } else {
            const int bar = t / b.lanes_per_bar;
            if(bar < b.num_bars) {
                p.off = b.off[bar];
                p.len = b.off[bar + 1] - p.off;
                p.count = b.count[bar];
            }
        }
    }
    return p;
}

// The (coefficient, bin) pairs of this thread's segment: 16-byte loads, coalesced across the threads, requested before
// the dB math so that their L2 latency is off the critical path.
template<class G> struct BarEntries {
    static constexpr int CMAX = G::P / 4 + 2; // the host builds segments of at most 4 * CMAX entries; the prefix-sum layout's lane table is three words
    f4 coef[CMAX];
    int base;
};
template<class G, bool PS_OK = true, bool PS_ONLY = false> WF_DEV void bars_fetch_entries(const BarArgs &b, int t, BarEntries<G> &be, bool finisher = false)
{
    constexpr int T = G::T;
    be.base = 0;
    if constexpr(PS_ONLY) { // (the instantiation displays bars in the prefix-sum layout and nothing else: spectrum_tick_kernel<.., DISP = 1>)
        if(finisher) { // wave-uniform
            const float *p = b.ps_tab + (size_t)(t & 63) * 4;
            WF_UNROLL
            for(int c = 0; c < 3; ++c)
                be.coef[c] = ld4(p + c * 256);
        }
        return;
    }
    if constexpr(!PS_OK) { // (the instantiation never runs the prefix-sum layout -- the Bluestein / mixed-radix kernels at their register caps)
        if(b.out == nullptr || b.num_segs == 0)
            return;
        be.base = b.lane_base[t];
        WF_UNROLL
        for(int c = 0; c < BarEntries<G>::CMAX; ++c)
            if(c < b.lane_blocks) // uniform
                be.coef[c] = ld4(b.lane_coef + (c * T + t) * 4);
        return;
    }
    if(b.out == nullptr || (b.num_segs == 0 && b.ps_lanes == 0))
        return;
    // One loop for both layouts (two loops writing the one array left half of it in scratch: ROCm 7.2's SROA gives up on the phi).
    // Prefix-sum layout: the three 16-byte words of this lane's sub-band edge (BarPsTables), requested only by the wavefront that
    // finishes sub-bands (`finisher`, wave-uniform).
    const bool ps = b.ps_lanes > 0;
    const float *p = ps ? b.ps_tab + (size_t)(t & 63) * 4 : b.lane_coef + (size_t)t * 4;
    const int stride = ps ? 256 : T * 4;
    int n = ps ? (finisher ? 3 : 0) : b.lane_blocks;
#if defined(__HIPCC__)
    n = __builtin_amdgcn_readfirstlane(n);
#endif
    if(!ps)
        be.base = b.lane_base[t];
    WF_UNROLL
    for(int c = 0; c < BarEntries<G>::CMAX; ++c)
        if(c < n) // uniform
            be.coef[c] = ld4(p + c * stride);
}

// mean dB of output o -> pixel row (reference src/source.cpp:1548-1557 bars, :1411 curve):
// y = lerp(border_top, border_bottom, clamp(ceiling - v, 0, range) / range).  Straight-line code: the quotient is
// tt * (1 / range) corrected by one residual step (q + fma(-q, range, tt) * inv: the correctly rounded quotient but for
// rare last-bit ties), and std::lerp's case distinction on the signs of its end points is a property of the configuration
// (BarArgs::lerp_mixed) -- testing it per output cost a dozen scalar branches per point.
WF_DEV float map_output(const BarArgs &b, float v)
{
    float tt = b.ceiling - v; // reference src/source.cpp:1550
    tt = fminf(fmaxf(tt, 0.0f), b.dbrange);
    float q = tt * b.inv_dbrange;
    q = fmaf(fmaf(-q, b.dbrange, tt), b.inv_dbrange, q);
    if(b.lerp_mixed) // border_top <= 0 <= border_bottom (or the reverse): t * b + (1 - t) * a
        return q * b.border_bottom + (1.0f - q) * b.border_top;
    return lerp_std(b.border_top, b.border_bottom, q);
}
// stores incl. the mirrored image (:1559-1564, :1419-1424): outputs above the middle repeat the lower ones
WF_DEV void put_output(const BarArgs &b, float *row, int o, float y)
{
    row[o] = y;
    WF_UNROLL
    for(int j = 0; j < 8; ++j) // (constant indices: indexed by a run-time j the kernel's argument block went to scratch, 784 B per lane)
        if(j < b.out2_n) // uniform
            row[(long long)o + b.out2_delta[j]] = y;
}
WF_DEV void store_output(const BarArgs &b, int o, float y, float *out_row, float *dup_row)
{
    if(!b.mirror) {
        put_output(b, out_row, o, y);
        if(dup_row) put_output(b, dup_row, o, y);
        return;
    }
    const int half = b.num_bars / 2;
    const int img = 2 * half - o;
    const bool own = o <= half;
    const bool image = o < half && img > half && img < b.num_bars;
    if(o == half + 1 && b.pre_out != nullptr) { // (one lane per row; the caller has pointed pre_out at this row's entry: BarArgs::pre_out)
        b.pre_out[0] = y;
        if(dup_row) b.pre_out[1] = y; // (the second row of a single captured channel shown as stereo: the next entry)
    }
    if(own) {
        put_output(b, out_row, o, y);
        if(dup_row) put_output(b, dup_row, o, y);
    }
    if(image) {
        put_output(b, out_row, img, y);
        if(dup_row) put_output(b, dup_row, img, y);
    }
}
WF_DEV void emit_output(const BarArgs &b, int o, float v, float *out_row, float *dup_row)
{
    store_output(b, o, map_output(b, v), out_row, dup_row);
}

// the outputs (bars or curve points) thread t finishes: o = t + T*k, k < out_steps <= KMAX
template<class G> struct OutVals {
    static constexpr int KMAX = (G::T <= 64) ? 16 : 8; // 1024 outputs per row on one wavefront, 8 per thread otherwise
    float v[KMAX];
};

// curve points of this thread from the dB row parked in LDS.  Steps are taken four at a time: the (L2) table loads of a
// group are issued together; the tables are padded to whole groups with zero coefficients.
// (TT: threads that share the row -- G::T, or 2 * G::T when the threads of both spectra of a workgroup finish the one row a
// mono mixdown displays, BarArgs::both_subs)
template<class G, int TT = G::T> WF_DEV void curve_row(const BarArgs &b, bool has_row, const float *db, int t, OutVals<G> &ov)
{
    constexpr int T = TT, K = OutVals<G>::KMAX;
    static_assert(K % 4 == 0, "curve steps are processed in groups of four");
    WF_UNROLL
    for(int k = 0; k < K; ++k)
        ov.v[k] = 0.0f;
    WF_UNROLL
    for(int g = 0; g < K / 4; ++g) {
        if(4 * g < b.out_steps && has_row) { // the first condition is uniform
            f4 c0[4], c1[4];
            int base[4];
            WF_UNROLL
            for(int j = 0; j < 4; ++j) {
                const int o = (4 * g + j) * T + t;
                c0[j] = ld4(b.cur_coef + 8 * o);
                c1[j] = ld4(b.cur_coef + 8 * o + 4);
                base[j] = b.cur_base[o];
            }
            WF_UNROLL
            for(int j = 0; j < 4; ++j) {
                const float *p = db + base[j];
                // kernel_convolve (reference src/filter.hpp:160-169): sum += samples[i] * weight, ascending taps
                float sum = p[0] * c0[j].x;
                sum = fmaf(p[1], c0[j].y, sum);
                sum = fmaf(p[2], c0[j].z, sum);
                sum = fmaf(p[3], c0[j].w, sum);
                sum = fmaf(p[4], c1[j].x, sum);
                sum = fmaf(p[5], c1[j].y, sum);
                sum = fmaf(p[6], c1[j].z, sum);
                sum = fmaf(p[7], c1[j].w, sum);
                ov.v[4 * g + j] = sum;
            }
        }
    }
}

// make_catrom_kernel's weights for u = x - floor(x), tension 0.5, evaluated as the reference does: sum += row[k] * matrix[j][k]
// over row = {1, u, u*u, u*u*u}, separately rounded products and sums (the zero matrix entries add exact zeros)
struct CatromW { float w0, w1, w2, w3; };
WF_DEV CatromW catrom_weights(float u)
{
#pragma clang fp contract(off) // every product and sum rounded on its own, as the reference's table was computed
    const float u2 = u * u, u3 = u2 * u;
    CatromW c;
    c.w0 = ((u * -0.5f) + u2) + (u3 * -0.5f);          // {0, -t, 2t, -t}
    c.w1 = (1.0f + (u2 * -2.5f)) + (u3 * 1.5f);        // {1, 0, t-3, 2-t}
    c.w2 = ((u * 0.5f) + (u2 * 2.0f)) + (u3 * -1.5f);  // {0, t, 3-2t, t-2}
    c.w3 = (u2 * -0.5f) + (u3 * 0.5f);                 // {0, 0, -t, t}
    return c;
}
// one Catmull-Rom point: kernel_convolve (reference src/filter.hpp:160-169) over taps floor(x) - 1 .. floor(x) + 2.  The
// positions lie in [1, M - 1] (init_interp clamps them to [lowbin, highbin]), so only the taps at M and M + 1 can fall
// outside the row: the caller parks two zeros there (a dropped tap adds an exact 0).
WF_DEV float catrom_point(const float *db, float x)
{
#pragma clang fp contract(off)
    const float fl = __builtin_floorf(x);
    const CatromW c = catrom_weights(x - fl);
    const float *p = db + ((int)fl - 1);
    float sum = p[0] * c.w0;
    sum = sum + (p[1] * c.w1);
    sum = sum + (p[2] * c.w2);
    sum = sum + (p[3] * c.w3);
    return sum;
}
template<class G, int TT = G::T> WF_DEV void curve_row_catrom(const BarArgs &b, bool has_row, const float *db, int t, OutVals<G> &ov)
{
    constexpr int T = TT, K = OutVals<G>::KMAX;
    WF_UNROLL
    for(int k = 0; k < K; ++k)
        ov.v[k] = 0.0f;
    // (requesting the positions before the dB math, as the bars' entries are, measured 8 % slower: the loads queue ahead
    // of the row stores)
    WF_UNROLL
    for(int g = 0; g < K / 4; ++g) {
        if(4 * g < b.out_steps && has_row) { // the first condition is uniform
            float x[4];
            WF_UNROLL
            for(int j = 0; j < 4; ++j)
                x[j] = b.cur_x[(4 * g + j) * T + t];
            WF_UNROLL
            for(int j = 0; j < 4; ++j)
                ov.v[4 * g + j] = catrom_point(db, x[j]);
        }
    }
}

// Curves wider than KMAX points per thread: a run-time loop over the steps; every point is mapped and stored as soon as it
// is produced, or -- with the Gaussian filter on -- staged behind the dB row ([pad | points | pad | weights] from
// db + stage_off, sized on the host) and filtered in a second loop.
template<class G, int TT = G::T, class Sync>
WF_DEV void curve_row_stream(const BarArgs &b, bool has_row, const float *db, float *lds, int t, float *out_row, float *dup_row, Sync sync)
{
    constexpr int T = TT;
    const int n = b.num_bars;
    const bool filtered = b.gauss_radius > 0;
    const int pad = b.gauss_radius - 1, size = 2 * b.gauss_radius - 1;
    float *vp = lds + b.stage_off;
    float *wl = vp + n + 2 * pad;
    if(filtered && has_row) {
        for(int i = t; i < pad; i += T) {
            vp[i] = 0.0f;
            vp[pad + n + i] = 0.0f;
        }
        for(int i = t; i < size; i += T)
            wl[i] = b.gauss[i];
    }
    if(has_row) {
        for(int k = 0; k < b.out_steps; ++k) {
            const int o = k * T + t;
            float v;
            if(b.curve == 2) {
                v = catrom_point(db, b.cur_x[o]);
            } else {
                const f4 c0 = ld4(b.cur_coef + 8 * o), c1 = ld4(b.cur_coef + 8 * o + 4);
                const float *p = db + b.cur_base[o];
                v = p[0] * c0.x;
                v = fmaf(p[1], c0.y, v);
                v = fmaf(p[2], c0.z, v);
                v = fmaf(p[3], c0.w, v);
                v = fmaf(p[4], c1.x, v);
                v = fmaf(p[5], c1.y, v);
                v = fmaf(p[6], c1.z, v);
                v = fmaf(p[7], c1.w, v);
            }
            if(o < n) {
                if(filtered)
                    vp[pad + o] = v;
                else
                    emit_output(b, o, v, out_row, dup_row);
            }
        }
    }
    if(filtered) {
        sync();
        if(has_row) {
            for(int o = t; o < n; o += T) {
                float sum = 0.0f;
                for(int tap = 0; tap < size; ++tap)
                    sum = fmaf(vp[o + tap], wl[tap], sum);
                emit_output(b, o, sum / b.gauss_wsum[o], out_row, dup_row);
            }
        }
    }
}

// optional Gaussian filter across the row's outputs, then mapping + stores.  `lds` is the spectrum's LDS as floats; it may
// alias everything the row used before (the dB row, the partials) and is written only after a barrier.
// weighted_avg (reference src/filter.hpp:133-157) divides the tap sum by the sum of the weights whose taps fall inside the
// row.  Here the row is staged with radius-1 zeros on either side (a dropped tap adds an exact 0), the weights sit next
// to it, and the divisor of every output comes from a table the host accumulated in the reference's order -- so the
// inner loop is one LDS read and one FMA per tap, no bounds logic.
template<class G, int TT = G::T, class Sync>
WF_DEV void outputs_finish(const BarArgs &b, bool has_row, OutVals<G> &ov, float *lds, int t, float *out_row, float *dup_row, Sync sync)
{
    constexpr int T = TT, K = OutVals<G>::KMAX;
    static_assert(K % 4 == 0, "outputs are filtered in groups of four");
    const int n = b.num_bars;
    if(b.gauss_radius > 0) {
        const int pad = b.gauss_radius - 1, size = 2 * b.gauss_radius - 1;
        float *vp = lds;               // [pad | n | pad]
        float *wl = lds + n + 2 * pad; // [size]
        sync();
        // only the spectrum that has a row stages anything: in mono display channel 1's buffer is what channel 0 reads the
        // partner's magnitudes from, and on the one-wavefront geometries nothing orders that read before this point
        if(has_row) {
            for(int i = t; i < pad; i += T) {
                vp[i] = 0.0f;
                vp[pad + n + i] = 0.0f;
            }
            for(int i = t; i < size; i += T)
                wl[i] = b.gauss[i];
        }
        if(has_row) {
            WF_UNROLL
            for(int k = 0; k < K; ++k)
                if(k < b.out_steps && k * T + t < n)
                    vp[pad + k * T + t] = ov.v[k];
        }
        sync();
        if(has_row) {
            WF_UNROLL
            for(int g = 0; g < K / 4; ++g) {
                if(4 * g < b.out_steps) { // uniform
                    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    int o[4];
                    float div[4];
                    WF_UNROLL
                    for(int j = 0; j < 4; ++j) {
                        o[j] = (4 * g + j) * T + t;
                        if(o[j] >= n)
                            o[j] = n - 1; // lanes past the end redo the last output; their result is not stored
                        div[j] = b.gauss_wsum[o[j]];
                    }
                    for(int tap = 0; tap < size; ++tap) {
                        const float w = wl[tap];
                        WF_UNROLL
                        for(int j = 0; j < 4; ++j)
                            sum[j] = fmaf(vp[o[j] + tap], w, sum[j]);
                    }
                    WF_UNROLL
                    for(int j = 0; j < 4; ++j)
                        ov.v[4 * g + j] = sum[j] / div[j];
                }
            }
        }
    }
    if(has_row) {
        float y[K];
        WF_UNROLL
        for(int k = 0; k < K; ++k)
            y[k] = map_output(b, ov.v[k]); // all of them, branch-free; the ones past the row are not stored
        WF_UNROLL
        for(int k = 0; k < K; ++k)
            if(k < b.out_steps && k * T + t < n)
                store_output(b, k * T + t, y[k], out_row, dup_row);
    }
}

// ---- bars, prefix-sum layout (BarPsTables, wf_host_tables.hpp) ---------------------------------------------------------------
// The tick of a bars display is as much bound by the vector units as by memory (cut-point builds, profiles/r05b_ps_cuts*.txt: 100
// more instructions per wavefront in the tail cost what they add to its ~1200, 8 %), so the layout is built around instruction
// count: every wavefront only parks its part of the row and the sums of its four-bin groups (12 additions, five 16-byte LDS
// stores); ONE wavefront finishes -- for both spectra of a workgroup where the sub-bands fit 32 lanes each -- and it alone forms
// the prefix sum, over 16-bin quads, in float64.
// A spectrum's LDS in this layout, as floats from `dbl`: [4 zeros | the row's M dB values | 4 zeros] (bins -4 .. M + 3: the taps
// kernel_convolve drops read an exact 0), then gs[M / 4] = the four-bin group sums, thread-major (group g = bin / 4 = t + T u of
// thread t, register group u, sits at t (P / 4) + u: one 16-byte store per thread), then qp[M / 16 + 1] doubles: the sum of all
// quads in front of quad Q (the last entry: the row's total).
constexpr size_t ps_lds_floats(size_t M) { return 8 + M + M / 4 + 2 * (M / 16 + 2); } // (constexpr: host and device)
WF_DEV float *ps_gs_area(float *dbl, int M) { return dbl + 8 + M; }
WF_DEV double *ps_qp_area(float *dbl, int M) { return reinterpret_cast<double *>(dbl + 8 + M + M / 4); }
template<class RG> WF_DEV void ps_park(float *dbl, int M, int t, const float (&d)[RG::P])
{
    constexpr int NG = RG::P / 4;
    static_assert(NG == 1 || NG == 2 || NG % 4 == 0, "group sums are stored as 4-, 8- or 16-byte words");
    store_row<RG>(dbl + 4, t, d);
    float *gs = ps_gs_area(dbl, M) + t * NG;
    float s[NG];
    WF_UNROLL
    for(int u = 0; u < NG; ++u)
        s[u] = (d[4 * u] + d[4 * u + 1]) + (d[4 * u + 2] + d[4 * u + 3]);
    if constexpr(NG == 1)
        gs[0] = s[0];
    else if constexpr(NG == 2)
        *reinterpret_cast<f2 *>(gs) = f2{s[0], s[1]};
    else {
        WF_UNROLL
        for(int u = 0; u < NG; u += 4)
            st4(gs + u, f4{s[u], s[u + 1], s[u + 2], s[u + 3]});
    }
    if(t == 0) {
        st4(dbl, f4{0.0f, 0.0f, 0.0f, 0.0f});
        st4(dbl + 4 + M, f4{0.0f, 0.0f, 0.0f, 0.0f});
    }
}
// Inclusive prefix of a double over the 64 lanes of a wavefront: the classic six DPP steps (row_shr 1, 2, 4, 8 inside rows of 16,
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) on the two halves of the value; a lane without a source adds +0.0
WF_DEV double wave_scan_f64(double v)
{
#if defined(__HIPCC__)
#define WF_SCAN64_STEP(CTRL, ROWS)                                                                                         \
    {                                                                                                                      \
        const long long b = __double_as_longlong(v);                                                                       \
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROWS, 0xf, true);                     \
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWS, 0xf, true);                              \
        v += __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);                                    \
    }
    WF_SCAN64_STEP(0x111, 0xf)
    WF_SCAN64_STEP(0x112, 0xf)
    WF_SCAN64_STEP(0x114, 0xf)
    WF_SCAN64_STEP(0x118, 0xf)
    WF_SCAN64_STEP(0x142, 0xa)
    WF_SCAN64_STEP(0x143, 0xc)
#undef WF_SCAN64_STEP
#endif
    return v;
}
// the value of lane l ^ 1 (quad_perm:[1,0,3,2])
WF_DEV int lane_swap1(int v)
{
#if defined(__HIPCC__)
    return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
#else
    return v;
#endif
}
// The finishing wavefront of a spectrum: lane `ll` (i) adds the group sums of its share of the quads and leaves their float64
// prefix in LDS, (ii) evaluates one edge of sub-band ll / 2 -- the look-up at its position and the dot product of its window --;
// the edges meet on the even lane, the sub-bands of a bar by the segmented scan, and the lane that finishes a bar maps and stores
// it.  (`emit` false: the spectrum produced no row -- everything is computed on whatever the buffer holds and nothing is stored;
// the DPP steps must not sit under a divergent branch.)
template<class G, class RG> WF_DEV void ps_finish(const BarArgs &b, const BarEntries<G> &be, float *dbl, int M, int ll, bool emit, float *out_row, float *dup_row)
{
    constexpr int T = RG::T, NG = RG::P / 4;
    const float *gs = ps_gs_area(dbl, M);
    double *qp = ps_qp_area(dbl, M);
    // group g = t + T u is stored at t NG + u; the four groups of quad Q are four consecutive threads of one register group
    auto quad_groups = [&](int Q) { const int g0 = 4 * Q; return gs + (g0 & (T - 1)) * NG + g0 / T; };
    constexpr int QUADS = G::M / 16, NQ = (QUADS + 63) / 64;
    double run = 0.0, ex[NQ];
    WF_UNROLL
    for(int k = 0; k < NQ; ++k) {
        const int Q = ll * NQ + k;
        const float *p = quad_groups(Q < QUADS ? Q : 0);
        const float qt = (p[0] + p[NG]) + (p[2 * NG] + p[3 * NG]);
        ex[k] = run;
        run += (double)((QUADS % 64 == 0 || Q < QUADS) ? qt : 0.0f);
    }
    const double inc = wave_scan_f64(run), base = inc - run;
    WF_UNROLL
    for(int k = 0; k < NQ; ++k) {
        const int Q = ll * NQ + k;
        if(QUADS % 64 == 0 || Q < QUADS)
            qp[Q] = base + ex[k];
    }
    if(ll == 63)
        qp[QUADS] = inc;
    wave_fence(); // (LDS operations of a wavefront execute in order: the look-ups below see the prefix)
    const int q = (int)f32_bits(be.coef[2].x);
    const uint32_t info = f32_bits(be.coef[2].y);
    const float *pw = dbl + q + 1; // bins q - 3 .. q + 3 of the row parked from dbl + 4
    float w[7];
    WF_UNROLL
    for(int j = 0; j < 7; ++j)
        w[j] = pw[j];
    // PS[min(q + 4, M)]: the quads in front (float64), the groups of the quad in front, the bins of the group in front (the top
    // three of the window; a clamped position is a multiple of 16)
    const int x = q + 4 < M ? q + 4 : M;
    const int ng = (x >> 2) & 3, nb = x & 3;
    const float *pg = quad_groups(x >> 4);
    double f = qp[x >> 4];
    f += (double)(ng >= 1 ? pg[0] : 0.0f);
    f += (double)(ng >= 2 ? pg[NG] : 0.0f);
    f += (double)(ng >= 3 ? pg[2 * NG] : 0.0f);
    f += (double)(nb >= 1 ? w[6] : 0.0f);
    f += (double)(nb >= 2 ? w[5] : 0.0f);
    f += (double)(nb >= 3 ? w[4] : 0.0f);
    float e0 = w[0] * be.coef[0].x, e1 = w[1] * be.coef[0].y;
    e0 = fmaf(w[2], be.coef[0].z, e0);
    e1 = fmaf(w[3], be.coef[0].w, e1);
    e0 = fmaf(w[4], be.coef[1].x, e0);
    e1 = fmaf(w[5], be.coef[1].y, e1);
    e0 = fmaf(w[6], be.coef[1].z, e0);
    const float e = e0 + e1;
    // the high edge's prefix and window sum come over from the odd lane
    const long long fb = __builtin_bit_cast(long long, f);
    const long long ob = ((long long)lane_swap1((int)(fb >> 32)) << 32) | (long long)(unsigned int)lane_swap1((int)(fb & 0xffffffffll));
    const float oe = __builtin_bit_cast(float, lane_swap1(__builtin_bit_cast(int, e)));
    const double dp = __builtin_bit_cast(double, ob) - f;
    float sub = (ll & 1) ? 0.0f : fmaf(be.coef[1].w, (float)dp, e + oe);
    sub = seg_prefix_scan(sub, info);
    const int bar = (int)((info >> 8) & 0xffu) - 1;
    if(emit && bar >= 0)
        emit_output(b, bar, sub / (float)(info >> 16), out_row, dup_row);
}

// Called by every thread of the workgroup (sync may be a block barrier); `has_row` says whether this spectrum produced one.
// Returns true when the bars were left in `ov` for outputs_finish (one bar per thread, k = 0), false when they have already
// been mapped and stored (chunked path; no filter there).
// arrivals / arrive_last (piece mode, several wavefronts per spectrum): the spectrum's arrival counter in LDS and the value it
// reads once every wavefront of the spectrum has counted itself in for the last time
// PIECES == false: the instantiation never runs the wave-private layout (the Bluestein kernels, at their register cap: the host
// does not build it for them)
template<class G, bool PIECES = true, class Sync, class XorSum>
WF_DEV bool bars_reduce_row(const BarArgs &b, const BarPre &pre, const BarEntries<G> &be, bool has_row, float *db, float *prod, int t,
                            float *out_row, float *dup_row, OutVals<G> &ov, Sync sync, XorSum xor_sum, int *arrivals = nullptr, int arrive_last = 0)
{
    constexpr int T = G::T;
    auto emit = [&](int bar, float sum, int cnt) { emit_output(b, bar, sum / (float)cnt, out_row, dup_row); };
    if(b.num_segs > 0) {
        // A: every thread forms the dot product of its own segment (padded bins have coefficient 0) -- four
        // independent partial sums in entry order
        float part = 0.0f;
        if(has_row) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            const float *p = db + be.base;
            WF_UNROLL
            for(int cc = 0; cc < BarEntries<G>::CMAX; ++cc) {
                if(cc < b.lane_blocks) { // uniform
                    const f4 w = be.coef[cc];
                    const f4 v = *reinterpret_cast<const f4 *>(p + 4 * cc);
                    a0 = fmaf(v.x, w.x, a0);
                    a1 = fmaf(v.y, w.y, a1);
                    a2 = fmaf(v.z, w.z, a2);
                    a3 = fmaf(v.w, w.w, a3);
                }
            }
            part = (a0 + a1) + (a2 + a3);
        }
        if(PIECES && b.piece_mode) {
            // Every segment read bins its own wavefront parked (LDS operations of a wave execute in order: no barrier in front
            // of them).  The partials of a piece are added over the lanes; its last lane holds the total.
            WF_BAR_STAMP(14);
            const uint32_t info = (uint32_t)pre.glen;
            part = seg_prefix_scan(part, info);
            const int slot = (int)(info >> 8) - 1;
            WF_BAR_STAMP(15);
            if constexpr(T == 64) {
                if(has_row && slot >= 0)
                    emit(slot, part, pre.count); // one wavefront: a bar is one piece
            } else {
                // the totals meet in LDS behind the row; a wavefront that has left its own counts itself in, and the one that
                // finds everybody else's count adds the pieces of every bar in slot order -- nobody waits for anybody
                if(has_row && slot >= 0)
                    prod[slot] = part;
                const int before = wave_arrive(arrivals, t & 63);
                if(before == arrive_last - 1 && has_row && pre.lead >= 0) {
                    float sum = 0.0f;
                    for(int k = pre.s0; k < pre.s1; ++k)
                        sum += prod[k];
                    emit(pre.lead, sum, pre.count);
                }
            }
            return false;
        }
        if(b.wave_local) {
            // every bar lives inside one wavefront: its partials are added by a segmented reduction over the lanes (six
            // shuffles; lane s ends up with the sum of segments [s, end of its bar)) -- no LDS round trip, no barrier
            WF_BAR_STAMP(14);
            WF_UNROLL
            for(int d = 1; d < 64; d *= 2) {
                const float o = wave_shfl_down(part, d);
                if(t + d < pre.s1)
                    part += o;
            }
            WF_BAR_STAMP(15);
            if(has_row && pre.lead >= 0)
                emit(pre.lead, part, pre.count);
            return false;
        }
        if(has_row)
            prod[t] = part; // parked behind the dB row
        sync();
        WF_BAR_STAMP(14);
        WF_BAR_STAMP(15);
        // B1: the partials of a bar are added in two levels -- groups of up to eight consecutive segments first (their
        // leaders read the eight values in one go), then one thread per bar adds its group sums in order.  A log-axis display
        // has bars of a few hundred bins = dozens of segments; one thread walking them alone was the longest chain of the phase.
        float *gsum = prod + T + 8; // behind the partials and the eight floats a leader may read past them
        if(has_row && pre.glen > 0) {
            float q[8];
            WF_UNROLL
            for(int j = 0; j < 8; ++j)
                q[j] = prod[t + j]; // inside the scratch for every t; entries past the group are not used
            float a0 = q[0], a1 = 0.0f;
            WF_UNROLL
            for(int j = 1; j < 8; ++j) {
                const float v = (j < pre.glen) ? q[j] : 0.0f;
                if(j & 1) a1 += v; else a0 += v;
            }
            gsum[t] = a0 + a1;
        }
        sync();
        // B2: one thread per bar
        WF_UNROLL
        for(int k = 0; k < OutVals<G>::KMAX; ++k)
            ov.v[k] = 0.0f;
        if(has_row && t < b.num_bars) {
            float a0 = 0.0f, a1 = 0.0f;
            int k = pre.s0;
            for(; k + 8 < pre.s1; k += 16) {
                a0 += gsum[k];
                a1 += gsum[k + 8];
            }
            if(k < pre.s1)
                a0 += gsum[k];
            ov.v[0] = (a0 + a1) / (float)pre.count;
        }
        return true;
    }
    // ---- tables larger than the scratch (very many bars): chunk by chunk, lanes_per_bar threads per bar ---------------
    const int lpb = b.lanes_per_bar;
    const int bars_per_pass = T / lpb;
    // With the Gaussian filter the bar means are parked in a staging area behind the product scratch (the host sized the
    // chunks around it) instead of being mapped at once; the filter runs when the last chunk is done.
    const bool filtered = b.gauss_radius > 0;
    const int gpad = b.gauss_radius - 1, gsize = 2 * b.gauss_radius - 1;
    float *vp = db + b.stage_off;                   // [gpad | num_bars | gpad]
    float *wl = vp + b.num_bars + 2 * gpad;         // [gsize]
    if(filtered && has_row) {
        for(int i = t; i < gpad; i += T) {
            vp[i] = 0.0f;
            vp[gpad + b.num_bars + i] = 0.0f;
        }
        for(int i = t; i < gsize; i += T)
            wl[i] = b.gauss[i];
    }
    for(int c = 0; c < b.num_chunks; ++c) {
        const bool single = (b.num_chunks == 1);
        const int bar_lo = single ? 0 : b.chunk[c], bar_hi = single ? b.num_bars : b.chunk[c + 1];
        const int e_lo = single ? 0 : b.off[bar_lo], e_hi = single ? b.entries : b.off[bar_hi];
        if(has_row) {
            for(int e = e_lo + t; e < e_hi; e += T)
                prod[e - e_lo] = db[b.bin[e]] * b.coef[e];
        }
        sync();
        for(int b0 = bar_lo; b0 < bar_hi; b0 += bars_per_pass) {
            const int bar = b0 + t / lpb;
            const int sub = t % lpb;
            const bool live = has_row && bar < bar_hi;
            const bool first_pass = (c == 0 && b0 == 0);
            float acc = 0.0f;
            int cnt = 1;
            if(live) {
                const int boff = first_pass ? pre.off : b.off[bar];
                const int n = first_pass ? pre.len : b.off[bar + 1] - boff;
                cnt = first_pass ? pre.count : b.count[bar];
                const int o = boff - e_lo;
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                int k = sub;
                for(; k + 3 * lpb < n; k += 4 * lpb) {
                    a0 += prod[o + k];
                    a1 += prod[o + k + lpb];
                    a2 += prod[o + k + 2 * lpb];
                    a3 += prod[o + k + 3 * lpb];
                }
                for(; k < n; k += lpb)
                    a0 += prod[o + k];
                acc = (a0 + a1) + (a2 + a3);
            }
            for(int m = lpb >> 1; m >= 1; m >>= 1)
                acc = xor_sum(acc, m);
            if(live && sub == 0) {
                if(filtered)
                    vp[gpad + bar] = acc / (float)cnt;
                else
                    emit(bar, acc, cnt);
            }
        }
        if(c + 1 < b.num_chunks)
            sync(); // prod is reused by the next chunk
    }
    if(filtered) {
        // apply_filter / weighted_avg (reference src/filter.hpp:133-157, :171-180) as in outputs_finish, one output per
        // thread and round
        sync();
        if(has_row) {
            for(int o = t; o < b.num_bars; o += T) {
                float sum = 0.0f;
                for(int tap = 0; tap < gsize; ++tap)
                    sum = fmaf(vp[o + tap], wl[tap], sum);
                emit_output(b, o, sum / b.gauss_wsum[o], out_row, dup_row);
            }
        }
    }
    return false;
}

} // namespace wf
