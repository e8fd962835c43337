// wf_host_tables.hpp -- host-side, per-configuration precompute for the spectrum path.
//
// Everything WAVSource::update() derives once per settings change and the hot loop then
// only reads (reference src/source.cpp:1169-1290, init_interp :837-896, init_rolloff
// :898-918, get_gravity src/source.hpp:301-312).  Computed on the host in float with the
// reference's expression order so the tables are bit-identical to the reference's, then
// uploaded once; the FFT twiddles are computed in double and rounded once, as FFTW does
// for its single-precision build (deps/fftw-3.3.11/kernel/trig.c:48-80).
#pragma once
#include "wf_config.h"
#include <cstdint>
#include <vector>

namespace wf {

struct cfloat { float re, im; };

struct HostTables {
    // spectrum
    std::vector<float> window;   // [N], empty when FFTWindow::NONE
    float window_sum = 1.0f;     // m_window_sum
    std::vector<float> slope;    // [M], empty when m_slope <= 0
    std::vector<float> rolloff;  // [M], empty unless rolloff_q > 0 && rolloff_rate > 0
    uint32_t output_channels = 1; // m_output_channels
    uint32_t display_channels = 1; // m_stereo ? 2 : 1
    // bars, or the points of the curve (cfg.curve): "outputs" of the render-time reduction
    int num_bars = 0;                    // m_num_bars, or m_width in curve mode
    std::vector<float> interp_indices;   // m_interp_indices after init_interp()
    int bar_rows_bins = 0;               // bins per row the bar tables index (fft_size / 2)
    std::vector<int> band_widths;        // m_band_widths
    std::vector<float> interp_weights;   // m_interp_kernel.weights (Lanczos: 8/sample, Catmull-Rom: 4/sample)
    int interp_radius = 0;               // m_interp_kernel.radius
    int interp_taps = 0;                 // m_interp_kernel.size
    float border_top = 0.0f, border_bottom = 0.0f, cpos = 0.0f; // render_bars geometry (:1480-1494)
    // device form of the bar reduction: one dot product per bar over a contiguous bin range (see BarArgs)
    std::vector<float> bar_coef;   // [entries]
    std::vector<int> bar_bin;      // [entries] the bin each coefficient multiplies
    std::vector<int> bar_off;      // [num_bars + 1]
    // Gaussian filter across the outputs (make_gauss_kernel, src/filter.hpp:40-65); empty when filter_mode is NONE
    std::vector<float> gauss;      // [2 * gauss_radius - 1]
    std::vector<float> gauss_wsum; // [num_bars] weighted_avg's divisor per output: the weights of the taps inside the row,
                                   // accumulated in tap order (== gauss_sum away from the edges)
    int gauss_radius = 0;
    float gauss_sum = 0.0f;
};

// get_settings()' repairs of out-of-range combinations (src/source.cpp:567-579): cutoffs swapped, ceiling <= floor, channel
// spacing in mono display or larger than the height.  Call first.
void normalize_config(wf_config &cfg);
// Level-meter mode (cfg.meter): applies update()'s overrides for the mode to `cfg` (src/source.cpp:1106-1128) and
// returns the meter buffer length, which replaces cfg.fft_size.  Call before build_host_tables.
uint32_t meter_config(wf_config &cfg);
// Waveform display mode (cfg.waveform): update()'s overrides (src/source.cpp:1130-1142) applied to `cfg`; fft_size becomes
// the row length (m_width); returns m_waveform_samples.  Call before build_host_tables.
uint32_t waveform_config(wf_config &cfg);
// returns 0 on success, a negative wf_hip error code otherwise
int build_host_tables(const wf_config &cfg, HostTables &out);
// bar ranges whose entries fit `cap_floats` of LDS scratch together (every single bar fits: len <= M + 7 <= cap)
std::vector<int> bar_chunks(const HostTables &t, size_t cap_floats);

// Device form of the bar tables when every thread of a spectrum can own one segment: the entries cut into at most
// `threads` segments of near-equal length (a multiple of 4, at most 4 * max_blocks) that never straddle a bar; bar b owns
// segments [bar_seg[b], bar_seg[b+1]); coefficient/bin tables re-laid lane-major and zero-padded.  false if the bars
// outnumber the threads or the segments would be longer than the kernel holds in registers.
struct BarLaneTables {
    std::vector<float> coef;  // [blocks][threads][4] lane-major: block c of segment s at [(c * threads + s) * 4, +4)
    std::vector<int> base;    // [threads] first bin of the segment's 4 * blocks consecutive bins, a multiple of 4
    std::vector<int> bar_seg; // [num_bars + 1]
    std::vector<int> seg_group; // [threads] > 0 where a group of that many (<= 8) consecutive segments of one bar starts
    // wave-local layout (all segments of a bar inside one wavefront, so that the partial sums are added by lane shuffles: no
    // LDS, no barrier): lead_bar[s] = the bar whose first segment s is (else -1), lead_end[s] = one past the last segment of
    // the bar segment s belongs to (0: unused thread)
    std::vector<int> lead_bar, lead_end;
    bool wave_local = false;
    int num_segs = 0, blocks = 0;
};
// wave_local: try the layout above first (needs every bar to fit 64 segments and the padded total to fit the threads)
bool bar_segments(const HostTables &t, int threads, int max_blocks, BarLaneTables &out, bool wave_local = false);

// Wave-private form of the same tables (round 4).  Thread t of a spectrum of `threads` threads with `points` bins each holds
// bins 4 (t + threads * u) .. + 3, u < points / 4, in registers when the row is finished, i.e. wavefront w = t / 64 owns the
// 256-bin chunks c = w + (threads / 64) * u.  A bar is cut into PIECES where the owner of its bins changes (never, on one
// wavefront); the segments of a piece are lanes of the wavefront that owns its bins and read only bins of that wavefront's
// chunks -- so a wavefront parks its part of the row in LDS and reads it back without waiting for anyone.  The piece's
// partial sums are added by a segmented prefix scan over the lanes (six DPP steps, flags in `info`), its last lane leaves the
// total in slot `piece index` of an LDS array, and whichever wavefront arrives last adds the slots of every bar (bar b on
// lane b: needs num_bars <= 64) in a fixed order.  false when it does not fit (more than 64 segments in one wavefront, more
// than 64 bars): bar_segments' layouts remain.
struct BarPieceTables {
    std::vector<float> coef;    // [blocks][threads][4] lane-major, as BarLaneTables::coef
    std::vector<int> base;      // [threads] first of the segment's 4 * blocks consecutive bins (inside one chunk of the thread's wavefront)
    std::vector<int> info;      // [threads] bits 0..5: the scan's steps this lane takes (1, 2, 4, 8 lanes down inside its row of 16; the
                                // last lane of the previous row; lane 31); bits 8..: 1 + slot when the lane is the last of a piece
                                // (one wavefront per spectrum: slot = the bar itself)
    std::vector<int> bar_piece; // [num_bars + 1] bar b owns slots [bar_piece[b], bar_piece[b + 1])
    int num_slots = 0, blocks = 0, num_segs = 0;
};
bool bar_pieces(const HostTables &t, int threads, int points, int max_blocks, BarPieceTables &out);

// Prefix-sum form of the bar reduction (round 5).  Inside a band every sample sits on the bin after its predecessor and -- as
// init_interp forms the positions as idx[i] + j in float (src/source.cpp:876-884) -- the fractional part of the position, and
// with it the sample's weight row, only changes where the sum crosses into the next binade.  A band is therefore a handful of
// SUB-BANDS [lo, hi) of consecutive bins with ONE weight row W (8 taps at bins ix - 3 .. ix + 4; Catmull-Rom's four sit in the
// middle), found here by comparing the reference's own rows bit for bit, and
//     sum_{ix in [lo, hi)} sum_t W[t] dB[ix - 3 + t]  =  SW (PS[hi + 4] - PS[lo + 4]) + E(lo) - E(hi),
//     PS[x] = sum_{m < x} dB[m],   SW = sum_t W[t],   E(q) = sum_{j < 7} C[j] dB[q - 3 + j],   C[j] = sum_{t <= j} W[t]
// (bins outside the row count as 0, which is kernel_convolve's clipping, src/filter.hpp:160-169): two look-ups into a prefix
// sum of the row, kept in float64, and two 7-tap edge corrections instead of count x 8 products -- no per-bin coefficient table
// (46 KB at N = 4096, 186 KB at 16384), no per-thread table loads, no read-back of the whole row.  Sub-bands of at most 7 bins
// are evaluated directly instead (SW = 0, the two windows carry the composite coefficients of the bins they cover): no
// cancellation for the narrow bands at the bottom of a log axis.  The first wavefront of a spectrum finishes: TWO lanes per sub-band
// (at most 32 sub-bands) -- lane 2 j its low edge (window coefficients clo, position lo), lane 2 j + 1 its high edge (chi, hi) --,
// one look-up and one window each; the halves meet on the even lane, the sub-bands of a bar by seg_prefix_scan (flags in `info`).
// POINT mode (plain band means, src/source.cpp:1525-1532) is the same with W = a single 1.
struct BarPsTables {
    // [3][64][4] floats, lane-major 16-byte words: {c[0..3]} {c[4..6], SW} {position, info, 0, 0} (the integers stored as bit
    // patterns; chi carries its sign).  info: bits 0..5 seg_prefix_scan's steps; bits 8..15: 1 + bar on the lane that finishes
    // the bar (the low-edge lane of its last sub-band); bits 16..31: the bar's count
    std::vector<float> tab;
    int num_lanes = 0; // 64
    int num_subs = 0;
};
bool bar_ps(const HostTables &t, int threads, BarPsTables &out);

// Curve mode (one output per thread and step): output o = k * threads + s reads the 8 consecutive dB bins starting at
// base[o] with coefficients coef[o][0..8) (its composite kernel shifted/zero-padded to 8 taps inside [0, M)); tables are
// padded to whole steps with zero coefficients.
struct CurveLaneTables {
    std::vector<float> coef; // [steps rounded up to 4][threads][8]  (empty when `x` is used)
    std::vector<int> base;   // [steps rounded up to 4][threads]
    std::vector<float> x;    // [steps rounded up to 4][threads] Catmull-Rom: the points' positions (weights evaluated on the device)
    int steps = 0;
};
bool curve_lanes(const HostTables &t, const wf_config &cfg, int threads, int max_steps, CurveLaneTables &out);

// FFT sizes that are not powers of two: Bluestein's algorithm over the complex FFT core.
//   Y_k = conj(w_k) * sum_j (y_j conj(w_j)) w_(k-j),  w_m = exp(i pi m^2 / n')   (jk = (j^2 + k^2 - (k-j)^2) / 2)
// i.e. a circular convolution a * b of length L with a_j = y_j conj(w_j), b_m = w_m, computed as IFFT_L(FFT_L(a) . FFT_L(b)).
// Two forms:
//  * packed (n <= 16384: inside one workgroup).  The n real samples are the n' = n/2 complex points
//    z_j = window_2j x_2j + i window_2j+1 x_2j+1; their n'-point DFT Z needs L >= 2n' - 1 = n - 1 (every k < n' is wanted),
//    and X_k follows from Z_k and Z_(n'-k) by the real split with W_n^k -- half the transform length of the direct form.
//    Tables: a[2j], a[2j+1] = the two complex factors of x_2j and x_2j+1 in a_j; b = FFT_L of the chirp;
//    q_k = conj(w_k) / L (Z_k = q_k * conj(R_k) for the twice-transformed R); qr_k = q_(n'-k mod n'); w[k] = W_n^k.
//  * direct (n > 16384: through device memory, wf_big.hpp).  y = window * x, n' = n, only k < n/2 wanted: L >= 3n/2,
//    |X_k| = |(a * b)_k|.  Tables: a_j = window_j conj(w_j), b.
// All in double, rounded once.
struct BluesteinTables {
    uint32_t L = 0;            // complex transform length (a power of two >= 512)
    bool packed = false;
    std::vector<cfloat> a;     // packed: [2L] (two factors per point, zero from point n/2 on); direct: [L], zero from n on
    std::vector<cfloat> b;     // [L] FFT_L of the chirp
    std::vector<cfloat> q, qr, w; // packed only: [n/2] each
};
// 0 when n is a power of two (no Bluestein needed)
uint32_t bluestein_length(uint32_t n);
// transforms of L = rows * 16384 complex points done in two steps (wf_big.hpp): tw_big[k1][n2] = W_L^(n2 k1); tws_big[k] =
// W_real_n^k, the real-split twiddles of a packed real_n-sample transform (real_n == 0: not built)
void build_big_twiddles(uint32_t L, uint32_t rows, uint32_t real_n, std::vector<cfloat> &tw_big, std::vector<cfloat> &tws_big);
void build_bluestein(const wf_config &cfg, const HostTables &t, BluesteinTables &out);
// FFT sizes above 16384 whose n/2 has a prime factor no mixed-radix plan takes: n/2 = C R points as C rows of R (decimation in
// frequency over the C columns, folded into the rows' fetch like big_mr_rows_kernel's), every row transformed by Bluestein INSIDE a
// workgroup's LDS (wf_big.hpp: big_br_rows_kernel) -- convolution length L = the power of two >= 2 R - 1 (returned; >= 4096).
// rowtw[k1][n2] = W_(n/2)^(n2 k1) conj(w_n2), the column twiddle and the chirp of row k1's point n2 in one factor;
// bhat = FFT_L of the chirp w_m = exp(i pi m^2 / R) (lags -(R-1) .. R-1, wrapped); q[k] = conj(w_k) / L: Z_row[k] = q[k] conj(R_k).
uint32_t build_bluestein_rows(uint32_t np, uint32_t C, std::vector<cfloat> &rowtw, std::vector<cfloat> &bhat, std::vector<cfloat> &q);

// FFT sizes with no prime factor above 23 and at most one of 17, 19, 23, 20, 25 (wf_mixed.hpp): the n/2-point transform as two to four mixed-radix passes instead of
// Bluestein.  plan_mixed_radix fills radix[] in pass order and returns the number of passes, 0 when np has another prime
// factor or no ordering fits: radices above 16 only in the first pass (the twiddled passes hold 2 (R - 1) more registers), and
// the last pass has one butterfly per thread at most (np / radix[last] <= threads).
int plan_mixed_radix(uint32_t np, uint32_t threads, int radix[4], uint64_t bluestein_points = 0);
// The exchange buffer of a mixed-radix spectrum of np complex points (MrPlan::half / s3 / lds_cf, wf_tick_phases.hpp): two halves of
// `half` = np rounded up to 16 points; never less than 576 complex units (the display phase parks the dB row and stages its
// inputs there: Geom::LDS_CF's floor) and never more than the container geometry's buffer (np <= M / 2).
inline uint32_t mr_exchange_half(uint32_t np) { return (np + 15u) & ~15u; }
inline uint32_t mr_exchange_cf(uint32_t np, uint32_t container_lds_cf)
{
    const uint32_t want = 2u * mr_exchange_half(np);
    return want < 576u ? (576u < container_lds_cf ? 576u : container_lds_cf) : (want < container_lds_cf ? want : container_lds_cf);
}
// the radices the one-wavefront container's small instantiation carries (spectrum_tick_kernel<.., MRS>): without the in-register
// DFTs of 13, 15, 16 and the first-pass radices its threads get by with 96 registers -- a fifth wave per SIMD
inline bool mr_small_radices(const int *radix, int passes)
{
    for(int i = 0; i < passes; ++i)
        switch(radix[i]) {
        case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 11: case 12: break;
        default: return false;
        }
    return passes > 0;
}
// (bluestein_points: the length L of the transforms Bluestein would run instead, 0: no alternative worth weighing -- a prime first
// pass is only planned where its p^2 work is expected to beat them)
// radix[0] a prime of 29 .. 127 (the rest of np is then planned behind it, radices up to 16): wp[m] = W_p^m, padded with ones to
// `entries` (the tick kernel stages the table where the power-of-two kernels keep R2 x R3 pass-2 twiddles)
void build_prime_twiddles(int p, size_t entries, std::vector<cfloat> &wp);
// tw: for every pass s >= 1 its twiddles [R_s][Ns] = W_(Ns R_s)^(k jm) (Ns = product of the radices before it), concatenated;
// tw_off[s] = where pass s starts.  w[k] = W_n^k, the real-split twiddles (k < n / 2)
void build_mixed_radix_tables(uint32_t n, int passes, const int radix[4], std::vector<cfloat> &tw, int tw_off[4], std::vector<cfloat> &w);

// vertex fill (cfg.vertices): the geometry constants of render_bars / render_curve and m_cap_verts (src/source.cpp:1293-1309)
struct VertexTables {
    int mode = 0;          // 0: bars, 1: curve triangle strip, 2: curve line strip, 3: stepped bars
    int step_stride = 0, max_steps = 0;
    int per_bar = 0, per_row = 0;
    int bar_stride = 0, cap_tris = 0;
    float cpos = 0, bottom = 0, channel_offset = 0, cap_radius = 0;
    int bottom_caps = 0, bot_offset = 0;
    int radial = 0;        // cap fans are full circles (m_radial)
    std::vector<float> cap_xy; // [cap_tris + 1][2]
};
void build_vertex_tables(const wf_config &cfg, int num_bars, VertexTables &out);

// get_gravity(seconds), src/source.hpp:301-312
float gravity_for(const wf_config &cfg, float seconds);
// DB_MIN, src/source.cpp:43
float db_min();

// FFT twiddle tables for the (R1, R2, R3) decomposition of the M-point complex FFT
void build_twiddles(int M, int R1, int R2, int R3, std::vector<cfloat> &tw1, std::vector<cfloat> &tw2,
                    std::vector<cfloat> &tws);

} // namespace wf
