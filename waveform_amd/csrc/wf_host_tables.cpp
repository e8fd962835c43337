// wf_host_tables.cpp -- see wf_host_tables.hpp.  Host-only (compiled by g++ with
// -ffp-contract=off so float expressions round exactly where the reference's do).
#include "wf_host_tables.hpp"
#include "wf_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <limits>
#include <numbers>

namespace wf {

namespace {

// src/math_funcs.hpp:25-29
inline float log_interp(float a, float b, float t) { return a * std::pow(b / a, t); }
// src/math_funcs.hpp:37-44
inline float sinc(float x)
{
    if(x == 0.0f)
        return 1.0f;
    const float tmp = std::numbers::pi_v<float> * x;
    return std::sin(tmp) / tmp;
}
// src/math_funcs.hpp:46-52
inline float lanczos(float x, float w)
{
    if(std::abs(x) < w)
        return sinc(x) * sinc(x / w);
    return 0.0f;
}

bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

// window table + m_window_sum, src/source.cpp:1190-1234
void build_window(const wf_config &cfg, HostTables &t)
{
    const size_t n = cfg.fft_size;
    if(cfg.window == WF_WINDOW_NONE) {
        t.window.clear();
        t.window_sum = (float)n;
        return;
    }
    t.window.resize(n);
    constexpr float pi = std::numbers::pi_v<float>;
    const size_t N = n - 1;
    constexpr float pi2 = 2 * pi;
    constexpr float pi4 = 4 * pi;
    constexpr float pi6 = 6 * pi;
    float *w = t.window.data();
    switch(cfg.window) {
    case WF_WINDOW_HAMMING:
        for(size_t i = 0; i < n; ++i)
            w[i] = 0.53836f - (0.46164f * std::cos((pi2 * i) / N));
        break;
    case WF_WINDOW_BLACKMAN:
        for(size_t i = 0; i < n; ++i)
            w[i] = 0.42f - (0.5f * std::cos((pi2 * i) / N)) + (0.08f * std::cos((pi4 * i) / N));
        break;
    case WF_WINDOW_BLACKMAN_HARRIS:
        for(size_t i = 0; i < n; ++i)
            w[i] = 0.35875f - (0.48829f * std::cos((pi2 * i) / N)) + (0.14128f * std::cos((pi4 * i) / N)) -
                   (0.01168f * std::cos((pi6 * i) / N));
        break;
    case WF_WINDOW_POWER_OF_SINE:
        for(size_t i = 0; i < n; ++i)
            w[i] = std::pow(std::sin((pi * i) / N), (float)cfg.sine_exponent);
        break;
    case WF_WINDOW_HANN:
    default:
        for(size_t i = 0; i < n; ++i)
            w[i] = 0.5f * (1 - std::cos((pi2 * i) / N));
        break;
    }
    float sum = 0.0f;
    for(size_t i = 0; i < n; ++i)
        sum += w[i];
    t.window_sum = sum;
}

// slope table, src/source.cpp:1282-1290
void build_slope(const wf_config &cfg, HostTables &t)
{
    t.slope.clear();
    if(!(cfg.slope > 0.0f))
        return;
    const size_t num_mods = cfg.fft_size / 2;
    const float maxmod = (float)(num_mods - 1);
    t.slope.resize(num_mods);
    for(size_t i = 0; i < num_mods; ++i)
        t.slope[i] = std::log10(log_interp(10.0f, 10000.0f, ((float)i * cfg.slope) / maxmod));
}

// init_rolloff, src/source.cpp:898-918
void build_rolloff(const wf_config &cfg, HostTables &t)
{
    t.rolloff.clear();
    if(!((cfg.rolloff_q > 0.0f) && (cfg.rolloff_rate > 0.0f)))
        return;
    const size_t sz = cfg.fft_size / 2;
    const float sr = (float)cfg.sample_rate;
    const float coeff = sr / (float)cfg.fft_size;
    const float ratio = std::exp2(cfg.rolloff_q);
    const float freq_low = (float)cfg.cutoff_low * ratio;
    const float freq_high = (float)cfg.cutoff_high / ratio;
    t.rolloff.resize(sz);
    t.rolloff[0] = 0.0f;
    for(size_t i = 1u; i < sz; ++i) {
        const float freq = i * coeff;
        const float ratio_low = freq_low / freq;
        const float ratio_high = freq / freq_high;
        const float low_attenuation = (ratio_low > 1.0f) ? (cfg.rolloff_rate * std::log2(ratio_low)) : 0.0f;
        const float high_attenuation = (ratio_high > 1.0f) ? (cfg.rolloff_rate * std::log2(ratio_high)) : 0.0f;
        t.rolloff[i] = low_attenuation + high_attenuation;
    }
}

// make_lanczos_kernel, src/filter.hpp:106-131 (radius 4 -> 8 taps per sample)
void build_lanczos(const std::vector<float> &indices, HostTables &t)
{
    const intmax_t radius = 4;
    const intmax_t size = (intmax_t)indices.size();
    t.interp_radius = (int)radius;
    t.interp_taps = (int)(radius * 2);
    t.interp_weights.assign((size_t)(size * radius * 2), 0.0f);
    const float fradius = (float)radius;
    for(intmax_t i = 0; i < size; ++i) {
        const float x = indices[(size_t)i];
        const intmax_t ix = (intmax_t)x;
        const intmax_t start = ix - radius + 1;
        const intmax_t stop = ix + radius;
        const intmax_t base = i * radius * 2;
        for(intmax_t j = start; j <= stop; ++j)
            t.interp_weights[(size_t)(base + (j - start))] = lanczos(x - j, fradius);
    }
}

// make_catrom_kernel, src/filter.hpp:67-104 (tension 0.5 -> 4 taps per sample)
void build_catrom(const std::vector<float> &indices, HostTables &t)
{
    const float tt = 0.5f;
    const float matrix[4][4] = {{0, -tt, 2 * tt, -tt}, {1, 0, tt - 3, 2 - tt}, {0, tt, 3 - (2 * tt), tt - 2}, {0, 0, -tt, tt}};
    const intmax_t size = (intmax_t)indices.size();
    t.interp_radius = 2;
    t.interp_taps = 4;
    t.interp_weights.assign((size_t)(size * 4), 0.0f);
    for(intmax_t i = 0; i < size; ++i) {
        const float u = indices[(size_t)i] - std::floor(indices[(size_t)i]);
        const float row[4] = {1, u, u * u, u * u * u};
        for(intmax_t j = 0; j < 4; ++j) {
            float sum = 0;
            for(intmax_t k = 0; k < 4; ++k)
                sum += row[k] * matrix[j][k];
            t.interp_weights[(size_t)((i * 4) + j)] = sum;
        }
    }
}

// render_bars geometry (src/source.cpp:1476-1494); render_curve maps onto [0, cpos - channel_offset] (:1411)
void render_geometry(const wf_config &cfg, bool curve, HostTables &t)
{
    const float center = (float)cfg.height / 2;
    const float bottom = (float)cfg.height;
    const float cpos = cfg.stereo ? center : bottom;
    const float cap_radius = (float)cfg.bar_width / 2.0f;
    const float channel_offset = cfg.channel_spacing * 0.5f;
    float border_top = cfg.rounded_caps ? cap_radius : 0.0f;
    float border_bottom = (cfg.rounded_caps && (!cfg.stereo || (cfg.channel_spacing > 0))) ? cpos - cap_radius : cpos;
    if(cfg.channel_spacing > 0)
        border_bottom -= channel_offset;
    if(cfg.min_bar_height > 0)
        border_bottom -= cfg.min_bar_height;
    border_bottom = std::clamp(border_bottom, border_top, cpos);
    if(curve) {
        border_top = 0.0f;
        border_bottom = cpos - channel_offset;
    }
    t.border_top = border_top;
    t.border_bottom = border_bottom;
    t.cpos = cpos;
}

// bar layout: update() :1267-1276, init_interp :837-896, render_bars geometry :1476-1494
void build_bars(const wf_config &cfg, HostTables &t)
{
    t.num_bars = 0;
    t.interp_indices.clear();
    t.band_widths.clear();
    t.interp_weights.clear();
    t.interp_radius = t.interp_taps = 0;
    t.bar_coef.clear();
    t.bar_bin.clear();
    t.bar_off.clear();
    const bool curve = !cfg.bars && cfg.curve;
    if(!cfg.bars && !curve)
        return;
    int num_bars;
    unsigned int sz;
    if(curve) { // render_curve: init_interp(m_width), one point per pixel column
        num_bars = (int)cfg.width;
        sz = cfg.width;
    } else {
        const int bar_stride = cfg.bar_width + cfg.bar_gap;
        num_bars = (int)(cfg.width / (unsigned int)bar_stride);
        if(((int)cfg.width - (num_bars * bar_stride)) >= cfg.bar_width)
            ++num_bars;
        sz = (unsigned int)(num_bars + 1);
    }
    t.num_bars = num_bars;

    const size_t fft_size = cfg.fft_size;
    const size_t maxbin = (fft_size / 2) - 1;
    const float sr = (float)cfg.sample_rate;
    const float lowbin = std::clamp((float)cfg.cutoff_low * fft_size / sr, 1.0f, (float)maxbin);
    const float highbin = std::clamp((float)cfg.cutoff_high * fft_size / sr, 1.0f, (float)maxbin);

    std::vector<float> idx(sz);
    if(cfg.log_scale) {
        for(unsigned int i = 0u; i < sz; ++i)
            idx[i] = std::clamp(log_interp(lowbin, highbin, (cfg.mirror_freq_axis ? i * 2.0f : (float)i) / (float)(sz - 1)), lowbin, highbin);
    } else {
        for(unsigned int i = 0u; i < sz; ++i)
            idx[i] = std::clamp(std::lerp(lowbin, highbin, (cfg.mirror_freq_axis ? i * 2.0f : (float)i) / (float)(sz - 1)), lowbin, highbin);
    }
    t.band_widths.resize((size_t)num_bars);
    for(int i = 0; i < num_bars; ++i)
        t.band_widths[(size_t)i] = curve ? 1 : std::max((int)(idx[(size_t)i + 1] - idx[(size_t)i]), 1);

    if(cfg.interp_mode != WF_INTERP_POINT) {
        if(!curve) { // bars: the indices so far are band starts; expand to one position per band sample (:878-889)
            std::vector<float> samples;
            for(int i = 0; i < num_bars; ++i) {
                const int count = t.band_widths[(size_t)i];
                for(int j = 0; j < count; ++j)
                    samples.push_back(idx[(size_t)i] + j);
            }
            t.interp_indices = std::move(samples);
        } else {
            t.interp_indices = std::move(idx);
        }
        if(cfg.interp_mode == WF_INTERP_LANCZOS)
            build_lanczos(t.interp_indices, t);
        else
            build_catrom(t.interp_indices, t);
    } else {
        t.interp_indices = std::move(idx);
    }

    render_geometry(cfg, curve, t);

    // Gaussian filter across the outputs: make_gauss_kernel(m_filter_radius), src/filter.hpp:40-65 (float throughout)
    t.gauss.clear();
    t.gauss_wsum.clear();
    t.gauss_radius = 0;
    t.gauss_sum = 0.0f;
    if(cfg.filter_mode == WF_FILTER_GAUSS) {
        const float sigma = std::max(std::abs(cfg.filter_radius), 0.01f);
        const int w = (int)std::ceil(3.0f * sigma);
        const int size = (2 * w) - 1;
        t.gauss.resize((size_t)size);
        t.gauss_radius = w;
        constexpr float pi2 = std::numbers::pi_v<float> * 2.0f;
        const float sigsqr = sigma * sigma;
        const float expdenom = 2.0f * sigsqr;
        const float coeff = (1.0f / (std::sqrt(pi2) * sigma));
        int j = 0;
        for(int i = -w + 1; i < w; ++i) {
            const float exponent = -((float)(i * i) / expdenom);
            const float weight = coeff * std::exp(exponent);
            t.gauss[(size_t)j++] = weight;
            t.gauss_sum += weight;
        }
        // weighted_avg (src/filter.hpp:133-157): at the edges the divisor is the running sum of the weights that are used
        t.gauss_wsum.assign((size_t)num_bars, t.gauss_sum);
        for(intmax_t o = 0; o < num_bars; ++o) {
            const intmax_t start = (o - w) + 1, stop = o + w;
            if((start < 0) || (stop > num_bars)) {
                float wsum = 0.0f;
                for(intmax_t i = std::max<intmax_t>(start, 0); i < std::min<intmax_t>(stop, num_bars); ++i)
                    wsum += t.gauss[(size_t)(i - start)];
                t.gauss_wsum[(size_t)o] = wsum;
            }
        }
    }

    // ---- composite per-bar kernels for the device (see BarArgs in wf_tick_phases.hpp) ---------------------------------
    const intmax_t M = (intmax_t)(cfg.fft_size / 2);
    t.bar_rows_bins = (int)M;
    t.bar_off.assign((size_t)num_bars + 1, 0);
    size_t k = 0; // running sample index (interpolating modes)
    for(int i = 0; i < num_bars; ++i) {
        const intmax_t count = t.band_widths[(size_t)i];
        intmax_t lo, hi; // bin range [lo, hi)
        std::vector<double> acc;
        if(cfg.interp_mode == WF_INTERP_POINT) {
            // sum += m_decibels[(size_t)m_interp_indices[i] + j], src/source.cpp:1529-1530
            lo = std::clamp<intmax_t>((intmax_t)t.interp_indices[(size_t)i], 0, M);
            hi = std::clamp<intmax_t>(lo + count, 0, M);
            acc.assign((size_t)(hi - lo), 1.0);
        } else {
            const intmax_t radius = t.interp_radius, taps = t.interp_taps;
            const intmax_t ix_first = (intmax_t)t.interp_indices[k];
            const intmax_t ix_last = (intmax_t)t.interp_indices[k + (size_t)count - 1];
            lo = std::clamp<intmax_t>(ix_first - radius + 1, 0, M);
            hi = std::clamp<intmax_t>(std::max(ix_last, ix_first) + radius + 1, 0, M);
            acc.assign((size_t)(hi - lo), 0.0);
            for(intmax_t s = 0; s < count; ++s, ++k) {
                // kernel_convolve(samples, sz, kernel, (intmax_t)x[k], l), src/filter.hpp:160-169
                const intmax_t index = (intmax_t)t.interp_indices[k];
                const intmax_t start = (index - radius) + 1;
                const intmax_t stop = std::min(index + radius + 1, M);
                for(intmax_t j = std::max<intmax_t>(start, 0); j < stop; ++j) {
                    if(j >= lo && j < hi && (j - start) < taps)
                        acc[(size_t)(j - lo)] += (double)t.interp_weights[k * (size_t)taps + (size_t)(j - start)];
                }
            }
        }
        t.bar_off[(size_t)i] = (int)t.bar_coef.size();
        for(size_t j = 0; j < acc.size(); ++j) {
            t.bar_coef.push_back((float)acc[j]);
            t.bar_bin.push_back((int)(lo + (intmax_t)j));
        }
    }
    t.bar_off[(size_t)num_bars] = (int)t.bar_coef.size();
}

} // namespace

std::vector<int> bar_chunks(const HostTables &t, size_t cap_floats)
{
    std::vector<int> chunks{0};
    int begin = 0;
    for(int b = 0; b < t.num_bars; ++b) {
        const size_t upto = (size_t)(t.bar_off[(size_t)b + 1] - t.bar_off[(size_t)begin]);
        if(upto > cap_floats && b > begin) {
            chunks.push_back(b);
            begin = b;
        }
    }
    chunks.push_back(t.num_bars);
    return chunks;
}

bool bar_segments(const HostTables &t, int threads, int max_blocks, BarLaneTables &out, bool wave_local)
{
    out = BarLaneTables{};
    if(t.num_bars <= 0 || t.num_bars > threads)
        return false;
    const int M = (int)t.bar_rows_bins;
    // A bar's entries sit on consecutive bins [lo, lo + len).  Segments are cut at bins that are multiples of 4, counted
    // from lo rounded down (the up to three bins before lo get coefficient 0), so that a thread reads its bins as 16-byte
    // words of the row parked in LDS.
    auto span = [&](int b, int &first, int &len) {
        const int o = t.bar_off[(size_t)b];
        len = t.bar_off[(size_t)b + 1] - o;
        const int lo = len > 0 ? t.bar_bin[(size_t)o] : 0;
        first = lo & ~3;
        return lo - first; // leading bins with coefficient 0
    };
    auto segs_of = [&](int b, int L) {
        int first, len;
        const int lead = span(b, first, len);
        return len == 0 ? 1 : (lead + len + L - 1) / L;
    };
    // slot of every bar's first segment for segment length L, or false if the layout does not fit the threads
    auto place = [&](int L, bool local, std::vector<int> &start) {
        start.assign((size_t)t.num_bars + 1, 0);
        int s = 0;
        for(int b = 0; b < t.num_bars; ++b) {
            const int k = segs_of(b, L);
            if(local) {
                if(k > 64)
                    return false;
                if((s % 64) + k > 64) // the bar would straddle a wavefront: start it in the next one
                    s = (s + 63) / 64 * 64;
            }
            start[(size_t)b] = s;
            s += k;
        }
        start[(size_t)t.num_bars] = s;
        return s <= threads;
    };
    std::vector<int> start;
    int L = 0;
    bool local = false;
    if(wave_local)
        for(int l = 4; l <= 4 * max_blocks && l <= M; l += 4)
            if(place(l, true, start)) {
                L = l;
                local = true;
                break;
            }
    if(L == 0)
        for(int l = 4; l <= 4 * max_blocks && l <= M; l += 4) // smallest multiple of 4 whose segment count fits the threads
            if(place(l, false, start)) {
                L = l;
                break;
            }
    if(L == 0)
        return false;
    out.wave_local = local;
    out.blocks = L / 4;
    out.coef.assign((size_t)out.blocks * threads * 4, 0.0f);
    out.base.assign((size_t)threads, 0);
    out.seg_group.assign((size_t)threads, 0);
    out.lead_bar.assign((size_t)threads, -1);
    out.lead_end.assign((size_t)threads, 0);
    for(int b = 0; b < t.num_bars; ++b) {
        int first, len;
        const int lead = span(b, first, len);
        const int o = t.bar_off[(size_t)b];
        const int segs = segs_of(b, L), s0 = start[(size_t)b];
        out.bar_seg.push_back(s0);
        out.lead_bar[(size_t)s0] = b;
        for(int g = 0; g < segs; ++g)
            out.lead_end[(size_t)(s0 + g)] = s0 + segs;
        for(int g = 0; g < segs; ++g) {
            const int s = s0 + g;
            // bins [bstart, bstart + L) of the row; a segment that would reach past the row moves down (its coefficients with it)
            int bstart = first + g * L;
            if(bstart + L > M)
                bstart = M - L;
            out.base[(size_t)s] = bstart;
            for(int k = 0; k < L; ++k) {
                const int bin = bstart + k, e = bin - (first + lead); // entry index within the bar
                // every bin belongs to exactly one segment of the bar: the one whose nominal range [first + g L, +L) holds it
                const bool mine = bin >= first + g * L && bin < first + (g + 1) * L;
                if(mine && e >= 0 && e < len)
                    out.coef[((size_t)(k / 4) * threads + s) * 4 + (size_t)(k % 4)] = t.bar_coef[(size_t)o + e];
            }
        }
        for(int k = s0; k < s0 + segs; k += 8)
            out.seg_group[(size_t)k] = std::min(8, s0 + segs - k);
    }
    out.bar_seg.push_back(start[(size_t)t.num_bars]); // (wave-local: bar_seg[b + 1] may lie beyond bar b's last segment: padding)
    out.num_segs = start[(size_t)t.num_bars];
    return true;
}

bool bar_pieces(const HostTables &t, int threads, int points, int max_blocks, BarPieceTables &out)
{
    out = BarPieceTables{};
    if(t.num_bars <= 0 || t.num_bars > 64 || threads < 64 || threads % 64 || points < 4 || points % 4)
        return false;
    const int M = (int)t.bar_rows_bins, wps = threads / 64, groups = points / 4;
    if(M < 256 * wps)
        return false; // (rows shorter than one chunk per wavefront: the zero-padded sizes keep bar_segments' layout)
    struct Piece { int bar, wave, lo, hi, cend; }; // bins [lo, hi) of one bar inside the chunk that ends at cend
    std::vector<Piece> pieces;
    out.bar_piece.assign((size_t)t.num_bars + 1, 0);
    for(int b = 0; b < t.num_bars; ++b) {
        out.bar_piece[(size_t)b] = (int)pieces.size();
        const int o = t.bar_off[(size_t)b], len = t.bar_off[(size_t)b + 1] - o;
        if(len <= 0)
            return false; // (a bar without entries has no last lane to emit it on one wavefront per spectrum: bar_segments' layouts take the display)
        const int lo = t.bar_bin[(size_t)o], hi = lo + len;
        for(int e = 1; e < len; ++e)
            if(t.bar_bin[(size_t)o + e] != lo + e)
                return false; // (entries of a bar sit on consecutive bins: build_bars)
        if(hi > 256 * wps * groups)
            return false;
        if(wps == 1)
            pieces.push_back({b, 0, lo, hi, M});
        else
            for(int c = lo / 256; c * 256 < hi; ++c)
                pieces.push_back({b, c % wps, std::max(lo, c * 256), std::min(hi, c * 256 + 256), c * 256 + 256});
    }
    out.bar_piece[(size_t)t.num_bars] = (int)pieces.size();
    // the smallest segment length (a multiple of 4 bins) with which every wavefront's pieces fit its 64 lanes
    int L = 0;
    for(int l = 4; l <= 4 * max_blocks && L == 0; l += 4) {
        std::vector<int> used((size_t)wps, 0);
        bool ok = true;
        for(const Piece &q : pieces) {
            const int k = ((q.lo & 3) + (q.hi - q.lo) + l - 1) / l;
            used[(size_t)q.wave] += k;
            ok = ok && k <= 64 && used[(size_t)q.wave] <= 64;
        }
        if(ok)
            L = l;
    }
    if(L == 0)
        return false;
    out.blocks = L / 4;
    out.coef.assign((size_t)out.blocks * threads * 4, 0.0f);
    out.base.assign((size_t)threads, 0);
    out.info.assign((size_t)threads, 0);
    for(int w = 0; w < wps; ++w) // lanes without a segment read (and multiply by zero) the first bins of their own wavefront
        for(int l = 0; l < 64; ++l)
            out.base[(size_t)(w * 64 + l)] = std::min(256 * w, M - L) & ~3;
    std::vector<int> next((size_t)wps, 0);
    for(size_t pi = 0; pi < pieces.size(); ++pi) {
        const Piece &q = pieces[pi];
        const int first = q.lo & ~3, k = ((q.lo - first) + (q.hi - q.lo) + L - 1) / L;
        const int l0 = next[(size_t)q.wave];
        next[(size_t)q.wave] += k;
        const int b_off = t.bar_off[(size_t)q.bar], b_lo = t.bar_bin[(size_t)b_off];
        for(int g = 0; g < k; ++g) {
            const int lane = l0 + g, s = q.wave * 64 + lane;
            // bins [bstart, bstart + L); a segment that would reach past its chunk (or the row) moves down, its coefficients with it
            int bstart = first + g * L;
            if(bstart + L > std::min(q.cend, M))
                bstart = std::min(q.cend, M) - L;
            out.base[(size_t)s] = bstart;
            for(int j = 0; j < L; ++j) {
                const int bin = bstart + j;
                const bool mine = bin >= first + g * L && bin < first + (g + 1) * L && bin >= q.lo && bin < q.hi;
                if(mine)
                    out.coef[((size_t)(j / 4) * threads + s) * 4 + (size_t)(j % 4)] = t.bar_coef[(size_t)b_off + (bin - b_lo)];
            }
            // the segmented inclusive prefix over lanes [l0, l0 + k): which of the scan's six steps this lane takes
            int flags = 0;
            for(int d = 0; d < 4; ++d)
                if(lane - (1 << d) >= l0 && (lane & 15) >= (1 << d))
                    flags |= 1 << d;
            if(lane >= 16 && ((lane >> 4) & 1) && l0 <= (lane & ~15) - 1)
                flags |= 1 << 4; // rows 1 and 3 add the last lane of the row before them
            if(lane >= 32 && l0 <= 31)
                flags |= 1 << 5; // rows 2 and 3 add lane 31
            out.info[(size_t)s] = flags | (g == k - 1 ? ((wps == 1 ? q.bar : (int)pi) + 1) << 8 : 0);
        }
    }
    out.num_segs = 0;
    for(int w = 0; w < wps; ++w)
        out.num_segs += next[(size_t)w];
    out.num_slots = (int)pieces.size();
    return true;
}

bool bar_ps(const HostTables &t, int threads, BarPsTables &out)
{
    out = BarPsTables{};
    const intmax_t M = t.bar_rows_bins;
    if(t.num_bars <= 0 || t.num_bars > 254 || threads < 64 || threads % 64 || M < 256 || M % 256 || M > 32767)
        return false;
    if(t.gauss_radius > 0 || (int)t.band_widths.size() != t.num_bars)
        return false;
    const bool point = t.interp_taps == 0;
    const int taps = t.interp_taps, radius = t.interp_radius;
    if(!point && !((taps == 8 && radius == 4) || (taps == 4 && radius == 2)))
        return false;
    struct Sub { int bar; intmax_t lo, hi; float w[8]; };
    std::vector<Sub> subs;
    size_t k = 0;
    for(int i = 0; i < t.num_bars; ++i) {
        const intmax_t count = t.band_widths[(size_t)i];
        if(count < 1 || count > 65535)
            return false;
        if(point) { // sum += m_decibels[(size_t)m_interp_indices[i] + j], src/source.cpp:1529-1530
            Sub s{i, 0, 0, {0, 0, 0, 1.0f, 0, 0, 0, 0}};
            s.lo = std::clamp<intmax_t>((intmax_t)t.interp_indices[(size_t)i], 0, M);
            s.hi = std::clamp<intmax_t>(s.lo + count, 0, M);
            subs.push_back(s);
            continue;
        }
        if(k + (size_t)count > t.interp_indices.size() || (k + (size_t)count) * (size_t)taps > t.interp_weights.size())
            return false;
        for(intmax_t j = 0; j < count; ++j, ++k) {
            const intmax_t ix = (intmax_t)t.interp_indices[k];
            float w[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // tap j of the sample's row weighs bin ix - radius + 1 + j: slot j + 4 - radius of the 8
            for(int q = 0; q < taps; ++q)
                w[q + 4 - radius] = t.interp_weights[k * (size_t)taps + (size_t)q];
            if(!subs.empty() && subs.back().bar == i && subs.back().hi == ix && std::memcmp(subs.back().w, w, sizeof w) == 0)
                ++subs.back().hi;
            else {
                Sub s{i, ix, ix + 1, {}};
                std::memcpy(s.w, w, sizeof w);
                subs.push_back(s);
            }
        }
    }
    for(const Sub &s : subs)
        if(s.lo < 0 || s.hi > M || s.hi < s.lo)
            return false;
    // two lanes per sub-band: lane 2 j its low edge, lane 2 j + 1 its high edge
    if(subs.size() > 32)
        return false; // (one finishing wavefront per spectrum; the other layouts take the displays with more)
    out.num_lanes = 64;
    out.num_subs = (int)subs.size();
    out.tab.assign((size_t)3 * 64 * 4, 0.0f);
    auto word = [&](int ln, int c, int e) -> float & { return out.tab[((size_t)c * 64 + (size_t)ln) * 4 + (size_t)e]; };
    auto bits = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    for(size_t a = 0; a < subs.size(); ++a) {
        const Sub &s = subs[a];
        float clo[7], chi[7], sw = 0.0f;
        if(s.hi - s.lo <= 7) {
            // direct: the composite coefficient of bin m is the sum of the taps of every sample of the sub-band that land on it
            auto coef = [&](intmax_t m) {
                double c = 0.0;
                for(intmax_t ix = s.lo; ix < s.hi; ++ix) {
                    const intmax_t q = m - ix + 3;
                    if(q >= 0 && q < 8)
                        c += (double)s.w[q];
                }
                return (float)c;
            };
            for(int j = 0; j < 7; ++j) {
                clo[j] = coef(s.lo - 3 + j);
                chi[j] = (s.hi - 3 + j > s.lo + 3) ? coef(s.hi - 3 + j) : 0.0f;
            }
        } else {
            double c = 0.0;
            for(int j = 0; j < 7; ++j) {
                c += (double)s.w[j];
                clo[j] = (float)c;
                chi[j] = -(float)c;
            }
            sw = (float)(c + (double)s.w[7]);
        }
        // the segmented inclusive prefix over the bar's lanes [l0, l1]: which of seg_prefix_scan's six steps a lane takes
        size_t first = a, last = a;
        while(first > 0 && subs[first - 1].bar == s.bar)
            --first;
        while(last + 1 < subs.size() && subs[last + 1].bar == s.bar)
            ++last;
        const int l0 = 2 * (int)first;
        for(int side = 0; side < 2; ++side) {
            const int l = 2 * (int)a + side;
            const float *c = side ? chi : clo;
            for(int j = 0; j < 4; ++j)
                word(l, 0, j) = c[j];
            word(l, 1, 0) = c[4]; word(l, 1, 1) = c[5]; word(l, 1, 2) = c[6]; word(l, 1, 3) = sw;
            uint32_t flags = 0;
            for(int d = 0; d < 4; ++d)
                if(l - (1 << d) >= l0 && (l & 15) >= (1 << d))
                    flags |= 1u << d;
            if(l >= 16 && ((l >> 4) & 1) && l0 <= (l & ~15) - 1)
                flags |= 1u << 4;
            if(l >= 32 && l0 <= 31)
                flags |= 1u << 5;
            // the bar is finished by the low-edge lane of its last sub-band (the sums meet on even lanes)
            const bool finishes = a == last && side == 0;
            const uint32_t info = flags | (finishes ? (uint32_t)(s.bar + 1) << 8 : 0u) | (uint32_t)t.band_widths[(size_t)s.bar] << 16;
            word(l, 2, 0) = bits((uint32_t)(side ? s.hi : s.lo));
            word(l, 2, 1) = bits(info);
        }
    }
    return true;
}

bool curve_lanes(const HostTables &t, const wf_config &cfg, int threads, int max_steps, CurveLaneTables &out)
{
    out = CurveLaneTables{};
    const int n = t.num_bars;
    const int M = (int)(cfg.fft_size / 2);
    if(n <= 0 || M < 8)
        return false;
    const int steps = (n + threads - 1) / threads;
    (void)max_steps; // more steps than a thread's registers hold: the kernel streams the points (BarArgs::stream_steps)
    out.steps = steps;
    const int padded = (steps + 3) / 4 * 4; // the kernel takes the steps four at a time
    if(cfg.interp_mode == WF_INTERP_CATROM) {
        // positions only; init_interp clamps them to [lowbin, highbin] within [1, M - 1], which the device relies on
        out.x.assign((size_t)padded * threads, 1.0f);
        for(int o = 0; o < n; ++o) {
            const float x = t.interp_indices[(size_t)o];
            if(!(x >= 1.0f && x <= (float)(M - 1)))
                return false;
            out.x[(size_t)o] = x;
        }
        return true;
    }
    out.coef.assign((size_t)padded * threads * 8, 0.0f);
    out.base.assign((size_t)padded * threads, 0);
    for(int o = 0; o < n; ++o) {
        const int e0 = t.bar_off[(size_t)o], len = t.bar_off[(size_t)o + 1] - e0;
        if(len > 8)
            return false; // not a one-sample output (never for curve tables: <= 2 * radius taps)
        int base = len > 0 ? t.bar_bin[(size_t)e0] : 0;
        if(base > M - 8)
            base = M - 8; // keep all eight reads inside the row; the coefficients move with it
        out.base[(size_t)o] = base;
        for(int j = 0; j < len; ++j)
            out.coef[(size_t)o * 8 + (size_t)(t.bar_bin[(size_t)e0 + j] - base)] = t.bar_coef[(size_t)e0 + j];
    }
    return true;
}

void build_vertex_tables(const wf_config &cfg, int num_bars, VertexTables &out)
{
    out = VertexTables{};
    const bool curve = !cfg.bars && cfg.curve;
    const float center = (float)cfg.height / 2;
    out.bottom = (float)cfg.height;
    out.cpos = cfg.stereo ? center : out.bottom;
    out.channel_offset = cfg.channel_spacing * 0.5f;
    if(curve) {
        out.mode = cfg.vertices == 2 ? 2 : 1;
        out.per_row = out.mode == 2 ? num_bars : 2 * num_bars; // src/source.cpp:985
        return;
    }
    out.mode = 0;
    out.bar_stride = cfg.bar_width + cfg.bar_gap;
    out.per_bar = 6;
    if(cfg.vertices == 3) { // stepped bars: create_vbuf, src/source.cpp:988-1000
        out.mode = 3;
        out.step_stride = cfg.step_width + cfg.step_gap;
        size_t max_steps = (size_t)((out.cpos - out.channel_offset) / out.step_stride);
        if(((int)out.cpos - (int)(max_steps * out.step_stride) - (int)out.channel_offset) > cfg.step_width)
            ++max_steps;
        out.max_steps = (int)max_steps;
        out.per_row = num_bars * 6 * out.max_steps;
        return;
    }
    if(cfg.rounded_caps) { // :1293-1309, float throughout
        constexpr float pi = std::numbers::pi_v<float>;
        out.cap_radius = (float)cfg.bar_width / 2.0f;
        out.cap_tris = std::max((int)((2 * pi * out.cap_radius) / 3.0f), 4);
        if(out.cap_tris & 1)
            out.cap_tris += 1;
        const float angle = (2 * pi) / (float)out.cap_tris;
        const int verts = out.cap_tris + 1;
        out.cap_xy.resize((size_t)verts * 2);
        for(int j = 0; j < verts; ++j) {
            const float a = j * angle;
            out.cap_xy[(size_t)2 * j] = out.cap_radius * std::cos(a);
            out.cap_xy[(size_t)2 * j + 1] = out.cap_radius * std::sin(a);
        }
        out.bottom_caps = (!cfg.stereo || cfg.channel_spacing > 0) ? 1 : 0;                   // :1645
        out.radial = cfg.radial ? 1 : 0;                                                      // :1632-1633, :1646-1647
        out.per_bar += 3 * (out.radial ? out.cap_tris : out.cap_tris / 2) * (1 + out.bottom_caps);
    }
    out.bot_offset = ((cfg.rounded_caps && !cfg.stereo) || cfg.channel_spacing > 0) ? 1 : 0; // :1619
    out.per_row = out.per_bar * num_bars;
}

float db_min()
{
    // const float WAVSource::DB_MIN = 20.0f * std::log10(std::numeric_limits<float>::min()); (src/source.cpp:43)
    static const float v = 20.0f * std::log10(std::numeric_limits<float>::min());
    return v;
}

float gravity_for(const wf_config &cfg, float seconds)
{
    constexpr float denom = 0.03868924705242879469662125316986f;
    constexpr float hi = denom * 5.0f;
    constexpr float lo = 0.0f;
    if((cfg.tsmoothing == WF_TSMOOTH_NONE) || (cfg.gravity <= 0.0f))
        return 0.0f;
    return (cfg.tsmoothing == WF_TSMOOTH_TVEXPONENTIAL) ? std::exp(-seconds / std::lerp(lo, hi, cfg.gravity)) : cfg.gravity;
}

void normalize_config(wf_config &cfg)
{
    // what get_settings() makes of out-of-range combinations before update() sees them (src/source.cpp:567-579)
    if((cfg.cutoff_high - cfg.cutoff_low) < 0) {
        cfg.cutoff_high = 17500;
        cfg.cutoff_low = 120;
    }
    if((cfg.ceiling_db - cfg.floor_db) < 1) {
        cfg.ceiling_db = 0;
        cfg.floor_db = -120;
    }
    if(!cfg.stereo || (((int)cfg.height - cfg.channel_spacing) < 1))
        cfg.channel_spacing = 0;
    if(cfg.vertices == 3) // display_mode STEPPED_BAR: m_rounded_caps = false (src/source.cpp:648-649)
        cfg.rounded_caps = 0;
}

uint32_t meter_config(wf_config &cfg)
{
    // "turn off stuff we don't need in this mode", src/source.cpp:1108-1118
    cfg.window = WF_WINDOW_NONE;
    cfg.interp_mode = WF_INTERP_POINT;
    cfg.filter_mode = WF_FILTER_NONE;
    cfg.slope = 0.0f;
    cfg.stereo = 0;
    cfg.normalize_volume = 0;
    cfg.mirror_freq_axis = 0;
    cfg.bars = 0;
    cfg.curve = 0;
    // "repurpose m_fft_size for meter buffer size", :1121
    cfg.fft_size = (uint32_t)(size_t((double)cfg.sample_rate * ((double)cfg.meter_ms / 1000.0)) & (size_t)-16);
    return cfg.fft_size;
}

uint32_t waveform_config(wf_config &cfg)
{
    // "turn off stuff we don't need in this mode", src/source.cpp:1132-1137
    cfg.window = WF_WINDOW_NONE;
    cfg.slope = 0.0f;
    cfg.mirror_freq_axis = 0;
    cfg.log_scale = 0;
    cfg.meter = 0;
    cfg.bars = 0;  // the curve through the points (render_curve) stays with the host's renderer
    cfg.curve = 0;
    cfg.fft_size = cfg.width; // "repurpose m_fft_size for buffer size", :1140
    return (uint32_t)(size_t)((double)cfg.sample_rate * ((double)cfg.meter_ms / 1000.0)); // m_waveform_samples, :1141
}

int build_host_tables(const wf_config &cfg, HostTables &out)
{
    if(cfg.waveform) {
        if(cfg.capture_channels < 1 || cfg.capture_channels > 2 || cfg.sample_rate == 0 || cfg.meter_ms <= 0 || cfg.width == 0)
            return WF_HIP_ERR_INVALID;
        if(((uint64_t)cfg.meter_ms * 1000000ull) / cfg.width == 0) // step_ns, src/source_generic.cpp:299
            return WF_HIP_ERR_INVALID;
        if(cfg.width > 8192u) // the kernel stages 2 * capture_channels rows of `width` floats in LDS
            return WF_HIP_ERR_UNSUPPORTED;
        out = HostTables{};
        out.window_sum = (float)cfg.fft_size;
        out.output_channels = ((cfg.capture_channels > 1) || cfg.stereo) ? 2u : 1u;
        out.display_channels = cfg.stereo ? 2u : 1u;
        return WF_HIP_OK;
    }
    if(cfg.meter) {
        // level meter: no FFT, no tables; one "bar" per captured channel through render_bars' mapping (:1257-1266, :1505-1509)
        if(cfg.capture_channels < 1 || cfg.capture_channels > 2 || cfg.sample_rate == 0 || cfg.meter_ms <= 0 || cfg.fft_size < 16)
            return WF_HIP_ERR_INVALID;
        if(cfg.ceiling_db <= cfg.floor_db)
            return WF_HIP_ERR_INVALID;
        out = HostTables{};
        out.window_sum = (float)cfg.fft_size;
        out.output_channels = ((cfg.capture_channels > 1) || cfg.stereo) ? 2u : 1u;
        out.display_channels = 1u;
        out.num_bars = (int)cfg.capture_channels;
        render_geometry(cfg, false, out);
        return WF_HIP_OK;
    }
    if(cfg.fft_size < 128 || (cfg.fft_size & 15u)) // the reference raises / aligns such values itself (src/source.cpp:562-565)
        return WF_HIP_ERR_UNSUPPORTED;
    if(cfg.fft_size > 65536u) // the reference's own ceiling ("enable large FFT", src/source.cpp:349, :359-363)
        return WF_HIP_ERR_UNSUPPORTED;
    if(cfg.capture_channels < 1 || cfg.capture_channels > 2 || cfg.sample_rate == 0)
        return WF_HIP_ERR_INVALID;
    if(cfg.bars && ((cfg.bar_width + cfg.bar_gap) <= 0 || cfg.width == 0))
        return WF_HIP_ERR_INVALID;
    if(!cfg.bars && cfg.curve && cfg.width < 2) // init_interp divides by (width - 1)
        return WF_HIP_ERR_INVALID;
    if((cfg.bars || cfg.curve) && cfg.ceiling_db <= cfg.floor_db) // the dB -> pixel mapping divides by (ceiling - floor)
        return WF_HIP_ERR_INVALID;
    if(cfg.filter_mode != WF_FILTER_NONE && (cfg.filter_mode != WF_FILTER_GAUSS || !std::isfinite(cfg.filter_radius)))
        return WF_HIP_ERR_INVALID;
    build_window(cfg, out);
    build_slope(cfg, out);
    build_rolloff(cfg, out);
    out.output_channels = ((cfg.capture_channels > 1) || cfg.stereo) ? 2u : 1u; // src/source.cpp:1171
    out.display_channels = cfg.stereo ? 2u : 1u;
    build_bars(cfg, out);
    return WF_HIP_OK;
}

uint32_t bluestein_length(uint32_t n)
{
    if(is_pow2(n))
        return 0;
    uint32_t L = 512;
    if(n <= 16384u) { // packed form: n/2 complex points, every output wanted: L >= 2 (n/2) - 1
        while(L < n - 1u)
            L <<= 1;
        return L; // <= 16384 complex points: inside one workgroup (the 32768-sample geometry at most)
    }
    while((uint64_t)L * 2 < (uint64_t)n * 3) // direct form: L >= 3n/2
        L <<= 1;
    return L; // 32768 .. 131072: wf_big.hpp
}

void build_big_twiddles(uint32_t L, uint32_t rows, uint32_t real_n, std::vector<cfloat> &tw_big, std::vector<cfloat> &tws_big)
{
    const double two_pi = 6.283185307179586476925286766559;
    const uint32_t L2 = L / rows;
    tw_big.resize((size_t)L);
    for(uint32_t k1 = 0; k1 < rows; ++k1)
        for(uint32_t n2 = 0; n2 < L2; ++n2) {
            const double a = -two_pi * (double)(((uint64_t)n2 * k1) % L) / (double)L;
            tw_big[(size_t)k1 * L2 + n2] = {(float)std::cos(a), (float)std::sin(a)};
        }
    tws_big.clear();
    if(real_n) { // W_n^k for the real split of the packed n-sample transform
        tws_big.resize((size_t)real_n / 2);
        for(uint32_t k = 0; k < real_n / 2; ++k) {
            const double a = -two_pi * (double)k / (double)real_n;
            tws_big[k] = {(float)std::cos(a), (float)std::sin(a)};
        }
    }
}

namespace {
// iterative radix-2 complex FFT in double (host, tables only)
void fft_double(std::vector<double> &re, std::vector<double> &im)
{
    const size_t n = re.size();
    for(size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for(; j & bit; bit >>= 1)
            j ^= bit;
        j ^= bit;
        if(i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    const double two_pi = 6.283185307179586476925286766559;
    for(size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len >> 1;
        for(size_t k = 0; k < half; ++k) {
            const double a = -two_pi * (double)k / (double)len;
            const double cr = std::cos(a), ci = std::sin(a);
            for(size_t i = k; i < n; i += len) {
                const double xr = re[i + half] * cr - im[i + half] * ci;
                const double xi = re[i + half] * ci + im[i + half] * cr;
                re[i + half] = re[i] - xr;
                im[i + half] = im[i] - xi;
                re[i] += xr;
                im[i] += xi;
            }
        }
    }
}
} // namespace

uint32_t build_bluestein_rows(uint32_t np, uint32_t C, std::vector<cfloat> &rowtw, std::vector<cfloat> &bhat, std::vector<cfloat> &q)
{
    const uint32_t R = np / C;
    uint32_t L = 1024; // (the smallest container: the 2048-sample geometry)
    while(L < 2u * R - 1u)
        L <<= 1;
    const double pi = 3.14159265358979323846264338327950288;
    auto chirp = [&](uint64_t m, double &cr, double &ci) { // w_m = exp(i pi m^2 / R), the phase reduced exactly
        const uint64_t ph = (m * m) % (2ull * R);
        const double a = pi * (double)ph / (double)R;
        cr = std::cos(a);
        ci = std::sin(a);
    };
    rowtw.resize((size_t)np);
    for(uint32_t k1 = 0; k1 < C; ++k1)
        for(uint32_t n2 = 0; n2 < R; ++n2) {
            double cr, ci;
            chirp(n2, cr, ci);
            const double a = -2.0 * pi * (double)(((uint64_t)n2 * k1) % np) / (double)np;
            const double tr = std::cos(a), ti = std::sin(a);
            // W_np^(n2 k1) * conj(w_n2)
            rowtw[(size_t)k1 * R + n2] = cfloat{(float)(tr * cr + ti * ci), (float)(ti * cr - tr * ci)};
        }
    std::vector<double> br(L, 0.0), bi(L, 0.0);
    for(uint32_t m = 0; m < R; ++m)
        chirp(m, br[m], bi[m]);
    for(uint32_t m = 1; m < R; ++m) // negative lags, wrapped
        chirp(m, br[L - m], bi[L - m]);
    fft_double(br, bi);
    bhat.resize(L);
    for(uint32_t k = 0; k < L; ++k)
        bhat[k] = cfloat{(float)br[k], (float)bi[k]};
    q.resize(R);
    for(uint32_t k = 0; k < R; ++k) {
        double cr, ci;
        chirp(k, cr, ci);
        q[k] = cfloat{(float)(cr / (double)L), (float)(-ci / (double)L)};
    }
    return L;
}

void build_bluestein(const wf_config &cfg, const HostTables &t, BluesteinTables &out)
{
    const uint32_t n = cfg.fft_size;
    out = BluesteinTables{};
    out.L = bluestein_length(n);
    if(out.L == 0)
        return;
    const uint32_t L = out.L;
    out.packed = n <= 16384u;
    const uint32_t np = out.packed ? n / 2 : n; // points of the transform Bluestein computes
    const double pi = 3.14159265358979323846264338327950288;
    auto chirp = [&](uint64_t m, double &cr, double &ci) { // w_m = exp(i pi m^2 / np), the phase reduced exactly
        const uint64_t ph = (m * m) % (2ull * np);
        const double a = pi * (double)ph / (double)np;
        cr = std::cos(a);
        ci = std::sin(a);
    };
    auto win = [&](uint32_t i) { return t.window.empty() ? 1.0 : (double)t.window[i]; };
    if(out.packed) {
        // a_j = (win_2j x_2j + i win_2j+1 x_2j+1) conj(w_j) = x_2j (win_2j conj w_j) + x_2j+1 (i win_2j+1 conj w_j)
        out.a.assign((size_t)2 * L, cfloat{0.0f, 0.0f});
        for(uint32_t j = 0; j < np; ++j) {
            double cr, ci;
            chirp(j, cr, ci);
            const double w0 = win(2 * j), w1 = win(2 * j + 1);
            out.a[(size_t)2 * j] = cfloat{(float)(w0 * cr), (float)(-w0 * ci)};
            out.a[(size_t)2 * j + 1] = cfloat{(float)(w1 * ci), (float)(w1 * cr)}; // i * (cr - i ci) = ci + i cr
        }
    } else {
        out.a.assign(L, cfloat{0.0f, 0.0f});
        for(uint32_t j = 0; j < n; ++j) {
            double cr, ci;
            chirp(j, cr, ci);
            out.a[j] = cfloat{(float)(win(j) * cr), (float)(-win(j) * ci)};
        }
    }
    std::vector<double> br(L, 0.0), bi(L, 0.0);
    // lags k - j: packed wants every k < np, i.e. -(np-1) .. np-1 (L >= 2 np - 1); direct wants k < n/2: -(n-1) .. n/2-1 (L >= 3n/2)
    const uint32_t pos = out.packed ? np : std::min(n, L - n + 1u);
    for(uint32_t m = 0; m < pos; ++m)
        chirp(m, br[m], bi[m]);
    for(uint32_t m = 1; m < np; ++m) // negative lags, wrapped
        chirp(m, br[L - m], bi[L - m]);
    fft_double(br, bi);
    out.b.resize(L);
    for(uint32_t k = 0; k < L; ++k)
        out.b[k] = cfloat{(float)br[k], (float)bi[k]};
    if(out.packed) {
        out.q.resize(np);
        out.qr.resize(np);
        out.w.resize(np);
        const double two_pi = 2.0 * pi;
        for(uint32_t k = 0; k < np; ++k) {
            double cr, ci;
            chirp(k, cr, ci);
            out.q[k] = cfloat{(float)(cr / (double)L), (float)(-ci / (double)L)};
            const double a = -two_pi * (double)k / (double)n;
            out.w[k] = cfloat{(float)std::cos(a), (float)std::sin(a)};
        }
        for(uint32_t k = 0; k < np; ++k)
            out.qr[k] = out.q[(np - k) % np];
    }
}

#ifndef WF_MR_PASS_COST
#define WF_MR_PASS_COST 24.0
#endif
int plan_mixed_radix(uint32_t np, uint32_t threads, int radix[4], uint64_t bluestein_points)
{
    static const int kRadices[] = {25, 23, 20, 19, 17, 16, 15, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    radix[0] = radix[1] = radix[2] = radix[3] = 0;
    if(np < 4)
        return 0;
    // what is left of np beyond the primes the radix set is made of: nothing, or ONE prime of 29 .. 127, which becomes the
    // first pass (mr_pass_prime: the DFTs by the definition); the rest of np is then planned behind it
    uint32_t lead = np;
    for(uint32_t p : {2u, 3u, 5u, 7u, 11u, 13u, 17u, 19u, 23u})
        while(lead % p == 0)
            lead /= p;
    if(lead != 1) {
        bool prime = lead >= 29 && lead <= 127;
        for(uint32_t d = 2; prime && d * d <= lead; ++d)
            prime = lead % d != 0;
        if(!prime || (np / lead) % 4u != 0) // (the prime pass takes its butterflies four at a time, 16-byte aligned)
            return 0;
        // p^2 multiply-adds per butterfly against Bluestein's two transforms of L points: measured on MI355X the
        // kernel's time goes like np (57 + p) here and like 66 L there -- N = 7808 (p = 61, L = 8192) +17 %, 6208 (97) +-0,
        // 8128 (127) -30 %
        if(bluestein_points && (uint64_t)np * (57u + lead) > 60u * bluestein_points)
            return 0;
    }
    const uint32_t rest = np / lead;    // planned from kRadices
    const bool led = lead != 1;         // ... behind a prime first pass: at most three more, none above 16
    const int max_n = led ? 3 : 4, min_n = led ? 1 : 2;
    int best[4] = {0, 0, 0, 0}, best_n = 0, best_min = 0, cur[4];
    bool best_small = false;
    double best_cost = 0.0;
    // Which factorisation, in which order?  Every pass is `for(j = t; j < np / R; j += threads)` over in-register DFTs of R points
    // (wf_mixed.hpp): a wavefront pays for ceil((np / R) / threads) butterflies per pass whether its lanes are busy or not, so
    // few passes of large radices -- round 3's rule -- leave most lanes idle once np / R drops below the thread count
    // (np = 400 on 64 threads as 25 x 16: 16 and 25 lanes busy; as 5 x 8 x 10: 80 / 128, 50 / 64, 40 / 64 -- and less than half
    // the arithmetic per lane).  The cost of an order = the instructions one thread issues: per pass, butterflies per thread x
    // (the DFT's arithmetic + 4 per twiddle + its LDS reads and writes, weighted) + a barrier's worth.  Constraints as before: a
    // radix above 16 only in front (it has no twiddled form) and none behind a prime first pass; the last pass one butterfly per
    // thread (np / R <= threads), R <= 16; the first pass's stores go out with stride R cf: even strides conflict in the LDS banks.
    auto dft_cost = [](int R) -> double { // VALU instructions of MrDft<R>::run, counted from its source
        switch(R) {
        case 2: return 4; case 3: return 14; case 4: return 16; case 5: return 40; case 6: return 48; case 7: return 60; case 8: return 58;
        case 9: return 100; case 10: return 116; case 11: return 140; case 12: return 128; case 13: return 192; case 15: return 222;
        case 16: return 160; case 17: return 320; case 19: return 396; case 20: return 288; case 23: return 572; default: return 464; // 25
        }
    };
    auto order_cost = [&](const int *order, int n) -> double {
        double c = 0.0;
        for(int s = 0; s < n; ++s) {
            const int R = order[s];
            const uint32_t nb = np / (uint32_t)R;
            const double iters = (double)((nb + threads - 1u) / threads);
            double per = dft_cost(R) + 3.0 * (2.0 * R);             // an LDS access ~ three arithmetic instructions of issue + wait
            if(s > 0 || led)
                per += 4.0 * (R - 1) + 2.0 * (R - 1);               // twiddles: complex multiplications + their (cached) loads
            if(s == 0 && !led) {
                int g = 1;
                while(g < 16 && R % (2 * g) == 0)
                    g *= 2;
                per += 1.5 * R * (g - 1);                           // first-pass stores, stride R: g-way bank conflicts
            }
            c += iters * per + WF_MR_PASS_COST;                      // + the pass's barrier and loop set-up
        }
        return c;
    };
    auto consider_by_cost = [&](int n) {
        int big = 0;
        for(int i = 0; i < n; ++i)
            big += cur[i] > 16;
        if(big > (led ? 0 : 1))
            return;
        int perm[4] = {0, 1, 2, 3};
        std::sort(perm, perm + n);
        do {
            int order[4];
            for(int i = 0; i < n; ++i)
                order[i] = cur[perm[i]];
            bool ok = order[n - 1] <= 16 && np / (uint32_t)order[n - 1] <= threads;
            for(int i = 1; i < n && ok; ++i)
                ok = order[i] <= 16;
            if(!ok)
                continue;
            double c = order_cost(order, n);
            if(mr_small_radices(order, n))
                c *= 0.87; // (the instantiation that carries only these radices keeps a fifth wave per SIMD: +11 .. +16 % measured)
            if(best_n == 0 || c < best_cost - 1e-9) {
                best_n = n;
                best_cost = c;
                for(int i = 0; i < n; ++i)
                    best[i] = order[i];
            }
        } while(std::next_permutation(perm, perm + n));
    };
    // Spectra of several wavefronts (threads > 64) keep round 3's rule -- the fewest passes, then the largest smallest radix: every
    // pass ends in a workgroup barrier there, and measured on MI355X the cost model's plans are no better (N = 1536 / 1920 on 128
    // threads: -6 % with a fourth pass, +-0 without) or worse (N = 6000 on 512 threads as 5 x 10 x 10 x 6 instead of 25 x 10 x 12:
    // -5.5 %; 4160: -1.5 %).  One wavefront per spectrum (N <= 1024): N = 800 1210 -> 958 vector instructions per spectrum,
    // +2.8 %; N = 960 +3 % (profiles/r04g_n800_phases.txt).
    auto consider_few_passes = [&](int n) {
        // order: a radix above 16 must be first (at most one, none behind a prime first pass); the last one is the largest radix
        // <= 16 with np / R <= threads; the first one otherwise an odd radix (its stores go out with stride R: conflict-free when
        // R is odd)
        int order[4], used[4] = {0, 0, 0, 0}, big = 0;
        for(int i = 0; i < n; ++i)
            big += cur[i] > 16;
        if(big > (led ? 0 : 1))
            return;
        int last = -1;
        for(int i = 0; i < n; ++i)
            if(cur[i] <= 16 && np / (uint32_t)cur[i] <= threads && (last < 0 || cur[i] > cur[last]))
                last = i;
        if(last < 0)
            return;
        used[last] = 1;
        int k = 0;
        if(n >= 2) {
            int first = -1;
            for(int i = 0; i < n; ++i)
                if(!used[i] && cur[i] > 16)
                    first = i;
            if(first < 0 && !led)
                for(int i = 0; i < n; ++i)
                    if(!used[i] && (cur[i] & 1) && (first < 0 || cur[i] > cur[first]))
                        first = i;
            if(first < 0)
                for(int i = 0; i < n; ++i)
                    if(!used[i] && (first < 0 || cur[i] > cur[first]))
                        first = i;
            used[first] = 1;
            order[k++] = cur[first];
        }
        for(int i = 0; i < n; ++i)
            if(!used[i])
                order[k++] = cur[i];
        order[k++] = cur[last];
        int mn = order[0];
        for(int i = 1; i < n; ++i)
            mn = std::min(mn, order[i]);
        // on the containers of two and four wavefronts (128 / 256 threads) a plan made of the small radices runs on the instantiation that carries only
        // those, a fifth wave per SIMD (mr_small_radices: N = 1600 +14 %, 1280 +16 %): among plans of equally many passes it goes first
        const bool small = threads <= 256u && !led && mr_small_radices(order, n);
        if(best_n == 0 || n < best_n || (n == best_n && (small > best_small || (small == best_small && mn > best_min)))) {
            best_n = n;
            best_min = mn;
            best_small = small;
            for(int i = 0; i < n; ++i)
                best[i] = order[i];
        }
    };
    auto consider = [&](int n) {
        if(threads <= 64u)
            consider_by_cost(n);
        else
            consider_few_passes(n);
    };
    auto rec = [&](auto &&self, uint32_t left, int depth, int max_idx) -> void {
        if(left == 1) {
            if(depth >= min_n)
                consider(depth);
            return;
        }
        if(depth == max_n)
            return;
        for(int i = max_idx; i < (int)(sizeof(kRadices) / sizeof(kRadices[0])); ++i)
            if(left % (uint32_t)kRadices[i] == 0) {
                cur[depth] = kRadices[i];
                self(self, left / (uint32_t)kRadices[i], depth + 1, i);
            }
    };
    rec(rec, rest, 0, 0);
    if(best_n == 0)
        return 0;
    int o = 0;
    if(led)
        radix[o++] = (int)lead;
    for(int i = 0; i < best_n; ++i)
        radix[o++] = best[i];
    return o;
}

void build_prime_twiddles(int p, size_t entries, std::vector<cfloat> &wp)
{
    const double two_pi = 6.283185307179586476925286766559;
    wp.assign(std::max(entries, (size_t)p), cfloat{1.0f, 0.0f});
    for(int m = 0; m < p; ++m) {
        const double a = -two_pi * (double)m / (double)p;
        wp[(size_t)m] = cfloat{(float)std::cos(a), (float)std::sin(a)};
    }
}

void build_mixed_radix_tables(uint32_t n, int passes, const int radix[4], std::vector<cfloat> &tw, int tw_off[4], std::vector<cfloat> &w)
{
    const double two_pi = 6.283185307179586476925286766559;
    const uint32_t np = n / 2;
    tw.clear();
    uint64_t ns = 1;
    for(int s = 0; s < 4; ++s) {
        tw_off[s] = (int)tw.size();
        if(s >= passes)
            continue;
        const uint64_t R = (uint64_t)radix[s], len = ns * R;
        if(s >= 1)
            for(uint64_t k = 0; k < R; ++k)
                for(uint64_t jm = 0; jm < ns; ++jm) {
                    const double a = -two_pi * (double)((k * jm) % len) / (double)len;
                    tw.push_back(cfloat{(float)std::cos(a), (float)std::sin(a)});
                }
        ns = len;
    }
    w.resize(np);
    for(uint32_t m = 0; m < np; ++m) {
        const double b = -two_pi * (double)m / (double)n;
        w[m] = cfloat{(float)std::cos(b), (float)std::sin(b)};
    }
}

void build_twiddles(int M, int R1, int R2, int R3, std::vector<cfloat> &tw1, std::vector<cfloat> &tw2, std::vector<cfloat> &tws)
{
    const double two_pi = 6.283185307179586476925286766559;
    const int M1 = M / R1;
    tw1.resize((size_t)M);
    for(int k1 = 0; k1 < R1; ++k1)
        for(int np = 0; np < M1; ++np) {
            const double a = -two_pi * (double)(((int64_t)np * k1) % M) / (double)M;
            tw1[(size_t)(k1 * M1 + np)] = {(float)std::cos(a), (float)std::sin(a)};
        }
    const int M2 = R2 * R3;
    tw2.resize((size_t)M2);
    for(int k2 = 0; k2 < R2; ++k2)
        for(int n3 = 0; n3 < R3; ++n3) {
            const double a = -two_pi * (double)((n3 * k2) % M2) / (double)M2;
            tw2[(size_t)(k2 * R3 + n3)] = {(float)std::cos(a), (float)std::sin(a)};
        }
    tws.resize((size_t)M);
    for(int k = 0; k < M; ++k) {
        const double a = -two_pi * (double)k / (double)(2 * M);
        tws[(size_t)k] = {(float)std::cos(a), (float)std::sin(a)};
    }
}

} // namespace wf

extern "C" void wf_config_defaults(wf_config *cfg)
{
    // get_defaults, src/source.cpp:119-174
    *cfg = wf_config{};
    cfg->fft_size = 4096;
    cfg->sample_rate = 48000;
    cfg->capture_channels = 2;
    cfg->stereo = 0;               // channel_mode "mono"
    cfg->window = WF_WINDOW_HANN;
    cfg->sine_exponent = 2;
    cfg->tsmoothing = WF_TSMOOTH_EXPONENTIAL;
    cfg->gravity = 0.65f;
    cfg->fast_peaks = 0;
    cfg->slope = 0.0f;
    cfg->rolloff_q = 0.0f;
    cfg->rolloff_rate = 0.0f;
    cfg->cutoff_low = 30;
    cfg->cutoff_high = 17500;
    cfg->floor_db = -65;
    cfg->ceiling_db = 0;
    cfg->normalize_volume = 0;
    cfg->volume_target = -8.0f;
    cfg->max_gain = 30.0f;
    cfg->bars = 0;                 // display_mode "curve"
    cfg->interp_mode = WF_INTERP_CATROM;
    cfg->log_scale = 1;
    cfg->mirror_freq_axis = 0;
    cfg->width = 800;
    cfg->height = 225;
    cfg->bar_width = 24;
    cfg->bar_gap = 6;
    cfg->channel_spacing = 0;
    cfg->min_bar_height = 0;
    cfg->rounded_caps = 0;
    cfg->curve = 0;                // no render-time outputs unless asked for
    cfg->filter_mode = WF_FILTER_NONE;
    cfg->filter_radius = 1.5f;
    cfg->meter = 0;
    cfg->meter_rms = 1;            // P_RMS_MODE default true
    cfg->meter_ms = 150;           // P_METER_BUF default
    cfg->waveform = 0;
    cfg->vertices = 0;
    cfg->step_width = 8;           // get_defaults, src/source.cpp:163-164
    cfg->step_gap = 4;
    cfg->radial = 0;               // P_RADIAL default
}
