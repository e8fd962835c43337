/*
 * wf_oracle.c -- see wf_oracle.h.  TEST INFRASTRUCTURE (CPU restatement of the reference path).
 * Build: gcc -O2 -std=c11 -ffp-contract=off (no FMA contraction: the reference's generic TU is
 * built without -mfma and with -std=c++20, i.e. -ffp-contract=off).
 */
#include "wf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI_F 3.14159274101257324219f /* std::numbers::pi_v<float> */

struct wfo_source {
    wf_config cfg;
    uint32_t n, m;              /* m_fft_size, m_fft_size / 2 */
    uint32_t cap_ch, out_ch;    /* m_capture_channels, m_output_channels */
    /* rings (CircularBuffer, in samples) */
    float *ring[2];
    size_t ring_len[2], ring_cap[2];
    uint32_t sync_delay;
    int hidden;
    int last_silent;            /* m_last_silent */
    float input_rms;
    /* volume normalisation producer (capture_audio's RMS part + update_input_rms), only with cfg.normalize_volume */
    float *rms_sync; size_t rms_sync_len, rms_sync_cap; /* m_rms_sync_buf (CircularBuffer of squared peaks), in samples */
    float *rms_buf; size_t rms_size, rms_pos;           /* m_input_rms_buf (circular), m_input_rms_size, m_input_rms_pos */
    /* tables */
    float *window;              /* NULL when FFTWindow::NONE */
    float window_sum;
    float *slope;               /* NULL unless m_slope > 0 */
    float *rolloff;             /* NULL unless rolloff active */
    /* buffers */
    float *fft_in;              /* m_fft_input */
    float *fft_out;             /* m_fft_output: interleaved re,im, m bins */
    float *tsmooth[2];
    float *decibels[2];
    /* bars */
    int num_bars;
    float *interp_indices; size_t n_indices;
    int *band_widths;
    float *weights; int radius, taps;
    float *bars[2];
    float border_top, border_bottom, cpos;
    float *gauss; int gauss_radius; float gauss_sum; float *filter_tmp; /* m_kernel (Gaussian), apply_filter's output buffer */
    /* FFT work */
    double *wr, *wi;            /* work arrays, n each */
    double *twr, *twi;          /* exp(-2 pi i k / n), k < n/2 */
};

/* ---- math_funcs.hpp ------------------------------------------------------------------------- */
static float log_interp_f(float a, float b, float t) { return a * powf(b / a, t); } /* :25-29 */
/* std::lerp as libstdc++ implements it (src/math_funcs.hpp:31-35 forwards to it) */
static float lerp_f(float a, float b, float t)
{
    if((a <= 0 && b >= 0) || (a >= 0 && b <= 0))
        return t * b + (1 - t) * a;
    if(t == 1)
        return b;
    const float x = a + t * (b - a);
    return ((t > 1) == (b > a)) ? (b < x ? x : b) : (b > x ? x : b);
}
static float sinc_f(float x) /* :37-44 */
{
    if(x == 0.0f)
        return 1.0f;
    const float tmp = PI_F * x;
    return sinf(tmp) / tmp;
}
static float lanczos_f(float x, float w) /* :46-52 */
{
    if(fabsf(x) < w)
        return sinc_f(x) * sinc_f(x / w);
    return 0.0f;
}
static float clamp_f(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; } /* std::clamp */

float wfo_db_min(void) { return 20.0f * log10f(FLT_MIN); } /* src/source.cpp:43 */
static float dbfs(float mag) { return (mag > 0.0f) ? 20.0f * log10f(mag) : wfo_db_min(); } /* src/source.hpp:293-299 */

/* get_gravity, src/source.hpp:301-312 */
float wfo_gravity(const wfo_source *s, float seconds)
{
    const float denom = 0.03868924705242879469662125316986f;
    const float hi = denom * 5.0f;
    const float lo = 0.0f;
    if((s->cfg.tsmoothing == WF_TSMOOTH_NONE) || (s->cfg.gravity <= 0.0f))
        return 0.0f;
    return (s->cfg.tsmoothing == WF_TSMOOTH_TVEXPONENTIAL) ? expf(-seconds / lerp_f(lo, hi, s->cfg.gravity)) : s->cfg.gravity;
}

/* ---- the DFT stage (stands in for fftwf_execute of an r2c plan) ---------------------------------- */
static void fft_prepare(wfo_source *s)
{
    const uint32_t n = s->n;
    s->wr = (double *)malloc(sizeof(double) * n);
    s->wi = (double *)malloc(sizeof(double) * n);
    /* the full circle exp(-2 pi i k / n); the power-of-two path uses its first half */
    s->twr = (double *)malloc(sizeof(double) * n);
    s->twi = (double *)malloc(sizeof(double) * n);
    for(uint32_t k = 0; k < n; ++k) {
        const double a = -2.0 * 3.14159265358979323846264338327950288 * (double)k / (double)n;
        s->twr[k] = cos(a);
        s->twi[k] = sin(a);
    }
}

/* iterative radix-2 decimation-in-time complex FFT in double; n is a power of two */
static void fft_double(double *re, double *im, const double *twr, const double *twi, uint32_t n)
{
    for(uint32_t i = 1, j = 0; i < n; ++i) {
        uint32_t bit = n >> 1;
        for(; j & bit; bit >>= 1)
            j ^= bit;
        j ^= bit;
        if(i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for(uint32_t len = 2; len <= n; len <<= 1) {
        const uint32_t half = len >> 1, step = n / len;
        for(uint32_t i = 0; i < n; i += len)
            for(uint32_t k = 0; k < half; ++k) {
                const double cr = twr[k * step], ci = twi[k * step];
                const double xr = re[i + k + half] * cr - im[i + k + half] * ci;
                const double xi = re[i + k + half] * ci + im[i + k + half] * cr;
                re[i + k + half] = re[i + k] - xr;
                im[i + k + half] = im[i + k] - xi;
                re[i + k] += xr;
                im[i + k] += xi;
            }
    }
}

/* Any other length (the reference takes every multiple of 16 in [128, 65536]): the same DFT sum, evaluated in double by
 * recursive decimation in time over the prime factors of n (smallest first); a prime length is the plain sum.  tw holds
 * the full circle exp(-2 pi i j / root), the sub-transform of length n uses every (root / n)-th entry.  In place on
 * (re, im)[0 .. n) read with `stride`; the result is contiguous in (yr, yi). */
static void dft_any(const double *xr, const double *xi, size_t stride, uint32_t n, double *yr, double *yi, const double *twr,
                    const double *twi, uint32_t root)
{
    uint32_t p = n;
    for(uint32_t f = 2; (uint64_t)f * f <= n; ++f)
        if(n % f == 0) {
            p = f;
            break;
        }
    const uint32_t m = n / p, step = root / n;
    if(m == 1) { /* prime (or 1): X[k] = sum_j x[j] W_n^(jk) */
        for(uint32_t k = 0; k < n; ++k) {
            double ar = 0.0, ai = 0.0;
            uint32_t idx = 0; /* (j * k) mod n */
            for(uint32_t j = 0; j < n; ++j) {
                const double cr = twr[(size_t)idx * step], ci = twi[(size_t)idx * step];
                const double vr = xr[j * stride], vi = xi[j * stride];
                ar += vr * cr - vi * ci;
                ai += vr * ci + vi * cr;
                idx += k;
                if(idx >= n)
                    idx -= n;
            }
            yr[k] = ar;
            yi[k] = ai;
        }
        return;
    }
    /* Y_r = DFT_m(x[r], x[r + p], ...) into block r; then X[k + m q] = sum_r (W_n^(r k) Y_r[k]) W_p^(r q) */
    for(uint32_t r = 0; r < p; ++r)
        dft_any(xr + r * stride, xi + r * stride, stride * p, m, yr + (size_t)r * m, yi + (size_t)r * m, twr, twi, root);
    double tr[64], ti[64], *hr = tr, *hi = ti;
    if(p > 64) {
        hr = (double *)malloc(sizeof(double) * 2 * p);
        hi = hr + p;
    }
    for(uint32_t k = 0; k < m; ++k) {
        for(uint32_t r = 0; r < p; ++r) {
            const size_t w = (size_t)(((uint64_t)r * k) % n) * step;
            const double vr = yr[(size_t)r * m + k], vi = yi[(size_t)r * m + k];
            hr[r] = vr * twr[w] - vi * twi[w];
            hi[r] = vr * twi[w] + vi * twr[w];
        }
        for(uint32_t q = 0; q < p; ++q) {
            double ar = 0.0, ai = 0.0;
            for(uint32_t r = 0; r < p; ++r) {
                const size_t w = (size_t)(((uint64_t)r * q) % p) * m * step; /* W_p^(r q) = W_n^(m r q) */
                ar += hr[r] * twr[w] - hi[r] * twi[w];
                ai += hr[r] * twi[w] + hi[r] * twr[w];
            }
            yr[(size_t)q * m + k] = ar;
            yi[(size_t)q * m + k] = ai;
        }
    }
    if(p > 64)
        free(hr);
}

static void r2c(wfo_source *s)
{
    const uint32_t n = s->n;
    if(n & (n - 1)) {
        double *xr = (double *)malloc(sizeof(double) * 4 * (size_t)n);
        double *xi = xr + n, *yr = xi + n, *yi = yr + n;
        for(uint32_t i = 0; i < n; ++i) {
            xr[i] = (double)s->fft_in[i];
            xi[i] = 0.0;
        }
        dft_any(xr, xi, 1, n, yr, yi, s->twr, s->twi, n);
        for(uint32_t k = 0; k < s->m; ++k) {
            s->fft_out[2 * k] = (float)yr[k];
            s->fft_out[2 * k + 1] = (float)yi[k];
        }
        free(xr);
        return;
    }
    for(uint32_t i = 0; i < n; ++i) {
        s->wr[i] = (double)s->fft_in[i];
        s->wi[i] = 0.0;
    }
    fft_double(s->wr, s->wi, s->twr, s->twi, n);
    for(uint32_t k = 0; k < s->m; ++k) {
        s->fft_out[2 * k] = (float)s->wr[k];
        s->fft_out[2 * k + 1] = (float)s->wi[k];
    }
}

void wfo_r2c(const float *in, uint32_t n, float *out_interleaved)
{
    wfo_source tmp;
    memset(&tmp, 0, sizeof(tmp));
    tmp.n = n;
    tmp.m = n / 2;
    tmp.fft_in = (float *)in;
    tmp.fft_out = out_interleaved;
    fft_prepare(&tmp);
    r2c(&tmp);
    free(tmp.wr); free(tmp.wi); free(tmp.twr); free(tmp.twi);
}

/* ---- setup: WAVSource::update() ------------------------------------------------------------------- */
static void build_window(wfo_source *s) /* src/source.cpp:1190-1234 */
{
    const size_t n = s->n;
    if(s->cfg.window == WF_WINDOW_NONE) {
        s->window = NULL;
        s->window_sum = (float)n;
        return;
    }
    s->window = (float *)malloc(sizeof(float) * n);
    const size_t N = n - 1;
    const float pi = PI_F;
    const float pi2 = 2 * pi, pi4 = 4 * pi, pi6 = 6 * pi;
    for(size_t i = 0; i < n; ++i) {
        float w;
        switch(s->cfg.window) {
        case WF_WINDOW_HAMMING:
            w = 0.53836f - (0.46164f * cosf((pi2 * i) / N));
            break;
        case WF_WINDOW_BLACKMAN:
            w = 0.42f - (0.5f * cosf((pi2 * i) / N)) + (0.08f * cosf((pi4 * i) / N));
            break;
        case WF_WINDOW_BLACKMAN_HARRIS:
            w = 0.35875f - (0.48829f * cosf((pi2 * i) / N)) + (0.14128f * cosf((pi4 * i) / N)) - (0.01168f * cosf((pi6 * i) / N));
            break;
        case WF_WINDOW_POWER_OF_SINE:
            w = powf(sinf((pi * i) / N), (float)s->cfg.sine_exponent);
            break;
        case WF_WINDOW_HANN:
        default:
            w = 0.5f * (1 - cosf((pi2 * i) / N));
            break;
        }
        s->window[i] = w;
    }
    float sum = 0.0f;
    for(size_t i = 0; i < n; ++i)
        sum += s->window[i];
    s->window_sum = sum;
}

static void build_slope(wfo_source *s) /* src/source.cpp:1282-1290 */
{
    s->slope = NULL;
    if(!(s->cfg.slope > 0.0f))
        return;
    const size_t num_mods = s->m;
    const float maxmod = (float)(num_mods - 1);
    s->slope = (float *)malloc(sizeof(float) * num_mods);
    for(size_t i = 0; i < num_mods; ++i)
        s->slope[i] = log10f(log_interp_f(10.0f, 10000.0f, ((float)i * s->cfg.slope) / maxmod));
}

static void build_rolloff(wfo_source *s) /* init_rolloff, src/source.cpp:898-918 */
{
    s->rolloff = NULL;
    if(!((s->cfg.rolloff_q > 0.0f) && (s->cfg.rolloff_rate > 0.0f)))
        return;
    const size_t sz = s->m;
    const float sr = (float)s->cfg.sample_rate;
    const float coeff = sr / (float)s->n;
    const float ratio = exp2f(s->cfg.rolloff_q);
    const float freq_low = (float)s->cfg.cutoff_low * ratio;
    const float freq_high = (float)s->cfg.cutoff_high / ratio;
    s->rolloff = (float *)malloc(sizeof(float) * sz);
    s->rolloff[0] = 0.0f;
    for(size_t i = 1u; i < sz; ++i) {
        const float freq = i * coeff;
        const float ratio_low = freq_low / freq;
        const float ratio_high = freq / freq_high;
        const float low_attenuation = (ratio_low > 1.0f) ? (s->cfg.rolloff_rate * log2f(ratio_low)) : 0.0f;
        const float high_attenuation = (ratio_high > 1.0f) ? (s->cfg.rolloff_rate * log2f(ratio_high)) : 0.0f;
        s->rolloff[i] = low_attenuation + high_attenuation;
    }
}

static void build_bars(wfo_source *s) /* update() :1267-1276, init_interp :837-896, render_bars :1476-1494 */
{
    const wf_config *c = &s->cfg;
    s->num_bars = 0;
    const int curve = !c->bars && c->curve; /* render_curve: init_interp(m_width), one point per pixel column */
    if(!c->bars && !curve)
        return;
    int num_bars;
    unsigned int sz;
    if(curve) {
        num_bars = (int)c->width;
        sz = c->width;
    } else {
        const int bar_stride = c->bar_width + c->bar_gap;
        num_bars = (int)(c->width / (unsigned int)bar_stride);
        if(((int)c->width - (num_bars * bar_stride)) >= c->bar_width)
            ++num_bars;
        sz = (unsigned int)(num_bars + 1);
    }
    s->num_bars = num_bars;
    const size_t maxbin = (size_t)s->m - 1;
    const float sr = (float)c->sample_rate;
    const float lowbin = clamp_f((float)c->cutoff_low * (float)s->n / sr, 1.0f, (float)maxbin);
    const float highbin = clamp_f((float)c->cutoff_high * (float)s->n / sr, 1.0f, (float)maxbin);
    float *idx = (float *)malloc(sizeof(float) * sz);
    for(unsigned int i = 0u; i < sz; ++i) {
        const float t = (c->mirror_freq_axis ? i * 2.0f : (float)i) / (float)(sz - 1);
        const float v = c->log_scale ? log_interp_f(lowbin, highbin, t) : lerp_f(lowbin, highbin, t);
        idx[i] = clamp_f(v, lowbin, highbin);
    }
    s->band_widths = (int *)malloc(sizeof(int) * (size_t)num_bars);
    size_t total = 0;
    for(int i = 0; i < num_bars; ++i) {
        const int w = curve ? 1 : (int)(idx[i + 1] - idx[i]);
        s->band_widths[i] = (w > 1) ? w : 1;
        total += (size_t)s->band_widths[i];
    }
    if(c->interp_mode != WF_INTERP_POINT) {
        if(curve) { /* the curve interpolates at the indices themselves (:874-876) */
            s->interp_indices = idx;
            s->n_indices = total = sz;
        } else {    /* bars: one position per band sample (:878-889) */
            s->interp_indices = (float *)malloc(sizeof(float) * total);
            s->n_indices = total;
            size_t k = 0;
            for(int i = 0; i < num_bars; ++i)
                for(int j = 0; j < s->band_widths[i]; ++j)
                    s->interp_indices[k++] = idx[i] + j;
            free(idx);
        }
        if(c->interp_mode == WF_INTERP_LANCZOS) { /* make_lanczos_kernel, src/filter.hpp:106-131 */
            const intmax_t radius = 4;
            s->radius = 4;
            s->taps = 8;
            s->weights = (float *)malloc(sizeof(float) * total * 8);
            for(size_t i = 0; i < total; ++i) {
                const float x = s->interp_indices[i];
                const intmax_t ix = (intmax_t)x;
                const intmax_t start = ix - radius + 1, stop = ix + radius;
                for(intmax_t j = start; j <= stop; ++j)
                    s->weights[i * 8 + (size_t)(j - start)] = lanczos_f(x - j, (float)radius);
            }
        } else { /* make_catrom_kernel, src/filter.hpp:67-104, t = 0.5 */
            const float t = 0.5f;
            const float matrix[4][4] = {{0, -t, 2 * t, -t}, {1, 0, t - 3, 2 - t}, {0, t, 3 - (2 * t), t - 2}, {0, 0, -t, t}};
            s->radius = 2;
            s->taps = 4;
            s->weights = (float *)malloc(sizeof(float) * total * 4);
            for(size_t i = 0; i < total; ++i) {
                const float u = s->interp_indices[i] - floorf(s->interp_indices[i]);
                const float row[4] = {1, u, u * u, u * u * u};
                for(int j = 0; j < 4; ++j) {
                    float sum = 0;
                    for(int k = 0; k < 4; ++k)
                        sum += row[k] * matrix[j][k];
                    s->weights[i * 4 + (size_t)j] = sum;
                }
            }
        }
    } else {
        s->interp_indices = idx;
        s->n_indices = sz;
    }
    for(int ch = 0; ch < 2; ++ch)
        s->bars[ch] = (float *)calloc((size_t)num_bars, sizeof(float));
    /* render_bars geometry, src/source.cpp:1480-1494 */
    const float center = (float)c->height / 2;
    const float bottom = (float)c->height;
    const float cpos = c->stereo ? center : bottom;
    const float cap_radius = (float)c->bar_width / 2.0f;
    const float channel_offset = c->channel_spacing * 0.5f;
    float border_top = c->rounded_caps ? cap_radius : 0.0f;
    float border_bottom = (c->rounded_caps && (!c->stereo || (c->channel_spacing > 0))) ? cpos - cap_radius : cpos;
    if(c->channel_spacing > 0)
        border_bottom -= channel_offset;
    if(c->min_bar_height > 0)
        border_bottom -= c->min_bar_height;
    border_bottom = clamp_f(border_bottom, border_top, cpos);
    if(curve) { /* render_curve maps onto lerp(0, cpos - channel_offset, .), src/source.cpp:1411 */
        border_top = 0.0f;
        border_bottom = cpos - channel_offset;
    }
    s->border_top = border_top;
    s->border_bottom = border_bottom;
    s->cpos = cpos;
    /* m_kernel = make_gauss_kernel(m_filter_radius), src/source.cpp:1279-1280, src/filter.hpp:40-65 */
    s->gauss = NULL;
    s->gauss_radius = 0;
    s->gauss_sum = 0.0f;
    if(c->filter_mode == WF_FILTER_GAUSS) {
        float sigma = fabsf(c->filter_radius);
        if(sigma < 0.01f)
            sigma = 0.01f;
        const int w = (int)ceilf(3.0f * sigma);
        const int size = (2 * w) - 1;
        s->gauss = (float *)malloc(sizeof(float) * (size_t)size);
        s->gauss_radius = w;
        const float pi2 = 3.14159265358979323846f * 2.0f;
        const float sigsqr = sigma * sigma;
        const float expdenom = 2.0f * sigsqr;
        const float coeff = (1.0f / (sqrtf(pi2) * sigma));
        int j = 0;
        for(int i = -w + 1; i < w; ++i) {
            const float exponent = -((float)(i * i) / expdenom);
            const float weight = coeff * expf(exponent);
            s->gauss[j++] = weight;
            s->gauss_sum += weight;
        }
        s->filter_tmp = (float *)calloc((size_t)num_bars, sizeof(float));
    }
}

/* ---- CircularBuffer (src/circular_buffer.hpp), in samples ------------------------------------------ */
static void ring_push(wfo_source *s, int ch, const float *src, size_t frames)
{
    if(s->ring_len[ch] + frames > s->ring_cap[ch]) {
        size_t cap = s->ring_cap[ch] ? s->ring_cap[ch] : 1024;
        while(cap < s->ring_len[ch] + frames)
            cap *= 2;
        s->ring[ch] = (float *)realloc(s->ring[ch], cap * sizeof(float));
        s->ring_cap[ch] = cap;
    }
    if(src != NULL)
        memcpy(s->ring[ch] + s->ring_len[ch], src, frames * sizeof(float));
    else
        memset(s->ring[ch] + s->ring_len[ch], 0, frames * sizeof(float)); /* push_back_zero */
    s->ring_len[ch] += frames;
}
static void ring_pop_front(wfo_source *s, int ch, size_t frames)
{
    if(frames > s->ring_len[ch])
        frames = s->ring_len[ch];
    memmove(s->ring[ch], s->ring[ch] + frames, (s->ring_len[ch] - frames) * sizeof(float));
    s->ring_len[ch] -= frames;
}

wfo_source *wfo_create(const wf_config *cfg)
{
    if(cfg == NULL || cfg->fft_size < 128 || (cfg->fft_size & 15))
        return NULL; /* src/source.cpp:562-565: at least 128, a multiple of 16 */
    wfo_source *s = (wfo_source *)calloc(1, sizeof(*s));
    s->cfg = *cfg;
    /* get_settings()' repairs, src/source.cpp:567-579 */
    if((s->cfg.cutoff_high - s->cfg.cutoff_low) < 0) {
        s->cfg.cutoff_high = 17500;
        s->cfg.cutoff_low = 120;
    }
    if((s->cfg.ceiling_db - s->cfg.floor_db) < 1) {
        s->cfg.ceiling_db = 0;
        s->cfg.floor_db = -120;
    }
    if(!s->cfg.stereo || (((int)s->cfg.height - s->cfg.channel_spacing) < 1))
        s->cfg.channel_spacing = 0;
    s->n = cfg->fft_size;
    s->m = cfg->fft_size / 2;
    s->cap_ch = cfg->capture_channels;
    s->out_ch = ((s->cap_ch > 1) || cfg->stereo) ? 2u : 1u; /* src/source.cpp:1171 */
    for(uint32_t i = 0; i < s->out_ch; ++i) {             /* :1172-1182 */
        s->decibels[i] = (float *)malloc(sizeof(float) * s->m);
        for(uint32_t k = 0; k < s->m; ++k)
            s->decibels[i][k] = wfo_db_min();
        if(cfg->tsmoothing != WF_TSMOOTH_NONE)
            s->tsmooth[i] = (float *)calloc(s->m, sizeof(float));
    }
    s->fft_in = (float *)calloc(s->n, sizeof(float));
    s->fft_out = (float *)calloc((size_t)s->n * 2, sizeof(float));
    fft_prepare(s);
    build_window(s);
    s->last_silent = 0;                                   /* :1236 */
    for(uint32_t i = 0; i < s->cap_ch; ++i)                /* :1243-1248: N samples of silence */
        ring_push(s, (int)i, NULL, s->n);
    build_bars(s);
    build_slope(s);
    build_rolloff(s);
    if(cfg->normalize_volume) { /* src/source.cpp:1144-1152 */
        s->input_rms = 0.0f;
        s->rms_size = (size_t)cfg->sample_rate & (size_t)-16;
        s->rms_pos = 0;
        s->rms_buf = (float *)calloc(s->rms_size, sizeof(float));
    }
    return s;
}

void wfo_destroy(wfo_source *s)
{
    if(s == NULL)
        return;
    for(int i = 0; i < 2; ++i) {
        free(s->ring[i]); free(s->tsmooth[i]); free(s->decibels[i]); free(s->bars[i]);
    }
    free(s->window); free(s->slope); free(s->rolloff); free(s->fft_in); free(s->fft_out);
    free(s->interp_indices); free(s->band_widths); free(s->weights); free(s->gauss); free(s->filter_tmp);
    free(s->wr); free(s->wi); free(s->twr); free(s->twi);
    free(s->rms_sync); free(s->rms_buf);
    free(s);
}

void wfo_set_sync_delay(wfo_source *s, uint32_t frames) { s->sync_delay = frames; }
void wfo_set_hidden(wfo_source *s, int hidden) { s->hidden = hidden; }
void wfo_set_input_rms(wfo_source *s, float rms) { s->input_rms = rms; }

/* capture_audio, src/source.cpp:1873-1886 (the A/V-sync amount is supplied by the caller in frames) */
void wfo_push_audio(wfo_source *s, const float *ch0, const float *ch1, uint32_t frames, int muted)
{
    const float *data[2] = {ch0, ch1};
    if(s->cfg.normalize_volume) { /* :1842-1871: the largest |sample| of all channels per frame, squared -- muted or not */
        if(s->rms_sync_len + frames > s->rms_sync_cap) {
            size_t cap = s->rms_sync_cap ? s->rms_sync_cap : 4096;
            while(cap < s->rms_sync_len + frames)
                cap *= 2;
            s->rms_sync = (float *)realloc(s->rms_sync, cap * sizeof(float));
            s->rms_sync_cap = cap;
        }
        for(uint32_t i = 0; i < frames; ++i) {
            float val = 0.0f;
            for(uint32_t ch = 0; ch < s->cap_ch; ++ch)
                if(data[ch] != NULL)
                    val = fmaxf(fabsf(data[ch][i]), val);
            s->rms_sync[s->rms_sync_len + i] = val * val;
        }
        s->rms_sync_len += frames;
        const size_t max_rms_size = (size_t)s->sync_delay + s->rms_size;
        if(s->rms_sync_len > max_rms_size) {
            const size_t drop = s->rms_sync_len - max_rms_size;
            memmove(s->rms_sync, s->rms_sync + drop, max_rms_size * sizeof(float));
            s->rms_sync_len = max_rms_size;
        }
    }
    for(uint32_t j = 0; j < s->cap_ch; ++j) {
        if(muted || data[j] == NULL)
            ring_push(s, (int)j, NULL, frames);
        else
            ring_push(s, (int)j, data[j], frames);
        const size_t max_size = (size_t)s->sync_delay + s->n;
        if(s->ring_len[j] > max_size)
            ring_pop_front(s, (int)j, s->ring_len[j] - max_size);
    }
}

/* WAVSourceGeneric::update_input_rms (src/source_generic.cpp:392-403) with sync_rms_buffer (src/source.cpp:810-835); the
 * reference calls it at the top of every tick() when m_normalize_volume (src/source.cpp:1330-1331) */
void wfo_update_input_rms(wfo_source *s)
{
    if(!s->cfg.normalize_volume)
        return;
    const size_t dtsize = s->sync_delay;
    if(s->rms_sync_len <= dtsize)
        return;
    size_t head = 0;
    while(s->rms_sync_len - head > dtsize) {
        const size_t consume = s->rms_sync_len - head - dtsize;
        const size_t max = s->rms_size - s->rms_pos;
        const size_t n = (consume >= max) ? max : consume;
        memcpy(s->rms_buf + s->rms_pos, s->rms_sync + head, n * sizeof(float));
        head += n;
        s->rms_pos = (consume >= max) ? 0 : s->rms_pos + n;
    }
    memmove(s->rms_sync, s->rms_sync + head, (s->rms_sync_len - head) * sizeof(float));
    s->rms_sync_len -= head;
    float sum = 0.0f;
    for(size_t i = 0; i < s->rms_size; ++i)
        sum += s->rms_buf[i];
    s->input_rms = sqrtf(sum / s->rms_size);
}
float wfo_input_rms(const wfo_source *s) { return s->input_rms; }

/* WAVSourceGeneric::tick_spectrum, src/source_generic.cpp:26-180 */
void wfo_tick(wfo_source *s, float seconds)
{
    const uint32_t outsz = s->m;
    const float DB_MIN = wfo_db_min();
    if(s->hidden) { /* :34-48 */
        if(s->last_silent)
            return;
        for(uint32_t ch = 0; ch < s->cap_ch; ++ch)
            if(s->tsmooth[ch] != NULL)
                memset(s->tsmooth[ch], 0, outsz * sizeof(float));
        for(int ch = 0; ch < (s->cfg.stereo ? 2 : 1); ++ch)
            for(uint32_t i = 0; i < outsz; ++i)
                s->decibels[ch][i] = DB_MIN;
        s->last_silent = 1;
        return;
    }
    const size_t dtsize = (size_t)s->sync_delay + s->n; /* :50-51, in samples */
    unsigned silent_channels = 0;
    for(uint32_t channel = 0; channel < s->cap_ch; ++channel) {
        if(s->ring_len[channel] >= dtsize) { /* :55-59 */
            ring_pop_front(s, (int)channel, s->ring_len[channel] - dtsize);
            memcpy(s->fft_in, s->ring[channel], s->n * sizeof(float));
        } else
            continue;

        int silent = 1; /* :63-72 */
        for(uint32_t i = 0; i < s->n; ++i)
            if(s->fft_in[i] != 0.0f) {
                silent = 0;
                s->last_silent = 0;
                break;
            }
        if(silent) { /* :74-95 */
            if(s->last_silent)
                continue;
            int outsilent = 1;
            const float floor = (float)(s->cfg.floor_db - 10);
            const uint32_t ch = s->cfg.stereo ? channel : 0u;
            for(uint32_t i = 0; i < outsz; ++i)
                if(s->decibels[ch][i] > floor) {
                    outsilent = 0;
                    break;
                }
            if(outsilent) {
                if(++silent_channels >= s->cap_ch)
                    s->last_silent = 1;
                continue;
            }
        }
        if(s->window != NULL) /* :97-103 */
            for(uint32_t i = 0; i < s->n; ++i)
                s->fft_in[i] *= s->window[i];
        r2c(s); /* :105-106 */

        const float mag_coefficient = 2.0f / s->window_sum; /* :110-135 */
        const float g = wfo_gravity(s, seconds);
        const float g2 = 1.0f - g;
        const int slope = s->cfg.slope > 0.0f;
        for(uint32_t i = 0; i < outsz; ++i) {
            const float real = s->fft_out[2 * i];
            const float imag = s->fft_out[2 * i + 1];
            float mag = hypotf(real, imag) * mag_coefficient;
            if(slope)
                mag *= s->slope[i];
            if(s->cfg.tsmoothing != WF_TSMOOTH_NONE) {
                float oldval = s->tsmooth[channel][i];
                if(s->cfg.fast_peaks)
                    oldval = (mag > oldval) ? mag : oldval; /* std::max(mag, oldval) */
                mag = (g * oldval) + (g2 * mag);
                s->tsmooth[channel][i] = mag;
            }
            s->decibels[channel][i] = mag;
        }
    }
    if(s->last_silent) /* :138-139 */
        return;
    if(s->out_ch > s->cap_ch) /* :141-142 */
        memcpy(s->decibels[1], s->decibels[0], outsz * sizeof(float));
    if(s->cfg.stereo) { /* :144-159 */
        for(int ch = 0; ch < 2; ++ch)
            for(uint32_t i = 0; i < outsz; ++i)
                s->decibels[ch][i] = dbfs(s->decibels[ch][i]);
    } else if(s->cap_ch > 1) {
        for(uint32_t i = 0; i < outsz; ++i)
            s->decibels[0][i] = dbfs((s->decibels[0][i] + s->decibels[1][i]) * 0.5f);
    } else {
        for(uint32_t i = 0; i < outsz; ++i)
            s->decibels[0][i] = dbfs(s->decibels[0][i]);
    }
    if(s->cfg.normalize_volume) { /* :161-167 */
        const float a = s->cfg.volume_target - dbfs(s->input_rms);
        const float volume_compensation = (s->cfg.max_gain < a) ? s->cfg.max_gain : a; /* std::min(a, max_gain) */
        for(int ch = 0; ch < (s->cfg.stereo ? 2 : 1); ++ch)
            for(uint32_t i = 1; i < outsz; ++i)
                s->decibels[ch][i] += volume_compensation;
    }
    if((s->cfg.rolloff_q > 0.0f) && (s->cfg.rolloff_rate > 0.0f)) { /* :169-179 */
        for(int ch = 0; ch < (s->cfg.stereo ? 2 : 1); ++ch)
            for(uint32_t i = 1; i < outsz; ++i) {
                const float val = s->decibels[ch][i] - s->rolloff[i];
                s->decibels[ch][i] = (val < DB_MIN) ? DB_MIN : val; /* std::max(val, DB_MIN) */
            }
    }
}

/* kernel_convolve, src/filter.hpp:160-169 */
static float kernel_convolve(const float *samples, size_t sz, const float *weights, int radius, intmax_t index, intmax_t kernel_base)
{
    const intmax_t start = (index - radius) + 1;
    intmax_t stop = index + radius + 1;
    if((intmax_t)sz < stop)
        stop = (intmax_t)sz;
    float sum = 0;
    for(intmax_t i = (start > 0 ? start : 0); i < stop; ++i)
        sum += samples[i] * weights[kernel_base + (i - start)];
    return sum;
}

/* weighted_avg, src/filter.hpp:133-157 */
static float weighted_avg(const float *samples, intmax_t n, const float *weights, int radius, float ksum, intmax_t index)
{
    const intmax_t start = (index - radius) + 1;
    const intmax_t stop = index + radius;
    float sum = 0;
    if((start < 0) || (stop > n)) {
        const intmax_t loopstart = start > 0 ? start : 0;
        const intmax_t loopstop = stop < n ? stop : n;
        float wsum = 0;
        for(intmax_t i = loopstart; i < loopstop; ++i) {
            const float weight = weights[i - start];
            wsum += weight;
            sum += samples[i] * weight;
        }
        return sum / wsum;
    }
    for(intmax_t i = start; i < stop; ++i)
        sum += samples[i] * weights[i - start];
    return sum / ksum;
}

/* render_bars / render_curve up to the vertex fill: interpolation (src/source.cpp:1500-1533 bars, :1380-1394 curve), the
 * optional filter across the outputs (:1396-1405, :1535-1545), the dB -> pixel mapping (:1548-1557, :1407-1417) and the
 * mirror (:1559-1564, :1419-1424) */
void wfo_render_bars(wfo_source *s)
{
    if(s->num_bars <= 0)
        return;
    const int curve = !s->cfg.bars && s->cfg.curve;
    const int dbrange = s->cfg.ceiling_db - s->cfg.floor_db;
    for(int channel = 0; channel < (s->cfg.stereo ? 2 : 1); ++channel) {
        const float *db = s->decibels[channel];
        float *out = s->bars[channel];
        if(curve) {
            if(s->cfg.interp_mode != WF_INTERP_POINT) { /* apply_interp_filter (curve), src/filter.hpp:182-192 */
                const intmax_t d = (intmax_t)s->radius * 2;
                for(intmax_t i = 0, j = 0; i < s->num_bars; ++i, j += d)
                    out[i] = kernel_convolve(db, s->m, s->weights, s->radius, (intmax_t)s->interp_indices[i], j);
            } else { /* :1391-1393 */
                for(int i = 0; i < s->num_bars; ++i)
                    out[i] = db[(int)s->interp_indices[i]];
            }
        } else if(s->cfg.interp_mode != WF_INTERP_POINT) { /* apply_interp_filter (bars), src/filter.hpp:194-211 */
            const intmax_t d = (intmax_t)s->radius * 2;
            intmax_t k = 0, l = 0;
            for(int i = 0; i < s->num_bars; ++i) {
                float sum = 0;
                const intmax_t count = s->band_widths[i];
                for(intmax_t j = 0; j < count; ++j, ++k, l += d)
                    sum += kernel_convolve(db, s->m, s->weights, s->radius, (intmax_t)s->interp_indices[k], l);
                out[i] = sum / (float)count;
            }
        } else { /* :1525-1532 */
            for(int i = 0; i < s->num_bars; ++i) {
                float sum = 0.0f;
                const size_t count = (size_t)s->band_widths[i];
                for(size_t j = 0; j < count; ++j)
                    sum += db[(size_t)s->interp_indices[i] + j];
                out[i] = sum / (float)count;
            }
        }
        if(s->gauss_radius > 0) { /* apply_filter, src/filter.hpp:171-180 */
            for(int i = 0; i < s->num_bars; ++i)
                s->filter_tmp[i] = weighted_avg(out, s->num_bars, s->gauss, s->gauss_radius, s->gauss_sum, i);
            memcpy(out, s->filter_tmp, sizeof(float) * (size_t)s->num_bars);
        }
        for(int i = 0; i < s->num_bars; ++i) /* :1548-1557 / :1407-1417 */
            out[i] = lerp_f(s->border_top, s->border_bottom, clamp_f(s->cfg.ceiling_db - out[i], 0.0f, (float)dbrange) / dbrange);
        if(s->cfg.mirror_freq_axis) { /* :1559-1564 / :1419-1424 */
            const unsigned half = (unsigned)s->num_bars / 2u;
            for(unsigned i = half + 1; i < (unsigned)s->num_bars; ++i)
                out[i] = out[half - (i - half)];
        }
    }
}

/* ---- vertex fill: what render_bars / render_curve write into the vertex buffer for one channel -------------------------- */
static void vset(float *out, size_t cap, size_t i, float x, float y)
{
    if(i < cap) {
        out[4 * i] = x; out[4 * i + 1] = y; out[4 * i + 2] = 0.0f; out[4 * i + 3] = 0.0f; /* vec3_set / vec3_add leave w = 0 */
    }
}

size_t wfo_fill_vertices(const wfo_source *s, int channel, int line, float *out, size_t cap)
{
    if(s->num_bars <= 0)
        return 0;
    const wf_config *c = &s->cfg;
    const int curve = !c->bars && c->curve;
    const float center = (float)c->height / 2;
    const float bottom = (float)c->height;
    const float cpos = c->stereo ? center : bottom;
    const float channel_offset = c->channel_spacing * 0.5f;
    const float *vals = s->bars[channel];
    if(curve) { /* src/source.cpp:1436-1461; x = the column, set once in update() (:1027-1038) */
        float offset = channel_offset;
        if(channel)
            offset = -offset;
        const float bot = cpos - offset;
        for(int i = 0; i < s->num_bars; ++i) {
            const float val = vals[i];
            const float y = channel == 0 ? val : bottom - val;
            if(line)
                vset(out, cap, (size_t)i, (float)i, y);
            else {
                vset(out, cap, (size_t)i * 2, (float)i, y);
                vset(out, cap, (size_t)i * 2 + 1, (float)i, bot);
            }
        }
        return line ? (size_t)s->num_bars : (size_t)s->num_bars * 2; /* gs_draw(.., vbdata->num), :1465, :985 */
    }
    const int bar_stride = c->bar_width + c->bar_gap;
    if(c->vertices == 3) { /* stepped bars, :1583-1607 (m_step_verts: init_steps, :920-933) */
        const int step_stride = c->step_width + c->step_gap;
        size_t max_steps = (size_t)((cpos - channel_offset) / step_stride); /* :1496-1498 */
        if(((int)cpos - (int)(max_steps * step_stride) - (int)channel_offset) > c->step_width)
            ++max_steps;
        const float sx[6] = {0.0f, (float)c->bar_width, 0.0f, (float)c->bar_width, 0.0f, (float)c->bar_width};
        const float sy[6] = {0.0f, 0.0f, (float)c->step_width, 0.0f, (float)c->step_width, (float)c->step_width};
        size_t vertpos = 0;
        for(int i = 0; i < s->num_bars; ++i) {
            const float val = vals[i];
            const float x = (float)(i * bar_stride);
            const float maxheight = (cpos - val - channel_offset);
            for(unsigned j = 0u; j < max_steps; ++j) {
                float y = (float)(j * step_stride);
                if(y >= maxheight)
                    break;
                if(channel)
                    y = cpos + y + channel_offset;
                else
                    y = cpos - y - channel_offset - c->step_width;
                for(int k = 0; k < 6; ++k)
                    vset(out, cap, vertpos + (size_t)k, sx[k] + x, sy[k] + y);
                vertpos += 6;
            }
        }
        return vertpos;
    }
    /* render_bars, plain bars (:1609-1657) */
    const float cap_radius = (float)c->bar_width / 2.0f; /* :1297 */
    int cap_tris = 0;
    float cap_xy[2 * 1024];
    if(c->rounded_caps) { /* m_cap_verts, :1293-1309 */
        const float pi = 3.14159265358979323846f; /* std::numbers::pi_v<float> */
        cap_tris = (int)((2 * pi * cap_radius) / 3.0f);
        if(cap_tris < 4)
            cap_tris = 4;
        if(cap_tris & 1)
            cap_tris += 1;
        if(cap_tris + 1 > 1024)
            return 0;
        const float angle = (2 * pi) / (float)cap_tris;
        for(int j = 0; j < cap_tris + 1; ++j) {
            const float a = j * angle;
            cap_xy[2 * j] = cap_radius * cosf(a);
            cap_xy[2 * j + 1] = cap_radius * sinf(a);
        }
    }
    size_t vertpos = 0;
    for(int i = 0; i < s->num_bars; ++i) {
        float val = vals[i];
        const float x1 = (float)(i * bar_stride);
        const float x2 = x1 + c->bar_width;
        float offset = (c->rounded_caps ? cap_radius : 0.0f) + channel_offset;
        if(channel) {
            val = bottom - val;
            offset = -offset;
        }
        const float bot = ((c->rounded_caps && !c->stereo) || (c->channel_spacing > 0)) ? (cpos - offset) : cpos;
        vset(out, cap, vertpos, x1, val);
        vset(out, cap, vertpos + 1, x2, val);
        vset(out, cap, vertpos + 2, x1, bot);
        vset(out, cap, vertpos + 3, x2, val);
        vset(out, cap, vertpos + 4, x1, bot);
        vset(out, cap, vertpos + 5, x2, bot);
        vertpos += 6;
        if(c->rounded_caps) {
            const float ccx = (float)(i * bar_stride) + cap_radius;
            const int half = cap_tris / 2;
            int start = c->radial ? 0 : (channel ? 0 : half), stop = c->radial ? cap_tris : start + half; /* :1632-1633 */
            for(int j = start; j < stop; ++j) {
                vset(out, cap, vertpos, cap_xy[2 * j] + ccx, cap_xy[2 * j + 1] + val);
                vset(out, cap, vertpos + 1, cap_xy[2 * (j + 1)] + ccx, cap_xy[2 * (j + 1) + 1] + val);
                vset(out, cap, vertpos + 2, ccx, val);
                vertpos += 3;
            }
            if(!c->stereo || (c->channel_spacing > 0)) {
                const float ccy = cpos - offset;
                start = c->radial ? 0 : (channel ? half : 0); /* :1646-1647 */
                stop = c->radial ? cap_tris : start + half;
                for(int j = start; j < stop; ++j) {
                    vset(out, cap, vertpos, cap_xy[2 * j] + ccx, cap_xy[2 * j + 1] + ccy);
                    vset(out, cap, vertpos + 1, cap_xy[2 * (j + 1)] + ccx, cap_xy[2 * (j + 1) + 1] + ccy);
                    vset(out, cap, vertpos + 2, ccx, ccy);
                    vertpos += 3;
                }
            }
        }
    }
    return vertpos; /* gs_draw(GS_TRIS, 0, vertpos), :1663 */
}

/* ---- accessors ------------------------------------------------------------------------------------- */
uint32_t wfo_fft_size(const wfo_source *s) { return s->n; }
uint32_t wfo_output_channels(const wfo_source *s) { return s->out_ch; }
int wfo_last_silent(const wfo_source *s) { return s->last_silent; }
size_t wfo_ring_samples(const wfo_source *s, int ch) { return s->ring_len[ch & 1]; }
const float *wfo_decibels(const wfo_source *s, int ch) { return s->decibels[ch & 1]; }
const float *wfo_tsmooth(const wfo_source *s, int ch) { return s->tsmooth[ch & 1]; }
float *wfo_tsmooth_mut(wfo_source *s, int ch) { return s->tsmooth[ch & 1]; }
const float *wfo_window(const wfo_source *s, float *sum) { if(sum) *sum = s->window_sum; return s->window; }
const float *wfo_slope(const wfo_source *s) { return s->slope; }
const float *wfo_rolloff(const wfo_source *s) { return s->rolloff; }
int wfo_num_bars(const wfo_source *s) { return s->num_bars; }
size_t wfo_interp_indices(const wfo_source *s, const float **out) { *out = s->interp_indices; return s->n_indices; }
size_t wfo_band_widths(const wfo_source *s, const int **out) { *out = s->band_widths; return (size_t)s->num_bars; }
size_t wfo_interp_weights(const wfo_source *s, const float **out, int *radius, int *taps)
{
    *out = s->weights;
    if(radius) *radius = s->radius;
    if(taps) *taps = s->taps;
    return s->weights ? s->n_indices * (size_t)s->taps : 0;
}
const float *wfo_bars(const wfo_source *s, int ch) { return s->bars[ch & 1]; }
