// wf_mixed.hpp -- FFT sizes with small prime factors and at most one prime factor of 17 .. 127, computed directly (device code; also compiled by g++ for
// tests/emu, which replays these functions lane by lane).
//
// The reference takes every multiple of 16 as fft_size (src/source.cpp:562-565) and FFTW gives it an O(n log n) plan for each.
// Here the sizes that are not powers of two ran Bluestein's algorithm (two power-of-two transforms of L >= n - 1 points for the
// n/2-point transform that is wanted: four to eight times the work of a neighbouring power of two).  Most sizes a user meets
// are "smooth", though: the automatic size is sample_rate / fps & -16 (src/source.cpp:1161-1167) -- 800, 1600, 960, 1920, 2000
// at 48 kHz, 1760, 1456, 880, 720, 352 at 44.1 kHz -- and the slider moves in steps of 64.  For n/2 = 2^a 3^b 5^c 7^d 11^e 13^f (times one of 17, 19, 23)
// the packed n/2-point transform is a Stockham autosort FFT of two to four passes whose radices come from
// {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16} and, in the first pass only, {17, 19, 20, 23, 25}:
//   * it lives where Bluestein's first transform lived: the fetch (windowed sample pairs, p1_fetch_blu) and the epilogue (real split
//     with W_n^k, slope, smoothing: p4_direct) are the Bluestein path's own, with tables that carry the plain window and ones;
//   * the exchange buffer of the power-of-two container (M >= n - 1 complex points) is two halves of >= n/2 points: a pass reads one
//     half and writes the other, one barrier per pass, a thread takes the butterflies t, t + T, ... one at a time (R points in
//     registers);
//   * the last pass has at most one butterfly per thread (the planner sees to it), keeps it across a barrier and leaves the
//     spectrum in the natural-order layout the epilogue reads (ex3).
// Pass s (radix R, Ns = product of the radices before it, nb = np / R butterflies): butterfly j reads x[j + k nb], k < R,
// multiplies by W_(Ns R)^(k (j mod Ns)) (a table per pass, [k][j mod Ns]: consecutive butterflies read consecutive entries), transforms, and writes
// y[(j / Ns) Ns R + (j mod Ns) + k Ns].  Natural order in, natural order out.
#pragma once
#include "wf_tick_phases.hpp"

namespace wf {

// ---- compile-time twiddles: cos / sin of 2 pi m / r as literals after unrolling -------------------------------------
constexpr double mr_pi = 3.14159265358979323846264338327950288;
constexpr double mr_sin_series(double x) // |x| <= pi / 4
{
    const double x2 = x * x;
    double term = x, sum = x;
    for(int i = 1; i < 12; ++i) {
        term *= -x2 / (double)((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr double mr_cos_series(double x) // |x| <= pi / 4
{
    const double x2 = x * x;
    double term = 1.0, sum = 1.0;
    for(int i = 1; i < 12; ++i) {
        term *= -x2 / (double)((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}
struct mr_cd { double c, s; };
// (cos, sin) of 2 pi m / r, reduced to the first octant by the symmetries of the circle (exact for the multiples of pi / 4)
constexpr mr_cd mr_cis(int m, int r)
{
    m %= r;
    if(m < 0)
        m += r;
    // angle = 2 pi m / r = (pi / 4) * (8 m / r): octant o and remainder
    const int num = 8 * m;        // angle in units of (pi / 4) / r ... o = floor(num / r)
    const int o = num / r;        // 0 .. 7
    const int rem = num - o * r;  // angle = (o + rem / r) pi / 4
    const double f = (double)rem / (double)r * (mr_pi / 4.0); // in [0, pi / 4)
    const double g = mr_pi / 4.0 - f;                          // in (0, pi / 4]
    double c = 0.0, s = 0.0;
    switch(o) {
    case 0: c = mr_cos_series(f); s = mr_sin_series(f); break;
    case 1: c = mr_sin_series(g); s = mr_cos_series(g); break;   // pi/4 + f = pi/2 - g
    case 2: c = -mr_sin_series(f); s = mr_cos_series(f); break;  // pi/2 + f
    case 3: c = -mr_cos_series(g); s = mr_sin_series(g); break;  // 3pi/4 + f = pi - g
    case 4: c = -mr_cos_series(f); s = -mr_sin_series(f); break; // pi + f
    case 5: c = -mr_sin_series(g); s = -mr_cos_series(g); break; // 5pi/4 + f = 3pi/2 - g
    case 6: c = mr_sin_series(f); s = -mr_cos_series(f); break;  // 3pi/2 + f
    default: c = mr_cos_series(g); s = -mr_sin_series(g); break; // 7pi/4 + f = 2pi - g
    }
    if(rem == 0) { // multiples of pi / 4: exact zeros and ones instead of 6e-17
        const int q = o & 7;
        const double h = 0.70710678118654752440084436210485;
        const double cc[8] = {1, h, 0, -h, -1, -h, 0, h}, ss[8] = {0, h, 1, h, 0, -h, -1, -h};
        c = cc[q];
        s = ss[q];
    }
    return mr_cd{c, s};
}
// W_R^m, m < R, as floats: a literal table per radix; the index is a compile-time constant after unrolling, so what reaches
// the instruction stream is the pair of immediates
template<int R> struct MrTwiddles {
    float c[R], s[R];
    constexpr MrTwiddles() : c{}, s{}
    {
        for(int m = 0; m < R; ++m) {
            const mr_cd w = mr_cis(m, R);
            c[m] = (float)w.c;
            s[m] = (float)w.s;
        }
    }
};
// d * W_R^m, W_R = exp(-2 pi i / R)
template<int R> WF_DEV cf mr_mul_w(cf d, int m)
{
    constexpr MrTwiddles<R> tab{};
    m %= R;
    if(m == 0)
        return d;
    if(4 * m == R)
        return cf{d.y, -d.x};
    if(2 * m == R)
        return cf{-d.x, -d.y};
    if(4 * m == 3 * R)
        return cf{-d.y, d.x};
    const float wr = tab.c[m], wi = -tab.s[m];
    return cf{fmaf(d.x, wr, -(d.y * wi)), fmaf(d.x, wi, d.y * wr)};
}

// ---- in-register DFTs, natural order in and out ------------------------------------------------------------------------
template<int R> struct MrDft;
template<> struct MrDft<2> {
    static WF_DEV void run(cf (&v)[2])
    {
        const cf a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};
template<> struct MrDft<3> {
    static WF_DEV void run(cf (&v)[3])
    {
        const float s = 0.86602540378443864676f; // sin(2 pi / 3)
        const cf t1 = cadd(v[1], v[2]), t2 = csub(v[1], v[2]);
        const cf m = cf{fmaf(-0.5f, t1.x, v[0].x), fmaf(-0.5f, t1.y, v[0].y)};
        const cf r = cf{s * t2.y, -s * t2.x}; // -i s (v1 - v2)
        v[0] = cadd(v[0], t1);
        v[1] = cadd(m, r);
        v[2] = csub(m, r);
    }
};
template<> struct MrDft<4> {
    static WF_DEV void run(cf (&v)[4])
    {
        const cf a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
        const cf jd = cf{d.y, -d.x}; // -i d
        v[0] = cadd(a, c);
        v[1] = cadd(b, jd);
        v[2] = csub(a, c);
        v[3] = csub(b, jd);
    }
};
template<> struct MrDft<5> {
    static WF_DEV void run(cf (&v)[5])
    {
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f; // cos(2 pi / 5), cos(4 pi / 5)
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;  // sin(2 pi / 5), sin(4 pi / 5)
        const cf a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]), a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
        const cf m1 = cf{fmaf(c2, a2.x, fmaf(c1, a1.x, v[0].x)), fmaf(c2, a2.y, fmaf(c1, a1.y, v[0].y))};
        const cf m2 = cf{fmaf(c1, a2.x, fmaf(c2, a1.x, v[0].x)), fmaf(c1, a2.y, fmaf(c2, a1.y, v[0].y))};
        const cf n1 = cf{fmaf(s2, b2.x, s1 * b1.x), fmaf(s2, b2.y, s1 * b1.y)};    // s1 b1 + s2 b2
        const cf n2 = cf{fmaf(-s1, b2.x, s2 * b1.x), fmaf(-s1, b2.y, s2 * b1.y)};  // s2 b1 - s1 b2
        const cf r1 = cf{n1.y, -n1.x}, r2 = cf{n2.y, -n2.x}; // -i n
        v[0] = cadd(v[0], cadd(a1, a2));
        v[1] = cadd(m1, r1);
        v[4] = csub(m1, r1);
        v[2] = cadd(m2, r2);
        v[3] = csub(m2, r2);
    }
};
// an odd prime radix by the definition, halved by symmetry: with a_j = x_j + x_(P-j), b_j = x_j - x_(P-j), j = 1 .. (P-1)/2,
//   X[k], X[P-k] = (x_0 + sum_j a_j cos(2 pi j k / P)) -+ i (sum_j b_j sin(2 pi j k / P));  (P-1)^2 real multiply-adds in all
template<int P> struct MrDftPrime {
    static WF_DEV void run(cf (&v)[P])
    {
        constexpr int H = (P - 1) / 2;
        constexpr MrTwiddles<P> tab{};
        cf a[H], b[H];
        WF_UNROLL
        for(int j = 0; j < H; ++j) {
            a[j] = cadd(v[j + 1], v[P - 1 - j]);
            b[j] = csub(v[j + 1], v[P - 1 - j]);
        }
        const cf x0 = v[0];
        cf s0 = x0;
        WF_UNROLL
        for(int j = 0; j < H; ++j)
            s0 = cadd(s0, a[j]);
        v[0] = s0;
        WF_UNROLL
        for(int k = 1; k <= H; ++k) {
            cf c = x0, d = cf{0.0f, 0.0f};
            WF_UNROLL
            for(int j = 0; j < H; ++j) {
                const int m = ((j + 1) * k) % P;
                c = cf{fmaf(tab.c[m], a[j].x, c.x), fmaf(tab.c[m], a[j].y, c.y)};
                d = cf{fmaf(tab.s[m], b[j].x, d.x), fmaf(tab.s[m], b[j].y, d.y)};
            }
            // -i d = (d.y, -d.x)
            v[k] = cf{c.x + d.y, c.y - d.x};
            v[P - k] = cf{c.x - d.y, c.y + d.x};
        }
    }
};
template<> struct MrDft<7> : MrDftPrime<7> {};
template<> struct MrDft<11> : MrDftPrime<11> {};
template<> struct MrDft<13> : MrDftPrime<13> {};
template<> struct MrDft<17> : MrDftPrime<17> {}; // (17, 19, 23: first pass only, like 20 and 25)
template<> struct MrDft<19> : MrDftPrime<19> {};
template<> struct MrDft<23> : MrDftPrime<23> {};
// R = A B by one Cooley-Tukey step in registers: n = B n1 + n2, k = k1 + A k2
template<int A, int B> struct MrDftCT {
    static WF_DEV void run(cf (&v)[A * B])
    {
        cf y[A * B];
        WF_UNROLL
        for(int n2 = 0; n2 < B; ++n2) {
            cf u[A];
            WF_UNROLL
            for(int n1 = 0; n1 < A; ++n1)
                u[n1] = v[B * n1 + n2];
            MrDft<A>::run(u);
            WF_UNROLL
            for(int k1 = 0; k1 < A; ++k1)
                y[k1 * B + n2] = mr_mul_w<A * B>(u[k1], k1 * n2);
        }
        WF_UNROLL
        for(int k1 = 0; k1 < A; ++k1) {
            cf u[B];
            WF_UNROLL
            for(int n2 = 0; n2 < B; ++n2)
                u[n2] = y[k1 * B + n2];
            MrDft<B>::run(u);
            WF_UNROLL
            for(int k2 = 0; k2 < B; ++k2)
                v[k1 + A * k2] = u[k2];
        }
    }
};
template<> struct MrDft<6> : MrDftCT<3, 2> {};
template<> struct MrDft<8> : MrDftCT<4, 2> {};
template<> struct MrDft<9> : MrDftCT<3, 3> {};
template<> struct MrDft<10> : MrDftCT<5, 2> {};
template<> struct MrDft<12> : MrDftCT<3, 4> {};
template<> struct MrDft<15> : MrDftCT<3, 5> {};
template<> struct MrDft<16> : MrDftCT<4, 4> {};
template<> struct MrDft<20> : MrDftCT<5, 4> {};
template<> struct MrDft<25> : MrDftCT<5, 5> {};

// one butterfly: inputs from `src` (linear), twiddles (TW: every pass but the first), DFT; X[k] left in v.
// The twiddled form holds 2 (R - 1) more registers while its table entries and its points are in flight: radices above 16
// are planned into the first pass only (plan_mixed_radix).
template<int R, bool TW> WF_DEV void mr_butterfly(const cf *src, const cf *tw, int j, int nb, int ns, cf (&v)[R])
{
    if constexpr(TW) {
        const int jm = j % ns;
        cf w[R];
        WF_UNROLL
        for(int k = 1; k < R; ++k) { // requested before the LDS reads so that the two latencies overlap
            const f2 q = ld2(reinterpret_cast<const float *>(tw + k * ns + jm));
            w[k] = cf{q.x, q.y};
        }
        WF_UNROLL
        for(int k = 0; k < R; ++k)
            v[k] = lds_ld2(src, j + k * nb);
        WF_UNROLL
        for(int k = 1; k < R; ++k)
            v[k] = cmul(v[k], w[k]);
    } else {
        WF_UNROLL
        for(int k = 0; k < R; ++k)
            v[k] = lds_ld2(src, j + k * nb);
    }
    MrDft<R>::run(v);
}
template<int R, bool TW> WF_DEV void mr_pass_r(const cf *src, cf *dst, const cf *tw, int np, int ns, int t, int T)
{
    const int nb = np / R;
    for(int j = t; j < nb; j += T) {
        cf v[R];
        mr_butterfly<R, TW>(src, tw, j, nb, ns, v);
        const int base = TW ? (j / ns) * ns * R + (j % ns) : j * R;
        WF_UNROLL
        for(int k = 0; k < R; ++k)
            lds_st2(dst, base + k * ns, v[k]);
    }
}
// A first pass whose radix is a prime p of 29 .. 127 (sizes such as 64 x 37: one position of the FFT-size slider in three has
// such a factor): the p-point DFTs by the definition, p^2 complex multiply-adds each -- up to p ~ 100 fewer operations than the two
// transforms of >= 2 np points Bluestein's algorithm takes, and no chirp tables.  Work item = (four consecutive butterflies, one
// output index k): per input index n one table entry W_p^(n k mod p) (LDS; k is the same across most of a wavefront) and JB
// consecutive points (16-byte LDS reads) feed JB accumulators.  np / p is a multiple of 4 (the planner sees to it).
template<int JB> WF_DEV void mr_pass_prime_jb(int p, const cf *src, cf *dst, const cf *wp, int np, int t, int T)
{
    const int nb = np / p, nq = nb / JB;
    for(int it = t; it < nq * p; it += T) {
        const int k = it / nq, j = JB * (it - k * nq);
        cf acc[JB];
        WF_UNROLL
        for(int i = 0; i < JB; ++i)
            acc[i] = cf{0.0f, 0.0f};
        int idx = 0;
        for(int n = 0; n < p; ++n) {
            const cf w = lds_ld2(wp, idx);
            WF_UNROLL
            for(int i = 0; i < JB; i += 2) {
                const f4 x = lds_ld4(src, j + n * nb + i);
                acc[i] = cf{fmaf(x.x, w.x, fmaf(-x.y, w.y, acc[i].x)), fmaf(x.x, w.y, fmaf(x.y, w.x, acc[i].y))};
                acc[i + 1] = cf{fmaf(x.z, w.x, fmaf(-x.w, w.y, acc[i + 1].x)), fmaf(x.z, w.y, fmaf(x.w, w.x, acc[i + 1].y))};
            }
            idx += k;
            idx = idx >= p ? idx - p : idx;
        }
        WF_UNROLL
        for(int i = 0; i < JB; ++i)
            lds_st2(dst, (j + i) * p + k, acc[i]);
    }
}
// (eight butterflies per work item where np / p allows: five LDS reads per 32 multiply-adds instead of three per 16)
WF_DEV void mr_pass_prime(int p, const cf *src, cf *dst, const cf *wp, int np, int t, int T)
{
    if(((np / p) & 7) == 0)
        mr_pass_prime_jb<8>(p, src, dst, wp, np, t, T);
    else
        mr_pass_prime_jb<4>(p, src, dst, wp, np, t, T);
}
// first pass (ns == 1: no twiddles): every planned radix
template<bool SMALL = false> WF_DEV void mr_pass_first(int R, const cf *src, cf *dst, int np, int t, int T)
{
    switch(R) {
    case 2: mr_pass_r<2, false>(src, dst, nullptr, np, 1, t, T); break;
    case 3: mr_pass_r<3, false>(src, dst, nullptr, np, 1, t, T); break;
    case 4: mr_pass_r<4, false>(src, dst, nullptr, np, 1, t, T); break;
    case 5: mr_pass_r<5, false>(src, dst, nullptr, np, 1, t, T); break;
    case 6: mr_pass_r<6, false>(src, dst, nullptr, np, 1, t, T); break;
    case 7: mr_pass_r<7, false>(src, dst, nullptr, np, 1, t, T); break;
    case 8: mr_pass_r<8, false>(src, dst, nullptr, np, 1, t, T); break;
    case 9: mr_pass_r<9, false>(src, dst, nullptr, np, 1, t, T); break;
    case 10: mr_pass_r<10, false>(src, dst, nullptr, np, 1, t, T); break;
    case 11: mr_pass_r<11, false>(src, dst, nullptr, np, 1, t, T); break;
    case 12: mr_pass_r<12, false>(src, dst, nullptr, np, 1, t, T); break;
    case 13: if constexpr(!SMALL) mr_pass_r<13, false>(src, dst, nullptr, np, 1, t, T); break;
    case 15: if constexpr(!SMALL) mr_pass_r<15, false>(src, dst, nullptr, np, 1, t, T); break;
    case 16: if constexpr(!SMALL) mr_pass_r<16, false>(src, dst, nullptr, np, 1, t, T); break;
    case 17: if constexpr(!SMALL) mr_pass_r<17, false>(src, dst, nullptr, np, 1, t, T); break;
    case 19: if constexpr(!SMALL) mr_pass_r<19, false>(src, dst, nullptr, np, 1, t, T); break;
    case 23: if constexpr(!SMALL) mr_pass_r<23, false>(src, dst, nullptr, np, 1, t, T); break;
    case 20: if constexpr(!SMALL) mr_pass_r<20, false>(src, dst, nullptr, np, 1, t, T); break;
    default: if constexpr(!SMALL) mr_pass_r<25, false>(src, dst, nullptr, np, 1, t, T); break;
    }
}
// middle passes: radices up to 16
template<bool SMALL = false> WF_DEV void mr_pass(int R, const cf *src, cf *dst, const cf *tw, int np, int ns, int t, int T)
{
    switch(R) {
    case 2: mr_pass_r<2, true>(src, dst, tw, np, ns, t, T); break;
    case 3: mr_pass_r<3, true>(src, dst, tw, np, ns, t, T); break;
    case 4: mr_pass_r<4, true>(src, dst, tw, np, ns, t, T); break;
    case 5: mr_pass_r<5, true>(src, dst, tw, np, ns, t, T); break;
    case 6: mr_pass_r<6, true>(src, dst, tw, np, ns, t, T); break;
    case 7: mr_pass_r<7, true>(src, dst, tw, np, ns, t, T); break;
    case 8: mr_pass_r<8, true>(src, dst, tw, np, ns, t, T); break;
    case 9: mr_pass_r<9, true>(src, dst, tw, np, ns, t, T); break;
    case 10: mr_pass_r<10, true>(src, dst, tw, np, ns, t, T); break;
    case 11: mr_pass_r<11, true>(src, dst, tw, np, ns, t, T); break;
    case 12: mr_pass_r<12, true>(src, dst, tw, np, ns, t, T); break;
    case 13: if constexpr(!SMALL) mr_pass_r<13, true>(src, dst, tw, np, ns, t, T); break;
    case 15: if constexpr(!SMALL) mr_pass_r<15, true>(src, dst, tw, np, ns, t, T); break;
    default: if constexpr(!SMALL) mr_pass_r<16, true>(src, dst, tw, np, ns, t, T); break;
    }
}
// the last pass: ns R == np, at most one butterfly per thread (j = t < nb = ns); X[j + k ns] handed to `store` behind `sync`
// (every thread has read its inputs: a store into the exchange buffer may overlap both halves)
template<int R, class Sync, class Store> WF_DEV void mr_last_r(bool process, const cf *src, const cf *tw, int ns, int t, Sync sync, Store store)
{
    cf v[R];
    const bool mine = process && t < ns;
    if(mine)
        mr_butterfly<R, true>(src, tw, t, ns, ns, v);
    sync();
    if(mine) {
        WF_UNROLL
        for(int k = 0; k < R; ++k)
            store(t + k * ns, v[k]);
    }
}
template<bool SMALL = false, class Sync, class Store> WF_DEV void mr_last(int R, bool process, const cf *src, const cf *tw, int ns, int t, Sync sync, Store store)
{
    switch(R) {
    case 2: mr_last_r<2>(process, src, tw, ns, t, sync, store); break;
    case 3: mr_last_r<3>(process, src, tw, ns, t, sync, store); break;
    case 4: mr_last_r<4>(process, src, tw, ns, t, sync, store); break;
    case 5: mr_last_r<5>(process, src, tw, ns, t, sync, store); break;
    case 6: mr_last_r<6>(process, src, tw, ns, t, sync, store); break;
    case 7: mr_last_r<7>(process, src, tw, ns, t, sync, store); break;
    case 8: mr_last_r<8>(process, src, tw, ns, t, sync, store); break;
    case 9: mr_last_r<9>(process, src, tw, ns, t, sync, store); break;
    case 10: mr_last_r<10>(process, src, tw, ns, t, sync, store); break;
    case 11: mr_last_r<11>(process, src, tw, ns, t, sync, store); break;
    case 12: mr_last_r<12>(process, src, tw, ns, t, sync, store); break;
    case 13: if constexpr(!SMALL) mr_last_r<13>(process, src, tw, ns, t, sync, store); else sync(); break;
    case 15: if constexpr(!SMALL) mr_last_r<15>(process, src, tw, ns, t, sync, store); else sync(); break;
    default: if constexpr(!SMALL) mr_last_r<16>(process, src, tw, ns, t, sync, store); else sync(); break;
    }
}

// The window of this spectrum as np complex points z_j = (win_2j x_2j, win_2j+1 x_2j+1), natural order, into lds[0 .. np).
// Returns whether any of this thread's samples is non-zero (the reference's silence scan, :63-72).
template<class G> WF_DEV bool mr_fetch(const TickArgs &a, int t, const float *x, uint32_t start, cf *lds)
{
    constexpr int T = G::T, P = G::P;
    const uint32_t np = a.blu_n >> 1; // <= M / 2 = T P / 2
    uint32_t acc = 0;
    if((start & 3u) == 0u) {
        // (uniform per spectrum) the window starts on a 16-byte boundary of the ring: two points per request -- samples and window
        // coefficients as 16-byte vectors, a group never straddles the ring's wrap.  np is a multiple of 8.
        constexpr int IT = P / 4; // 2 T IT = M / 2 >= np
        f4 v[IT], w[IT];
        WF_UNROLL
        for(int i = 0; i < IT; ++i) {
            const uint32_t idx = 2u * (uint32_t)(t + T * i);
            const uint32_t at = idx < np ? idx : 0u;
            v[i] = ld4(x + ((start + 2u * at) & a.ring_mask));
            w[i] = ld4(a.window + 2u * at);
        }
        WF_UNROLL
        for(int i = 0; i < IT; ++i) {
            const uint32_t idx = 2u * (uint32_t)(t + T * i);
            if(idx < np) {
                acc |= f32_bits(v[i].x) | f32_bits(v[i].y) | f32_bits(v[i].z) | f32_bits(v[i].w);
                lds_st4(lds, (int)idx, cf{v[i].x * w[i].x, v[i].y * w[i].y}, cf{v[i].z * w[i].z, v[i].w * w[i].w});
            }
        }
        return (acc & 0x7fffffffu) != 0;
    }
    f2 v[P / 2], w[P / 2];
    WF_UNROLL
    for(int i = 0; i < P / 2; ++i) {
        const uint32_t idx = (uint32_t)(t + T * i);
        const bool in = idx < np;
        const uint32_t s = start + 2u * (in ? idx : 0u);
        v[i] = f2{x[s & a.ring_mask], x[(s + 1u) & a.ring_mask]};
        w[i] = ld2(a.window + 2u * (in ? idx : 0u));
    }
    WF_UNROLL
    for(int i = 0; i < P / 2; ++i) {
        const uint32_t idx = (uint32_t)(t + T * i);
        if(idx < np) {
            acc |= f32_bits(v[i].x) | f32_bits(v[i].y);
            lds_st2(lds, (int)idx, cf{v[i].x * w[i].x, v[i].y * w[i].y});
        }
    }
    return (acc & 0x7fffffffu) != 0;
}

// The whole transform.  On entry the np windowed points sit in lds[0 .. np) (natural order); Z[k] is handed to store(k, Z[k]) by
// the thread that finishes it.  Called by ALL threads of the spectrum (sync is its barrier).
// SMALL: the instantiation that carries the radices 2 ... 12 only (wf::mr_small_radices: the host picks it for
// plans made of those) -- the register-hungry in-register DFTs compile to nothing
template<class G, bool SMALL = false, class Sync, class Store> WF_DEV void mr_transform_to(const MrPlan &p, bool process, int np, int t, cf *lds, const cf *wp_lds, Sync sync, Store store)
{
    const int H = p.half; // second half of the exchange buffer (MrPlan::half >= np)
    sync(); // the fetch has written the first half
    if(process) {
        if(!SMALL && p.radix[0] > 25) { // (uniform) a prime of 29 .. 127: by the definition, its twiddles in LDS at wp_lds
            if constexpr(!SMALL)
                mr_pass_prime(p.radix[0], lds, lds + H, wp_lds, np, t, G::T);
        } else
            mr_pass_first<SMALL>(p.radix[0], lds, lds + H, np, t, G::T);
    }
    int ns = p.radix[0], cur = 1;
    for(int s = 1; s + 1 < p.passes; ++s) {
        const int R = p.radix[s];
        sync();
        if(process)
            mr_pass<SMALL>(R, lds + cur * H, lds + (1 - cur) * H, p.tw + p.tw_off[s], np, ns, t, G::T);
        cur ^= 1;
        ns *= R;
    }
    sync();
    mr_last<SMALL>(p.radix[p.passes - 1], process, lds + cur * H, p.tw + p.tw_off[p.passes - 1], ns, t, sync, store);
}
// ... with Z[k] left at mr_z_addr(p, k) of the exchange buffer, visible to every thread of the spectrum on return
template<class G, bool SMALL = false, class Sync> WF_DEV void mr_transform(const MrPlan &p, bool process, int np, int t, cf *lds, const cf *wp_lds, Sync sync)
{
    mr_transform_to<G, SMALL>(p, process, np, t, lds, wp_lds, sync, [lds, &p](int k, cf v) { lds_st2(lds, mr_z_addr(p, k), v); });
    sync();
}

// A plan known at compile time (three passes R0 x R1 x R2): the same pass functions with every length, stride and trip count a
// constant -- the divisions by the stride become multiplications, the loops over a thread's butterflies unroll, no dispatch on the
// radix.  For the sizes the plugin picks by itself (sample_rate / fps & -16).
// One pass of a compile-time plan IN PLACE (one wavefront per spectrum): every butterfly of this thread is read first (NB / 64
// rounds of R points in registers), then all of them are transformed and written back into the SAME buffer at their Stockham
// positions.  LDS operations of a wavefront execute in order and nobody else touches the spectrum's buffer, so the reads of every
// lane are in front of the writes of every lane without a barrier.  Half the exchange buffer of the two-halves form: 3.3 instead of
// 6.4 KB per spectrum at N = 800 -- more spectra per CU, which is what these sizes scale with (profiles/r04g_n800_phases.txt).
template<int T, int R, bool TW, int NB, int NS> WF_DEV void mr_pass_inplace(bool process, cf *buf, const cf *tw, int t)
{
    constexpr int ROUNDS = (NB + T - 1) / T;
    cf v[ROUNDS][R];
    if(process) {
        WF_UNROLL
        for(int r = 0; r < ROUNDS; ++r) {
            const int j = t + r * T;
            if(ROUNDS * T == NB || j < NB)
                mr_butterfly<R, TW>(buf, tw, j, NB, NS, v[r]);
        }
    }
    wave_fence();
    if(process) {
        WF_UNROLL
        for(int r = 0; r < ROUNDS; ++r) {
            const int j = t + r * T;
            if(ROUNDS * T == NB || j < NB) {
                const int base = TW ? (j / NS) * NS * R + (j % NS) : j * R;
                WF_UNROLL
                for(int k = 0; k < R; ++k)
                    lds_st2(buf, base + k * NS, v[r][k]);
            }
        }
    }
    wave_fence();
}
template<class G, int R0, int R1, int R2> WF_DEV void mr_transform_fixed_inplace(const MrPlan &p, bool process, int t, cf *lds)
{
    static_assert(G::T == 64, "in place: one wavefront per spectrum");
    constexpr int np = R0 * R1 * R2;
    wave_fence(); // the fetch has written the points
    mr_pass_inplace<G::T, R0, false, np / R0, 1>(process, lds, nullptr, t);
    mr_pass_inplace<G::T, R1, true, np / R1, R0>(process, lds, p.tw, t);
    // the last pass: one butterfly per thread at most, Z[k] into the four-plane layout (mr_z_addr) of the same buffer
    static_assert(R0 * R1 <= G::T, "the last pass of a compile-time plan has one butterfly per thread at most");
    cf v[R2];
    const bool mine = process && t < R0 * R1;
    if(mine)
        mr_butterfly<R2, true>(lds, p.tw + R1 * R0, t, R0 * R1, R0 * R1, v);
    wave_fence();
    if(mine) {
        WF_UNROLL
        for(int k = 0; k < R2; ++k)
            lds_st2(lds, mr_z_addr(p, t + k * R0 * R1), v[k]);
    }
    wave_fence();
}
template<class G, int R0, int R1, int R2, class Sync> WF_DEV void mr_transform_fixed(const MrPlan &p, bool process, int t, cf *lds, Sync sync)
{
    constexpr int np = R0 * R1 * R2;
    if constexpr(G::T == 64 && R0 * R1 <= G::T) { // (one wavefront per spectrum: in place -- the host sizes the buffer accordingly, setup_launch_blu)
        (void)sync;
        mr_transform_fixed_inplace<G, R0, R1, R2>(p, process, t, lds);
        return;
    }
    constexpr int H = (np + 15) & ~15; // wf::mr_exchange_half
    sync();
    if(process)
        mr_pass_r<R0, false>(lds, lds + H, nullptr, np, 1, t, G::T);
    sync();
    if(process)
        mr_pass_r<R1, true>(lds + H, lds, p.tw, np, R0, t, G::T);
    sync();
    mr_last_r<R2>(process, lds, p.tw + R1 * R0, R0 * R1, t, sync, [lds, &p](int k, cf v) { lds_st2(lds, mr_z_addr(p, k), v); });
    sync();
}

} // namespace wf
