// mr_check.cpp -- host check of the mixed-radix path (waveform_amd/csrc/wf_mixed.hpp + plan_mixed_radix), built and run by
// tests/test_cpu_units.py::test_mixed_radix_plans_and_transforms.  Test infrastructure, not part of the product.
//  * every in-register DFT against the definition in double;
//  * for EVERY multiple of 16 in [128, 16384] that is not a power of two: a plan exists for every size with no prime factor above
//    13 and for none with two large prime factors; one prime factor of 17 .. 127 on top is the planner's call (it weighs a prime
//    first pass against Bluestein's transforms); every plan multiplies to n / 2, uses the radices the kernel instantiates, keeps
//    radices above 16 in the first pass and leaves the last pass one butterfly per thread of the container geometry;
//  * for a spread of sizes: the passes themselves (mr_pass_first / mr_pass / the last pass's butterflies), lane by lane,
//    against a double DFT.
#include <cstdio>
#include <cmath>
#include <complex>
#include <vector>
#include "wf_geometry.hpp"
#include "wf_mixed.hpp"
#include "wf_host_tables.hpp"
using namespace wf;
static int failures = 0;
#define CHECK(c, ...) do { if(!(c)) { ++failures; std::printf("FAIL " __VA_ARGS__); std::printf("\n"); } } while(0)

template<int R> static void check_dft()
{
    cf v[R];
    std::complex<double> x[R];
    for(int i = 0; i < R; ++i) {
        x[i] = {std::sin(1.0 + i * 0.7), std::cos(0.3 * i * i)};
        v[i] = cf{(float)x[i].real(), (float)x[i].imag()};
    }
    MrDft<R>::run(v);
    double err = 0;
    for(int k = 0; k < R; ++k) {
        std::complex<double> s = 0;
        for(int n = 0; n < R; ++n)
            s += x[n] * std::polar(1.0, -2 * M_PI * k * n / R);
        err = std::max(err, std::abs(s - std::complex<double>(v[k].x, v[k].y)));
    }
    CHECK(err < 2e-6, "radix %d: max error %g", R, err);
}

static unsigned container_threads(unsigned n) // threads of the geometry the Bluestein / mixed-radix instantiation runs in (M >= n - 1)
{
    unsigned L = 512;
    while(L < n - 1)
        L <<= 1;
    switch(2 * L) { case 1024: return 64; case 2048: return 64; case 4096: return 128; case 8192: return 256; case 16384: return 512; default: return 1024; } // (GBig: the power-of-two kernel of that container runs on 512 threads of 32 points)
}

int main()
{
    check_dft<2>(); check_dft<3>(); check_dft<4>(); check_dft<5>(); check_dft<6>(); check_dft<8>(); check_dft<9>(); check_dft<10>();
    check_dft<12>(); check_dft<15>(); check_dft<16>(); check_dft<20>(); check_dft<25>(); check_dft<7>(); check_dft<11>(); check_dft<13>(); check_dft<17>(); check_dft<19>(); check_dft<23>();
    int planned = 0, small_plans = 0;
    for(unsigned n = 128; n <= 16384; n += 16) {
        unsigned r = n;
        for(unsigned p : {2u, 3u, 5u, 7u, 11u, 13u})
            while(r % p == 0)
                r /= p;
        const bool pow2 = (n & (n - 1)) == 0;
        // one factor 17 / 19 / 23 (a radix of the first pass only, like 20 and 25) or one prime of 29 .. 127 (the first pass by the
        // definition) may come on top; whether the rest then still orders into a plan is the planner's business: such sizes are
        // checked for consistency only
        bool one_big = false;
        for(unsigned p = 17; p <= 127; ++p)
            if(r == p)
                one_big = true;
        const bool smooth = r == 1;
        if(pow2)
            continue;
        int radix[4];
        const unsigned np = n / 2, T = container_threads(n);
        unsigned Lb = 512; // what Bluestein would transform instead: the planner weighs a prime first pass against it
        while(Lb < n - 1)
            Lb <<= 1;
        const int passes = plan_mixed_radix(np, T, radix, Lb);
        if(!one_big)
            CHECK((passes > 0) == smooth, "n = %u: plan %d passes, smooth %d", n, passes, (int)smooth);
        if(passes <= 0)
            continue;
        ++planned;
        unsigned long long prod = 1;
        for(int i = 0; i < passes; ++i) {
            const int v = radix[i];
            prod *= (unsigned long long)v;
            bool lead_prime = i == 0 && v >= 29 && v <= 127;
            for(int d = 2; lead_prime && d * d <= v; ++d)
                lead_prime = v % d != 0;
            const bool known = lead_prime || ((v == 17 || v == 19 || v == 23) && i == 0) || v == 7 || v == 11 || v == 13 || v == 2 || v == 3 || v == 4 || v == 5 || v == 6 || v == 8 || v == 9 || v == 10 || v == 12 || v == 15 || v == 16 || v == 20 || v == 25;
            CHECK(known && (v <= 16 || i == 0) && !(radix[0] > 25 && i > 0 && v > 16), "n = %u: radix %d in pass %d", n, v, i);
        }
        CHECK(passes >= 2 && passes <= 4 && prod == np, "n = %u: %d passes, product %llu", n, passes, prod);
        CHECK(np / (unsigned)radix[passes - 1] <= T, "n = %u: last pass has %u butterflies for %u threads", n, np / radix[passes - 1], T);
        // the exchange buffer sized by the transform (MrPlan::half / s3 / lds_cf as setup_launch_blu fills them): both halves hold the
        // points, the four-plane layout of the finished Z fits, never beyond the container geometry's buffer (M + padding >= 2 np)
        unsigned M = 512;
        while(M < 2 * np)
            M <<= 1;
        const unsigned container = M + M / 16 + 64; // (a lower bound of every Geom<2 M, ..>::LDS_CF: EX1 / EX3 padding)
        const unsigned half = mr_exchange_half(np), cfb = mr_exchange_cf(np, container), s3 = half / 4 + 4;
        CHECK(half >= np && half % 16 == 0 && half <= M / 2 + 15, "n = %u: half %u", n, half);
        CHECK(cfb >= 2 * half || cfb == container, "n = %u: buffer %u for two halves of %u", n, cfb, half);
        CHECK(3 * s3 + (np - 1) / 4 < cfb, "n = %u: Z's last plane ends at %u of %u", n, 3 * s3 + (np - 1) / 4, cfb);
        CHECK(cfb >= 576 || cfb == container, "n = %u: the display phase's floor (576), got %u", n, cfb);
        // the small-radix instantiation is picked for plans made of its radices only
        bool all_small = true;
        for(int i = 0; i < passes; ++i)
            all_small = all_small && (radix[i] == 2 || radix[i] == 3 || radix[i] == 4 || radix[i] == 5 || radix[i] == 6 || radix[i] == 7 || radix[i] == 8 || radix[i] == 9 || radix[i] == 10 || radix[i] == 11 || radix[i] == 12);
        CHECK(mr_small_radices(radix, passes) == all_small, "n = %u: mr_small_radices", n);
        small_plans += all_small && T <= 256;
    }
    CHECK(small_plans >= 30, "only %d sizes on the small-radix instantiation", small_plans); // (34: the automatic sizes 640 ... 2400 among them)
    std::printf("small-radix plans on one to four wavefronts: %d\n", small_plans);
    std::printf("planned %d sizes\n", planned);
    for(unsigned n : {800u, 1600u, 960u, 1920u, 2000u, 320u, 144u, 8000u, 1536u, 15552u, 12288u, 6000u, 4160u, 1760u, 1456u, 880u, 352u, 16016u, 224u, 1824u, 304u, 1088u, 1472u, 5888u, 14720u, 4144u, 2368u, 1856u, 8128u, 16256u, 592u, 7808u}) {
        int radix[4], off[4];
        const unsigned np = n / 2, T = container_threads(n);
        const int passes = plan_mixed_radix(np, T, radix);
        std::vector<cfloat> twf, w;
        build_mixed_radix_tables(n, passes, radix, twf, off, w);
        std::vector<cf> tw(twf.size());
        for(size_t i = 0; i < tw.size(); ++i)
            tw[i] = cf{twf[i].re, twf[i].im};
        unsigned M = 1;
        while(M < 2 * np)
            M <<= 1;
        const unsigned H = M / 2;
        std::vector<cf> lds(2 * M + 64);
        std::vector<std::complex<double>> x(np);
        for(unsigned i = 0; i < np; ++i) {
            x[i] = {std::sin(0.1 * i) + 0.3 * std::cos(1.7 * i), std::cos(0.37 * i)};
            lds[i] = cf{(float)x[i].real(), (float)x[i].imag()};
        }
        int ns = radix[0], cur = 1;
        std::vector<cfloat> wpf;
        build_prime_twiddles(radix[0] > 25 ? radix[0] : 2, 128, wpf);
        std::vector<cf> wp(wpf.size());
        for(size_t i = 0; i < wp.size(); ++i)
            wp[i] = cf{wpf[i].re, wpf[i].im};
        for(unsigned t = 0; t < T; ++t) {
            if(radix[0] > 25)
                mr_pass_prime(radix[0], lds.data(), lds.data() + H, wp.data(), (int)np, (int)t, (int)T);
            else
                mr_pass_first(radix[0], lds.data(), lds.data() + H, (int)np, (int)t, (int)T);
        }
        for(int s = 1; s + 1 < passes; ++s) {
            for(unsigned t = 0; t < T; ++t)
                mr_pass(radix[s], lds.data() + cur * H, lds.data() + (1 - cur) * H, tw.data() + off[s], (int)np, ns, (int)t, (int)T);
            cur ^= 1;
            ns *= radix[s];
        }
        // the last pass through the middle-pass code (same butterflies, linear output): one butterfly per thread
        std::vector<cf> out(np);
        for(unsigned t = 0; t < T; ++t)
            mr_pass(radix[passes - 1], lds.data() + cur * H, out.data(), tw.data() + off[passes - 1], (int)np, ns, (int)t, (int)T);
        double err = 0, mx = 0;
        for(unsigned k = 0; k < np; k += 3) {
            std::complex<double> s = 0;
            for(unsigned m = 0; m < np; ++m)
                s += x[m] * std::polar(1.0, -2 * M_PI * (double)((unsigned long long)k * m % np) / np);
            err = std::max(err, std::abs(s - std::complex<double>(out[k].x, out[k].y)));
            mx = std::max(mx, std::abs(s));
        }
        std::printf("n = %u: %d passes %dx%dx%dx%d, max error %.3g of %.3g\n", n, passes, radix[0], radix[1], radix[2], radix[3], err, mx);
        CHECK(err <= 2e-6 * mx, "n = %u: transform error %g of %g", n, err, mx);
    }
    std::printf(failures ? "FAILED %d\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
