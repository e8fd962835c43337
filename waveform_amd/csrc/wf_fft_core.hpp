// wf_fft_core.hpp -- the per-spectrum math of the fused tick kernel, written as
// per-thread phase functions over private register arrays and one LDS exchange
// buffer.  Replaces FFTW's r2c plan + the reference's post-FFT loop
// (reference: src/source_generic.cpp:97-135, fftwf_execute at :105-106, plan at
// src/source.cpp:1187; semantics: SURVEY.md Appendix A.2).
//
// Decomposition (one spectrum = N real samples -> M = N/2 complex bins):
//   z[n] = x[2n] + i x[2n+1]                     (pack: two samples = one complex)
//   Z    = DFT_M(z) as three in-register passes of radix R1, R2, R3 (M = R1*R2*R3),
//          T = M/P threads per spectrum, P points per thread, data exchanged
//          through LDS between passes (T = 64: one wavefront owns the spectrum)
//   X[k] = real-split of Z (needs Z[k] and Z[M-k]), k = 0..M-1; Nyquist bin dropped
//          exactly as the reference does (src/source_avx2.cpp:29).
// Index algebra (n = n1*R2*R3 + n2*R3 + n3, k = k1 + R1*k2 + R1*R2*k3):
//   pass 1: A[k1][n'] = sum_n1 z[n1*(M/R1) + n'] W_R1^(n1 k1),   n' = n2*R3 + n3
//           A'        = A * W_M^(n' k1)                         (table tw1[k1][n'])
//   pass 2: B[k1][k2][n3] = sum_n2 A'[k1][n2*R3+n3] W_R2^(n2 k2)
//           B'            = B * W_(R2R3)^(n3 k2)                (table tw2[k2][n3])
//   pass 3: Z[k1 + R1 k2 + R1R2 k3] = sum_n3 B'[k1][k2][n3] W_R3^(n3 k3)
//
// The same source is compiled (a) by hipcc into the gfx950 kernel and (b) by g++
// into tests/emu, a lane-by-lane wavefront emulator used on the GPU-less build
// box to check the index algebra and count LDS bank conflicts.  (b) is a test
// harness only -- the product library contains no CPU path.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WF_DEV __device__ __forceinline__
#define WF_UNROLL _Pragma("unroll")
#else
#include <cmath>
#define WF_DEV inline
#define WF_UNROLL
#endif

namespace wf {

struct cf { float x, y; };
struct alignas(16) f4 { float x, y, z, w; };
struct alignas(8) f2 { float x, y; };

WF_DEV cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
WF_DEV cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
WF_DEV cf cmul(cf a, cf w) { return {fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.x, w.y, a.y * w.x)}; }

// cos(2 pi q / 32), q = 0..8 (double-rounded-to-float literals)
WF_DEV float cos32(int q)
{
    switch(q) {
    case 0: return 1.0f;
    case 1: return 0.98078528040323043f;
    case 2: return 0.92387953251128674f;
    case 3: return 0.83146961230254524f;
    case 4: return 0.70710678118654752f;
    case 5: return 0.55557023301960222f;
    case 6: return 0.38268343236508977f;
    case 7: return 0.19509032201612827f;
    default: return 0.0f;
    }
}

// d * W_32^q with W_32 = exp(-2 pi i / 32), q in [0, 16); q is a compile-time constant
// after unrolling so every branch folds away.
WF_DEV cf mul_w32(cf d, int q)
{
    if(q == 0) return d;
    if(q == 8) return {d.y, -d.x};
    if(q == 4) { const float c = 0.70710678118654752f; return {(d.x + d.y) * c, (d.y - d.x) * c}; }
    if(q == 12) { const float c = 0.70710678118654752f; return {(d.y - d.x) * c, -(d.x + d.y) * c}; }
    // general: W = (cos t, -sin t), t = 2 pi q / 32; sin t = cos(2 pi (8 - q)/32)
    float wr, wi;
    if(q < 8) { wr = cos32(q); wi = -cos32(8 - q); }
    else { wr = -cos32(16 - q); wi = -cos32(q - 8); }
    return {fmaf(d.x, wr, -(d.y * wi)), fmaf(d.x, wi, d.y * wr)};
}

// (hf ? W_32^j : 1) for hf in {0.0f, 1.0f}, j in [0, 16) a compile-time constant after unrolling
WF_DEV cf half_twiddle32(int j, float hf)
{
    float wr, wi; // W_32^j = (cos t, -sin t), t = 2 pi j / 32
    if(j <= 8) { wr = cos32(j); wi = -cos32(8 - j); }
    else { wr = -cos32(16 - j); wi = -cos32(j - 8); }
    return {fmaf(hf, wr, 1.0f - hf), hf * wi}; // exact for both values of hf
}

constexpr int ilog2(int v) { return (v <= 1) ? 0 : 1 + ilog2(v >> 1); }
constexpr int brev(int v, int bits) { return (bits == 0) ? 0 : ((v & 1) << (bits - 1)) | brev(v >> 1, bits - 1); }

// In-register radix-R DFT (R = 2..32, power of two), decimation in frequency:
// on return X[k] sits in v[brev(k)] -- callers index with brev<R>, which costs nothing
// because every index is a compile-time constant after unrolling.
template<int R> WF_DEV void dft_dif(cf (&v)[R])
{
    WF_UNROLL
    for(int h = R / 2; h >= 1; h >>= 1) {
        WF_UNROLL
        for(int s = 0; s < R; s += 2 * h) {
            WF_UNROLL
            for(int j = 0; j < h; ++j) {
                const cf a = v[s + j];
                const cf b = v[s + j + h];
                v[s + j] = cadd(a, b);
                v[s + j + h] = mul_w32(csub(a, b), j * (16 / h));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// FFT geometry for one spectrum
// ---------------------------------------------------------------------------------------
template<int N_, int T_, int R1_, int R2_, int R3_> struct Geom {
    static constexpr int N = N_;            // real samples
    static constexpr int M = N_ / 2;        // complex points / output bins
    static constexpr int T = T_;            // threads per spectrum
    static constexpr int P = M / T_;        // points per thread
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr int B1 = P / R1_;      // pass-1 butterflies per thread (1 or 2)
    static constexpr int H2 = (R2_ > P) ? R2_ / P : 1; // threads sharing one pass-2 butterfly (N = 32768 only)
    static constexpr int B2 = (R2_ > P) ? 1 : P / R2_;
    static constexpr int H3 = (R3_ > P) ? R3_ / P : 1; // threads sharing one pass-3 butterfly (radix 32 with 16 points per thread)
    static constexpr int B3 = (R3_ > P) ? 1 : P / R3_;
    static constexpr int M1 = M / R1_;      // = R2*R3 = T*B1
    // LDS exchange layouts (complex units).  ex1: [k1][n'] with padded row stride S1;
    // ex2: [q' = k1 + R1*k2][n3] rows of R3 with an XOR swizzle on 16-byte chunks;
    // ex3: four planes by k mod 4 -- Z[k] at (k & 3) * S3 + (k >> 2) -- so that the real split, whose threads own four
    // consecutive bins and need Z[k] and Z[M-k], reads consecutive 8-byte words across a wavefront in both directions.
    // ex1 row stride: pass 2 reads, per half wavefront, 32/R3 rows of R3 consecutive points (8 bytes each); the rows must
    // start 2*R3 banks apart, i.e. S1 % 32 == R3 % 32 (any padding does for R3 = 32: one row fills all 64 banks)
    static constexpr int S1 = M1 + ((R3_ < 32) ? R3_ : 8);
    static constexpr int EX1_SIZE = R1_ * S1;
    static constexpr int EX2_SIZE = M;
    static constexpr int S3 = M / 4 + 4;    // ex3 plane stride (see ex3_addr)
    static constexpr int EX3_SIZE = 4 * S3;
    // per spectrum, in cf; never less than the 1024-point geometry's 576: the bars / curve phase parks the dB row and stages
    // the filter's inputs in this buffer, and the smallest geometry would otherwise hold fewer outputs per row (800-point
    // filtered curves) than the zero-padded runs on the 1024-point geometry it replaced did
    static constexpr int LDS_NEED = (EX1_SIZE > EX3_SIZE) ? EX1_SIZE : EX3_SIZE;
    static constexpr int LDS_CF = LDS_NEED > 576 ? LDS_NEED : 576;
    static_assert(R1_ * R2_ * R3_ == N_ / 2, "radices must multiply to M");
    static_assert(B1 == 1 || B1 == 2, "pass 1 loads 8 or 16 bytes per thread");
    static_assert(T_ * B1 == M1, "pass-1 butterflies must tile the threads");
    static_assert(B2 >= 1 && (R3_ % B2) == 0, "pass-2 butterflies of a thread share k1");
    static_assert(H2 == 1 || (H2 == 2 && T_ == R1_ * R3_ * 2), "a pass-2 butterfly is split over at most two threads");
    static_assert(H3 == 1 || H3 == 2, "a pass-3 butterfly is split over at most two threads");
    static_assert(B3 >= 1 && T_ * B3 == R1_ * R2_ * H3, "pass-3 butterflies must tile the threads");
    static_assert((P % 4) == 0, "epilogue handles 4 bins per step");
};

template<class G> WF_DEV int ex1_addr(int k1, int np) { return k1 * G::S1 + np; }
// ex2 row q' = k1 + R1*k2 holds n3 = 0..R3-1; rows are R3 complex = R3/2 16-byte chunks.
// chunk c of row q' is stored at chunk c ^ swz(q') to spread the 16-lane b128 groups over banks.
template<class G> WF_DEV int ex2_swz(int q)
{
    constexpr int CH = G::R3 / 2;                  // chunks per row
    constexpr int ROWS_PER_BANKROW = (32 / G::R3) > 0 ? (32 / G::R3) : 1; // rows per 256-byte bank row
    if(CH < 2) return 0;
    return (q / ROWS_PER_BANKROW) & (CH - 1);
}
template<class G> WF_DEV int ex2_addr(int q, int n3)
{
    const int chunk = (n3 >> 1) ^ ex2_swz<G>(q);
    return q * G::R3 + chunk * 2 + (n3 & 1);
}
template<class G> WF_DEV int ex3_addr(int k) { return (k & 3) * G::S3 + (k >> 2); }

// Address algebra the compiler does not find by itself (every helper below is checked against the plain formulas by
// tests/emu, which replays the kernel's phase functions lane by lane):
//  * ex3 is affine in steps that are multiples of 4 bins:  ex3_addr(k + 4*m) == ex3_addr(k) + m
//  * ex2: for q = k1 + R1*k2 the swizzle term of a thread's store is its k2 = 0 address XOR a compile-time constant:
//        ex2_addr(k1 + R1*k2, n3) == (ex2_addr(k1, n3) ^ ex2_xor<G>(k2)) + k2*R1*R3
//    (R1 is a multiple of the rows per bank row, so swz(q) = (k1/RPB + (R1/RPB)*k2) mod CH, and k1/RPB < R1/RPB has no
//    bit in common with (R1/RPB)*k2.)
template<class G> constexpr int ex3_step(int bins) { return bins / 4; } // bins % 4 == 0
template<class G> constexpr int ex2_xor(int k2)
{
    constexpr int CH = G::R3 / 2;
    constexpr int RPB = (32 / G::R3) > 0 ? (32 / G::R3) : 1;
    static_assert(G::R1 % RPB == 0, "ex2 swizzle algebra needs R1 to be a multiple of the rows per bank row");
    if(CH < 2) return 0;
    return (((G::R1 / RPB) * k2) & (CH - 1)) * 2;
}

} // namespace wf
