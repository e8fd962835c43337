// wf_geometry.hpp -- the FFT decompositions the library ships, one per supported FFT size.
// T threads per spectrum (1, 1, 2, 4, 8, 8 wavefronts for N = 1024 ... 32768); every thread owns
// P = N/(2T) complex points (4, 8 or 16; 32 at N = 32768, where a workgroup has a CU to itself anyway); pass 1 fetches 16-byte vectors where the radices allow (8-byte ones at 512 and 32768).  Measured alternatives with 32 points per
// thread (N = 4096 on one wavefront, 8192 on two, 16384 on four) held 156-168 VGPRs and ran 15-20 % slower (16384 as 16x16x32 on
// four wavefronts again at the end of round 3: +-0 plain, -2 % with the bars of BASELINE configs[3]).
#pragma once
#include "wf_fft_core.hpp"

namespace wf {
using G512 = Geom<512, 64, 4, 8, 8>;         // four points per thread; both radix-8 passes shared by thread pairs.  256 and 128 run
                                              // zero-padded on it (DEC)
using G1024 = Geom<1024, 64, 4, 16, 8>;      // radix 4 first: pass 1 fetches 16-byte vectors (10 requests per thread instead of 23 eight-byte
                                              // ones), the radix-16 second pass shared by thread pairs: +1.3 % over 8x8x8 (0.693 -> 0.703), +1.8 % with bars
using G2048 = Geom<2048, 64, 8, 16, 8>;
using G4096 = Geom<4096, 128, 8, 16, 16>;     // two wavefronts
using G8192 = Geom<8192, 256, 8, 16, 32>;     // four wavefronts; radix 8 first so that pass 1 fetches 16-byte vectors (23 requests
                                              // per thread instead of the 47 eight-byte ones of 16x16x16: 53 -> 57 % of the HBM peak),
                                              // radix-32 last pass shared by thread pairs
// (measured alternatives in steady state: 2048 as 8x8x16 +-0, 4096 as 8x32x8 -2.5 %, 8192 as 8x32x16 -2 %)
using G16384 = Geom<16384, 512, 8, 32, 32>;   // eight wavefronts; radix 8 first (16-byte pass-1 vectors: 51.7 -> 53.5 % over
                                              // 16x16x32), both radix-32 passes shared by thread pairs
#ifndef WF_G32768_T
#define WF_G32768_T 512 // (1024 threads of 16 points: 0.54 / 0.48 of the HBM peak at 512 / 2048 streams where 512 threads of 32 reach 0.56 / 0.53)
#endif
using G32768 = Geom<32768, WF_G32768_T, 16, 32, 32>; // eight wavefronts of 32 points per thread = one workgroup per spectrum (132 KB of LDS: one per CU,
                                              // two waves per SIMD with 256 registers each -- 177-199 used, no scratch); both radix-32 passes
                                              // whole in a thread.  The reference's "large FFT" range.

// the row transform of the paths through device memory (wf_big.hpp), and the container of the Bluestein / mixed-radix
// instantiations of the largest geometry: 16384 complex points on 1024 threads of 16 (G32768's radices)
using GBig = Geom<32768, 1024, 16, 32, 32>;
constexpr uint32_t BIG_L2 = GBig::M;       // 16384
constexpr int BIG_TP = GBig::T * GBig::P;  // bins per epilogue workgroup (16384)

// calls f(G{}) for the geometry of fft_size n; returns false for unsupported sizes
template<class F> inline bool dispatch_geometry(uint32_t n, F &&f)
{
    // -DWF_GEOM_ONLY=<N>: development builds that instantiate one geometry (a full build compiles ~70 kernels)
#ifdef WF_GEOM_ONLY
#define WF_GEOM_CASE(N_, G_) case N_: if constexpr(N_ == WF_GEOM_ONLY) { f(G_{}); return true; } else return false;
#else
#define WF_GEOM_CASE(N_, G_) case N_: f(G_{}); return true;
#endif
    switch(n) {
    WF_GEOM_CASE(512, G512)
    WF_GEOM_CASE(1024, G1024)
    WF_GEOM_CASE(2048, G2048)
    WF_GEOM_CASE(4096, G4096)
    WF_GEOM_CASE(8192, G8192)
    WF_GEOM_CASE(16384, G16384)
    WF_GEOM_CASE(32768, G32768)
    default: return false;
    }
#undef WF_GEOM_CASE
}
} // namespace wf
