// wf_geometry.hpp -- the FFT decompositions the library ships, one per supported FFT size.
// T threads per spectrum (1, 1, 2, 4, 8 wavefronts for N = 1024 ... 16384); every thread owns
// P = N/(2T) complex points (8 or 16) so pass 1 fetches 8/16-byte vectors.
#pragma once
#include "wf_fft_core.hpp"

namespace wf {
using G1024 = Geom<1024, 64, 8, 8, 8>;
using G2048 = Geom<2048, 64, 8, 16, 8>;
#ifndef WF_G4096_T
#define WF_G4096_T 128
#endif
#if WF_G4096_T == 64
using G4096 = Geom<4096, 64, 16, 16, 8>;   // one wavefront, 32 points per thread
#else
using G4096 = Geom<4096, 128, 8, 16, 16>;  // two wavefronts, 16 points per thread: half the registers, twice the waves
#endif
#ifndef WF_G8192_T
#define WF_G8192_T 256
#endif
#if WF_G8192_T == 128
using G8192 = Geom<8192, 128, 16, 16, 16>;  // two wavefronts, 32 points per thread
#else
using G8192 = Geom<8192, 256, 16, 16, 16>;  // four wavefronts, 16 points per thread (one radix-16 butterfly per pass)
#endif
#ifndef WF_G16384_T
#define WF_G16384_T 512
#endif
#if WF_G16384_T == 256
using G16384 = Geom<16384, 256, 16, 16, 32>; // four wavefronts, 32 points per thread
#else
using G16384 = Geom<16384, 512, 16, 16, 32>; // eight wavefronts, 16 points per thread; the radix-32 pass is shared by thread pairs
#endif

// calls f(G{}) for the geometry of fft_size n; returns false for unsupported sizes
template<class F> inline bool dispatch_geometry(uint32_t n, F &&f)
{
    switch(n) {
    case 1024: f(G1024{}); return true;
    case 2048: f(G2048{}); return true;
    case 4096: f(G4096{}); return true;
    case 8192: f(G8192{}); return true;
    case 16384: f(G16384{}); return true;
    default: return false;
    }
}
} // namespace wf
