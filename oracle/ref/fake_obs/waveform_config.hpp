/*
 * What the reference's CMake would generate from src/waveform_config.hpp.in
 * (CMakeLists.txt:214) for an x86-64 Linux build with ENABLE_X86_SIMD=ON.
 * TEST INFRASTRUCTURE for the oracle build only.
 */
#pragma once
#define HAVE_OBS_PROP_ALPHA
#define ENABLE_X86_SIMD
#define WAVEFORM_VERSION "1.9.1"
#define WAVEFORM_ARCH "x64"
#define WAV_FORCE_INLINE __attribute__((always_inline)) inline
