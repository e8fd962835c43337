/*
 * Stand-in for google/cpu_features' cpuinfo_x86.h (TEST INFRASTRUCTURE).
 * The reference pulls cpu_features in as a git submodule (deps/cpu_features)
 * that is EMPTY in /root/reference, and uses it only at src/source.cpp:34-39 to
 * pick WAVSourceAVX2 / WAVSourceAVX / WAVSourceGeneric.  The oracle harness
 * decides that itself: env WF_REF_ISA = generic (default) | avx | avx2 sets the
 * three feature bits the plugin reads, so one build serves as the bit-exact
 * "generic" oracle and as the AVX2 CPU baseline.
 */
#pragma once
#include <cstdlib>
#include <cstring>
namespace cpu_features {
struct X86Features { int fma3, avx, avx2; };
struct X86Info { X86Features features; };
inline X86Info GetX86Info()
{
    X86Info i{};
    const char *isa = std::getenv("WF_REF_ISA");
    if(isa != nullptr && std::strcmp(isa, "avx2") == 0) { i.features.fma3 = 1; i.features.avx = 1; i.features.avx2 = 1; }
    else if(isa != nullptr && std::strcmp(isa, "avx") == 0) { i.features.fma3 = 1; i.features.avx = 1; }
    return i;
}
}
