// wf_emu.cpp -- TEST HARNESS: a lane-by-lane wavefront emulator for the tick kernel.
//
// Compiles the *same* phase functions the gfx950 kernel is made of
// (waveform_amd/csrc/wf_tick_phases.hpp) with g++ and runs them thread by thread on
// the host, phase by phase, so that on the GPU-less build box we can check the FFT
// index algebra / epilogue against the oracle and count LDS bank conflicts with the
// bank model of MI355X_MICROARCH.md (LDS section).  Nothing here is part of the
// product: libwaveform_hip.so never links this file and has no CPU path.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <algorithm>

// ---- LDS access trace ---------------------------------------------------------------------
namespace emu {
struct Access { int idx; int bytes; int is_write; };
static thread_local std::vector<Access> *g_trace = nullptr;
// bounds check of every exchange-buffer access ("a debug build that bounds-checks LDS indices"): idx counts
// complex points (8 bytes); g_lds_cf = the spectrum's buffer in complex points (Geom::LDS_CF), 0 = not armed
static thread_local int g_lds_cf = 0;
static thread_local long g_lds_violations = 0;
inline void trace(int idx, int bytes, int is_write)
{
    if(g_lds_cf > 0 && (idx < 0 || (long)idx * 8 + bytes > (long)g_lds_cf * 8))
        ++g_lds_violations;
    if(g_trace)
        g_trace->push_back({idx, bytes, is_write});
}
}
#define WF_LDS_TRACE(idx, bytes, is_write) ::emu::trace((idx), (bytes), (is_write))

#include "../../waveform_amd/csrc/wf_geometry.hpp"
#include "../../waveform_amd/csrc/wf_tick_phases.hpp"
#include "../../waveform_amd/csrc/wf_host_tables.hpp"
#include "../../include/wf_hip.h"

namespace {

// lane groups serviced together, per instruction kind (MI355X_MICROARCH.md §LDS)
// returns group id of lane l (0..63) and the bank modulus
struct Rule { int ngroups; int modulus; int (*group)(int lane); };
int grp_half(int l) { return l >> 5; }
int grp_b128_read(int l)
{
    // 4 x 16: {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}
    const int h = l >> 5, m = l & 31;
    const bool first = (m < 4) || (m >= 12 && m < 16) || (m >= 20 && m < 28);
    return h * 2 + (first ? 0 : 1);
}
int grp_contig16(int l) { return l >> 4; }
int grp_contig8(int l) { return l >> 3; }

Rule rule_for(int bytes, int is_write)
{
    if(!is_write) {
        if(bytes == 8) return {2, 64, grp_half};          // ds_read_b64
        return {4, 64, grp_b128_read};                    // ds_read_b128
    }
    if(bytes == 8) return {4, 32, grp_contig16};          // ds_write_b64
    return {8, 32, grp_contig8};                          // ds_write_b128
}

struct ConflictStats { uint64_t instr = 0, ideal_cycles = 0, actual_cycles = 0; };

// per-wave census: traces[lane] = sequence of accesses of that lane in this phase
void census(const std::vector<std::vector<emu::Access>> &traces, ConflictStats st[2])
{
    const size_t n = traces[0].size();
    for(size_t i = 0; i < n; ++i) {
        const auto a0 = traces[0][i];
        const Rule r = rule_for(a0.bytes, a0.is_write);
        ConflictStats &s = st[a0.is_write ? 1 : 0];
        s.instr++;
        for(int g = 0; g < r.ngroups; ++g) {
            // bank -> set of distinct dword addresses
            std::map<int, std::vector<int>> banks;
            for(int l = 0; l < 64; ++l) {
                if(r.group(l) != g)
                    continue;
                const auto a = traces[(size_t)l][i];
                const int dw0 = a.idx * 2; // cf index -> dword index
                for(int d = 0; d < a.bytes / 4; ++d) {
                    const int dw = dw0 + d;
                    auto &v = banks[dw % r.modulus];
                    if(std::find(v.begin(), v.end(), dw) == v.end())
                        v.push_back(dw);
                }
            }
            size_t worst = 1;
            for(auto &kv : banks)
                worst = std::max(worst, kv.second.size());
            s.ideal_cycles += 1;
            s.actual_cycles += worst;
        }
    }
}

template<class G> int run_geometry(const wf_config &cfg, const wf::HostTables &tab, uint32_t n_streams, uint32_t ring_cap,
                                   const float *ring, const uint32_t *wpos, uint32_t delay, float seconds, float *tsmooth,
                                   float *decibels, uint64_t *stats)
{
    using namespace wf;
    std::vector<cfloat> tw1, tw2, tws;
    build_twiddles(G::M, G::R1, G::R2, G::R3, tw1, tw2, tws);

    TickArgs a{};
    a.ring = ring;
    a.wpos = wpos;
    a.ring_cap = ring_cap;
    a.ring_mask = ring_cap - 1;
    a.delay = delay;
    const std::vector<float> ones_n((size_t)G::N, 1.0f), ones_m((size_t)G::M, 1.0f);
    a.window = tab.window.empty() ? ones_n.data() : tab.window.data();
    a.tw1 = reinterpret_cast<const cf *>(tw1.data());
    a.tw2 = reinterpret_cast<const cf *>(tw2.data());
    a.tws = reinterpret_cast<const cf *>(tws.data());
    a.slope = tab.slope.empty() ? ones_m.data() : tab.slope.data();
    a.rolloff = tab.rolloff.empty() ? nullptr : tab.rolloff.data();
    a.tsmooth = tsmooth;
    a.decibels = decibels;
    a.half_coef = 0.5f * (2.0f / tab.window_sum);
    a.slope_step = tab.slope.empty() ? 0.0f : (float)(3.0 * (double)cfg.slope / (double)(cfg.fft_size / 2 - 1)); // Policy<G>::SLOPE_LINEAR (as make_args sets it)
    a.g = gravity_for(cfg, seconds);
    a.g2 = 1.0f - a.g;
    a.db_min = db_min();
    a.n_streams = n_streams;
    a.cap_ch = cfg.capture_channels;
    a.out_ch = tab.output_channels;
    a.mode = 0;
    if(cfg.tsmoothing != WF_TSMOOTH_NONE) a.mode |= WF_MODE_TSMOOTH;
    if(cfg.fast_peaks) a.mode |= WF_MODE_FAST_PEAKS;
    if(cfg.stereo) a.mode |= WF_MODE_STEREO;
    if(!tab.slope.empty()) a.mode |= WF_MODE_SLOPE;
    if(!tab.rolloff.empty()) a.mode |= WF_MODE_ROLLOFF;
    if(!tab.window.empty()) a.mode |= WF_MODE_WINDOW;

    constexpr int T = G::T, P = G::P, M = G::M;
    ConflictStats st[2];
    std::vector<cf> lds((size_t)G::LDS_CF); // (exactly the kernel's buffer: the AddressSanitizer build of this file sees an access past it)
    emu::g_lds_cf = G::LDS_CF;
    std::vector<std::vector<emu::Access>> traces((size_t)T);
    struct Regs { cf v[P]; float mag[P]; float d[P]; P1Regs<G> r1; P4Regs<G> r4; };
    std::vector<Regs> regs((size_t)T);

    int phase_no = 0; // development aid: WF_EMU_VERBOSE prints the census per phase
    auto census_waves = [&](bool enable) {
        if(!enable) return;
        for(int w = 0; w < T / 64; ++w) {
            std::vector<std::vector<emu::Access>> sub(traces.begin() + w * 64, traces.begin() + (w + 1) * 64);
            census(sub, st);
        }
    };
    auto phase = [&](bool do_census, auto &&fn) {
        for(int t = 0; t < T; ++t) {
            traces[(size_t)t].clear();
            emu::g_trace = &traces[(size_t)t];
            fn(t);
        }
        emu::g_trace = nullptr;
        const ConflictStats before[2] = {st[0], st[1]};
        census_waves(do_census);
        if(do_census && getenv("WF_EMU_VERBOSE"))
            fprintf(stderr, "[emu N=%d] phase %d: reads %llu instr ideal %llu actual %llu | writes %llu instr ideal %llu actual %llu\n", G::N, phase_no,
                    (unsigned long long)(st[0].instr - before[0].instr), (unsigned long long)(st[0].ideal_cycles - before[0].ideal_cycles),
                    (unsigned long long)(st[0].actual_cycles - before[0].actual_cycles), (unsigned long long)(st[1].instr - before[1].instr),
                    (unsigned long long)(st[1].ideal_cycles - before[1].ideal_cycles), (unsigned long long)(st[1].actual_cycles - before[1].actual_cycles));
        ++phase_no;
    };

    const uint32_t n_spec = n_streams * cfg.capture_channels;
    for(uint32_t spec = 0; spec < n_spec; ++spec) {
        const uint32_t stream = spec / cfg.capture_channels, ch = spec % cfg.capture_channels;
        const float *x = ring + (size_t)spec * ring_cap;
        const uint32_t start = (wpos[stream] - delay - (uint32_t)G::N) & a.ring_mask;
        float *ts = tsmooth ? tsmooth + (size_t)spec * M : nullptr;
        float *out = decibels + ((size_t)stream * a.out_ch + ch) * M;
        const bool c = (spec == 0);
        const bool aligned = (start % 4u) == 0;
        std::fill(lds.begin(), lds.end(), cf{1e30f, 1e30f}); // poison: unwritten reads show up
        phase(c, [&](int t) {
            if(aligned) p1_fetch<G, true>(a, t, x, start, regs[(size_t)t].r1);
            else p1_fetch<G, false>(a, t, x, start, regs[(size_t)t].r1);
            p1_window_pass1<G>(a, t, regs[(size_t)t].r1, lds.data());
            p4_prefetch<G>(a, t, ts, regs[(size_t)t].r4);
        });
        phase(c, [&](int t) { p2_read<G>(t, lds.data(), regs[(size_t)t].v); });
        phase(false, [&](int t) { p2_pass2_write<G>(a.tw2, t, lds.data(), regs[(size_t)t].v); });
        phase(c, [&](int t) { p3_read<G>(t, lds.data(), regs[(size_t)t].v); });
        phase(c, [&](int t) { p3_pass3_write<G>(t, lds.data(), regs[(size_t)t].v); });
        phase(c, [&](int t) { p4_split_smooth<G>(a, t, lds.data(), ts, regs[(size_t)t].r1.wb, regs[(size_t)t].r4, regs[(size_t)t].mag); });
        phase(false, [&](int t) {
            p4_db<G>(a, t, regs[(size_t)t].mag, regs[(size_t)t].d, a.vol_comp);
            store_row<G>(out, t, regs[(size_t)t].d);
        });
    }
    if(stats) {
        stats[0] = st[0].instr; stats[1] = st[0].ideal_cycles; stats[2] = st[0].actual_cycles;
        stats[3] = st[1].instr; stats[4] = st[1].ideal_cycles; stats[5] = st[1].actual_cycles;
    }
    return 0;
}

} // namespace

extern "C" {

// Emulates wf_hip_tick for n_streams streams on host memory.  Mono mixdown is not emulated
// (the emulator checks per-spectrum math; the cross-wave paths are covered on the GPU).
// stats[6] = {read instr, read ideal cycles, read actual cycles, write instr, ideal, actual}
int wfemu_tick(const wf_config *cfg, uint32_t n_streams, uint32_t ring_cap, const float *ring, const uint32_t *wpos,
               uint32_t delay, float seconds, float *tsmooth, float *decibels, uint64_t *stats)
{
    wf::HostTables tab;
    const int rc = wf::build_host_tables(*cfg, tab);
    if(rc != 0)
        return rc;
    int ret = -1;
    const bool ok = wf::dispatch_geometry(cfg->fft_size, [&](auto g) {
        using G = decltype(g);
        ret = run_geometry<G>(*cfg, tab, n_streams, ring_cap, ring, wpos, delay, seconds, tsmooth, decibels, stats);
    });
    if(ok && ret == 0 && emu::g_lds_violations != 0) {
        emu::g_lds_violations = 0;
        return -77; // an exchange-buffer access outside the spectrum's LDS buffer
    }
    return ok ? ret : WF_HIP_ERR_UNSUPPORTED;
}

// host tables of the product (waveform_amd/csrc/wf_host_tables.cpp) without a device.
// which: 0 window, 1 slope, 2 rolloff, 3 interp_indices, 4 interp_weights, 5 band_widths (as float),
//        7 bar_coef, 8 bar_bin, 9 bar_off (device form of the bar reduction, flat),
//        6 scalars {window_sum, gravity(seconds), db_min, num_bars, radius, taps, border_top, border_bottom, out_ch}
// returns the element count (copies at most cap elements), or a negative error
long wfemu_host_table(const wf_config *cfg, int which, float seconds, float *out, long cap)
{
    wf::HostTables tab;
    const int rc = wf::build_host_tables(*cfg, tab);
    if(rc != 0)
        return rc;
    std::vector<float> v;
    switch(which) {
    case 0: v = tab.window; break;
    case 1: v = tab.slope; break;
    case 2: v = tab.rolloff; break;
    case 3: v = tab.interp_indices; break;
    case 4: v = tab.interp_weights; break;
    case 5: v.assign(tab.band_widths.begin(), tab.band_widths.end()); break;
    case 7: v = tab.bar_coef; break;
    case 8: v.assign(tab.bar_bin.begin(), tab.bar_bin.end()); break;
    case 9: v.assign(tab.bar_off.begin(), tab.bar_off.end()); break;
    case 6:
        v = {tab.window_sum, wf::gravity_for(*cfg, seconds), wf::db_min(), (float)tab.num_bars, (float)tab.interp_radius,
             (float)tab.interp_taps, tab.border_top, tab.border_bottom, (float)tab.output_channels};
        break;
    default: return -1;
    }
    for(long i = 0; i < (long)v.size() && i < cap; ++i)
        out[i] = v[(size_t)i];
    return (long)v.size();
}

// the per-thread segment form of the bar tables (wf::bar_segments) for `threads` threads per spectrum.
// which: 0 lane coef [blocks][threads][4], 1 lane base [threads] (as float), 2 bar_seg (as float), 3 scalars {num_segs, blocks}
// returns the element count, -1000 if the segment form does not exist for this configuration, other negatives on error
long wfemu_bar_lanes(const wf_config *cfg, int threads, int max_blocks, int which, float *out, long cap)
{
    wf::HostTables tab;
    const int rc = wf::build_host_tables(*cfg, tab);
    if(rc != 0)
        return rc;
    wf::BarLaneTables lanes;
    if(!wf::bar_segments(tab, threads, max_blocks, lanes))
        return -1000;
    std::vector<float> v;
    switch(which) {
    case 0: v = lanes.coef; break;
    case 1: v.assign(lanes.base.begin(), lanes.base.end()); break;
    case 2: v.assign(lanes.bar_seg.begin(), lanes.bar_seg.end()); break;
    case 3: v = {(float)lanes.num_segs, (float)lanes.blocks}; break;
    default: return -1;
    }
    for(long i = 0; i < (long)v.size() && i < cap; ++i)
        out[i] = v[(size_t)i];
    return (long)v.size();
}

// the wave-private form (wf::bar_pieces).  which: 0 coef [blocks][threads][4], 1 base, 2 info, 3 bar_piece (ints as float),
// 4 scalars {num_segs, blocks, num_slots}; -1000 if the form does not exist for this configuration
long wfemu_bar_pieces(const wf_config *cfg, int threads, int points, int max_blocks, int which, float *out, long cap)
{
    wf::HostTables tab;
    const int rc = wf::build_host_tables(*cfg, tab);
    if(rc != 0)
        return rc;
    wf::BarPieceTables pc;
    if(!wf::bar_pieces(tab, threads, points, max_blocks, pc))
        return -1000;
    std::vector<float> v;
    switch(which) {
    case 0: v = pc.coef; break;
    case 1: v.assign(pc.base.begin(), pc.base.end()); break;
    case 2: v.assign(pc.info.begin(), pc.info.end()); break;
    case 3: v.assign(pc.bar_piece.begin(), pc.bar_piece.end()); break;
    case 4: v = {(float)pc.num_segs, (float)pc.blocks, (float)pc.num_slots}; break;
    default: return -1;
    }
    for(long i = 0; i < (long)v.size() && i < cap; ++i)
        out[i] = v[(size_t)i];
    return (long)v.size();
}

// the prefix-sum form (wf::bar_ps).  which: 0 the lane table [3][64][4] (integers as bit patterns), 1 scalars
// {num_lanes, num_subs}; -1000 if the form does not exist for this configuration
long wfemu_bar_ps(const wf_config *cfg, int threads, int which, float *out, long cap)
{
    wf::HostTables tab;
    const int rc = wf::build_host_tables(*cfg, tab);
    if(rc != 0)
        return rc;
    wf::BarPsTables ps;
    if(!wf::bar_ps(tab, threads, ps))
        return -1000;
    std::vector<float> v;
    switch(which) {
    case 0: v = ps.tab; break;
    case 1: v = {(float)ps.num_lanes, (float)ps.num_subs}; break;
    default: return -1;
    }
    for(long i = 0; i < (long)v.size() && i < cap; ++i)
        out[i] = v[(size_t)i];
    return (long)v.size();
}

// the tables of the rows-by-Bluestein form of the sizes above 16384 (wf::build_bluestein_rows).  which: 0 column twiddle x opening chirp
// [C][R], 1 FFT_L(chirp) [L], 2 closing chirp / L [R]; interleaved re, im.  Returns the floats of the table (L with out == nullptr && which < 0)
long wfemu_bluestein_rows(uint32_t np, uint32_t c, int which, float *out, long cap)
{
    std::vector<wf::cfloat> rowtw, bhat, q;
    const uint32_t L = wf::build_bluestein_rows(np, c, rowtw, bhat, q);
    if(which < 0)
        return (long)L;
    const std::vector<wf::cfloat> &v = which == 0 ? rowtw : which == 1 ? bhat : q;
    for(long i = 0; i < (long)v.size() && 2 * i + 1 < cap; ++i) {
        out[2 * i] = v[(size_t)i].re;
        out[2 * i + 1] = v[(size_t)i].im;
    }
    return 2 * (long)v.size();
}

int wfemu_lds_bytes(uint32_t fft_size)
{
    int r = -1;
    wf::dispatch_geometry(fft_size, [&](auto g) { r = decltype(g)::LDS_CF * 8; });
    return r;
}

} // extern "C"
