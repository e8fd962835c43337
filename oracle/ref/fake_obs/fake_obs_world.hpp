/*
 * fake_obs_world.hpp -- TEST INFRASTRUCTURE.
 * The harness-facing side of the fake libobs (see obs-module.h): concrete
 * definitions of the "opaque" handles and the knobs a real OBS process would
 * own (audio configuration, frame rate, clock, audio sources).
 */
#pragma once
#include "obs-module.h"
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <array>
#include <vector>

namespace fakeobs {
struct Value {
    enum Kind { STRING, INT, DOUBLE, BOOL } kind = STRING;
    std::string s;
    long long i = 0;
    double d = 0.0;
    bool b = false;
};
}

struct obs_data {
    std::map<std::string, fakeobs::Value> vals;
    std::map<std::string, fakeobs::Value> defaults;
};

struct obs_source {
    std::string name;
    uint32_t flags = 0;
    bool showing = true;
    std::vector<std::pair<obs_source_audio_capture_t, void *>> audio_cbs;
    // libobs' obs_source::audio_cb_mutex (a recursive pthread mutex): held while the audio thread runs the capture callbacks
    // (source_signal_audio_data) and by add / remove -- which is why the reference's m_mtx is recursive and its capture_audio only
    // try_locks for 10 ms (src/source.hpp:98-101, src/source.cpp:1822-1824): update() removes the callback under m_mtx while the
    // audio thread may be inside it, holding this one
    std::recursive_mutex audio_cb_mtx;
};

namespace fakeobs {
void set_audio_info(uint32_t samples_per_sec, int channels);
void set_video_fps(uint32_t num, uint32_t den);
void set_clock_ns(uint64_t ns); // thread-local
uint64_t clock_ns();
void set_log_level(int level);
const obs_source_info *registered_source_info();

obs_source *create_source(const char *name, uint32_t flags);
void destroy_source(obs_source *src);
void push_audio(obs_source *src, const audio_data *audio, bool muted);

// every gs_draw call since the last clear_draws(): the draw mode, the vertex count and a copy of the loaded vertex buffer's
// first `num` points (x, y, z, w as OBS' vec3 holds them) -- what render_bars / render_curve hand to the GPU
// (mode -1: a gs_vertexbuffer_flush -- render_bars / render_curve flush once per displayed channel and then draw, unless a
// channel of stepped bars has no vertices at all, src/source.cpp:1661-1664)
struct Draw { int mode; uint32_t start, num; std::vector<float> points; };
std::vector<Draw> &draws(); // thread-local
std::map<std::string, std::array<float, 4>> &shader_values(); // thread-local: the last value every shader parameter was set to
void clear_draws();

obs_data *data_create();
void data_destroy(obs_data *d);
void data_set_from_text(obs_data *d, const char *name, const char *text);
}
