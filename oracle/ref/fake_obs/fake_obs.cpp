/*
 * fake_obs.cpp -- TEST INFRASTRUCTURE, not product code.
 *
 * Headless implementation of the libobs entry points declared in
 * fake_obs/obs-module.h, enough to run the reference plugin's
 * obs_source_info callbacks (create/update/video_tick/video_render and the
 * audio-capture callback) without OBS.  See obs-module.h for the rationale.
 *
 * World model:
 *   - settings       : std::map<string, value> with typed defaults
 *   - audio sources  : named objects; the harness pushes audio_data packets
 *                      through whatever callback the plugin registered
 *   - clock          : thread-local settable counter (os_gettime_ns)
 *   - graphics       : vertex buffers are plain host allocations; effects,
 *                      techniques and params are dummy non-null handles
 */
#include "obs-module.h"
#include "fake_obs_world.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <array>
#include <vector>

namespace fakeobs {

static std::mutex g_world_mtx;
static obs_audio_info g_audio_info{48000, SPEAKERS_STEREO};
static obs_video_info g_video_info{60, 1, 1920, 1080, 1920, 1080};
static obs_source_info g_registered{};
static bool g_have_registered = false;
static std::map<std::string, obs_source *> g_sources;
static thread_local uint64_t t_clock_ns = 1000000000ull;
static int g_log_level = LOG_WARNING;

void set_audio_info(uint32_t samples_per_sec, int channels)
{
    std::lock_guard lock(g_world_mtx);
    g_audio_info.samples_per_sec = samples_per_sec;
    switch(channels) {
    case 1: g_audio_info.speakers = SPEAKERS_MONO; break;
    case 2: g_audio_info.speakers = SPEAKERS_STEREO; break;
    case 3: g_audio_info.speakers = SPEAKERS_2POINT1; break;
    case 4: g_audio_info.speakers = SPEAKERS_4POINT0; break;
    case 5: g_audio_info.speakers = SPEAKERS_4POINT1; break;
    case 6: g_audio_info.speakers = SPEAKERS_5POINT1; break;
    case 8: g_audio_info.speakers = SPEAKERS_7POINT1; break;
    default: g_audio_info.speakers = SPEAKERS_UNKNOWN; break;
    }
}
void set_video_fps(uint32_t num, uint32_t den)
{
    std::lock_guard lock(g_world_mtx);
    g_video_info.fps_num = num;
    g_video_info.fps_den = den;
}
void set_clock_ns(uint64_t ns) { t_clock_ns = ns; }
uint64_t clock_ns() { return t_clock_ns; }
void set_log_level(int level) { g_log_level = level; }
const obs_source_info *registered_source_info() { return g_have_registered ? &g_registered : nullptr; }

obs_source *create_source(const char *name, uint32_t flags)
{
    auto src = new obs_source();
    src->name = name;
    src->flags = flags;
    std::lock_guard lock(g_world_mtx);
    g_sources[src->name] = src;
    return src;
}
void destroy_source(obs_source *src)
{
    if(src == nullptr)
        return;
    {
        std::lock_guard lock(g_world_mtx);
        g_sources.erase(src->name);
    }
    delete src;
}
void push_audio(obs_source *src, const audio_data *audio, bool muted)
{
    // index loop: the plugin may add/remove callbacks re-entrantly
    std::lock_guard lock(src->audio_cb_mtx);
    for(size_t i = 0; i < src->audio_cbs.size(); ++i) {
        auto cb = src->audio_cbs[i];
        cb.first(cb.second, src, audio, muted);
    }
}

obs_data *data_create() { return new obs_data(); }
void data_destroy(obs_data *d) { delete d; }

static Value *find(obs_data *d, const char *name)
{
    auto it = d->vals.find(name);
    if(it != d->vals.end())
        return &it->second;
    it = d->defaults.find(name);
    if(it != d->defaults.end())
        return &it->second;
    return nullptr;
}

void data_set_from_text(obs_data *d, const char *name, const char *text)
{
    Value v;
    const Value *def = nullptr;
    auto it = d->defaults.find(name);
    if(it != d->defaults.end())
        def = &it->second;
    auto kind = def ? def->kind : Value::STRING;
    if(def == nullptr) {
        // no default registered for this key: infer from the text
        char *end = nullptr;
        if(!strcmp(text, "true") || !strcmp(text, "false"))
            kind = Value::BOOL;
        else if((void)strtoll(text, &end, 10), (end != text && *end == '\0'))
            kind = Value::INT;
        else if((void)strtod(text, &end), (end != text && *end == '\0'))
            kind = Value::DOUBLE;
    }
    v.kind = kind;
    switch(kind) {
    case Value::BOOL: v.b = (!strcmp(text, "true") || !strcmp(text, "1")); break;
    case Value::INT: v.i = strtoll(text, nullptr, 10); break;
    case Value::DOUBLE: v.d = strtod(text, nullptr); break;
    default: v.s = text; break;
    }
    d->vals[name] = v;
}

} // namespace fakeobs

using namespace fakeobs;

// gs_draw's log (fake_obs_world.hpp)
struct gs_vertex_buffer;
static thread_local gs_vertex_buffer *g_loaded_vb = nullptr;
static thread_local std::vector<fakeobs::Draw> g_draws;
std::vector<fakeobs::Draw> &fakeobs::draws() { return g_draws; }
void fakeobs::clear_draws() { g_draws.clear(); }
// shader parameters by name: what set_shader_vars hands the effect during a render (thread-local, like the draws) -- the last value
// of each, four floats (a float parameter in [0], a bool as 0 / 1)
static thread_local std::map<std::string, std::array<float, 4>> g_shader_vals;
std::map<std::string, std::array<float, 4>> &fakeobs::shader_values() { return g_shader_vals; }

extern "C" {

/* ---- audio / video info ------------------------------------------------- */
bool obs_get_audio_info(obs_audio_info *oai) { std::lock_guard lock(g_world_mtx); *oai = g_audio_info; return true; }
bool obs_get_video_info(obs_video_info *ovi) { std::lock_guard lock(g_world_mtx); *ovi = g_video_info; return true; }
struct audio_output { audio_output_info info; };
static audio_output g_audio_out{{"fake", 48000, AUDIO_FORMAT_FLOAT_PLANAR, SPEAKERS_STEREO}};
audio_t *obs_get_audio(void) { return &g_audio_out; }
const audio_output_info *audio_output_get_info(const audio_t *audio) { return &audio->info; }
bool audio_output_connect(audio_t *, size_t, const audio_convert_info *, audio_output_callback_t, void *) { return false; }
void audio_output_disconnect(audio_t *, size_t, audio_output_callback_t, void *) {}

/* ---- sources -------------------------------------------------------------- */
void obs_register_source_s(const obs_source_info *info, size_t size)
{
    std::lock_guard lock(g_world_mtx);
    memset(&g_registered, 0, sizeof(g_registered));
    memcpy(&g_registered, info, size < sizeof(g_registered) ? size : sizeof(g_registered));
    g_have_registered = true;
}
void obs_enum_sources(obs_enum_proc_t enum_proc, void *param)
{
    std::vector<obs_source *> list;
    {
        std::lock_guard lock(g_world_mtx);
        for(auto &kv : g_sources)
            list.push_back(kv.second);
    }
    for(auto s : list)
        if(!enum_proc(param, s))
            break;
}
uint32_t obs_source_get_output_flags(const obs_source_t *source) { return source->flags; }
const char *obs_source_get_name(const obs_source_t *source) { return source->name.c_str(); }
bool obs_source_showing(const obs_source_t *source) { return source ? source->showing : true; }
void obs_source_release(obs_source_t *) {}
obs_source_t *obs_get_source_by_name(const char *name)
{
    std::lock_guard lock(g_world_mtx);
    auto it = g_sources.find(name);
    return (it == g_sources.end()) ? nullptr : it->second;
}
// The harness destroys a fake source only after the WAVSource that captured it, so a
// weak reference can simply hold the pointer (no global lock on the per-tick path;
// in real libobs this is a lock-free refcount check as well).
struct obs_weak_source { obs_source *src; };
obs_weak_source_t *obs_source_get_weak_source(obs_source_t *source) { return new obs_weak_source{source}; }
obs_source_t *obs_weak_source_get_source(obs_weak_source_t *weak) { return weak->src; }
void obs_weak_source_release(obs_weak_source_t *weak) { delete weak; }
void obs_source_add_audio_capture_callback(obs_source_t *source, obs_source_audio_capture_t callback, void *param)
{
    std::lock_guard lock(source->audio_cb_mtx);
    source->audio_cbs.emplace_back(callback, param);
}
void obs_source_remove_audio_capture_callback(obs_source_t *source, obs_source_audio_capture_t callback, void *param)
{
    std::lock_guard lock(source->audio_cb_mtx);
    auto &v = source->audio_cbs;
    for(size_t i = 0; i < v.size(); ++i)
        if(v[i].first == callback && v[i].second == param) { v.erase(v.begin() + (long)i); break; }
}

/* ---- settings --------------------------------------------------------------- */
const char *obs_data_get_string(obs_data_t *data, const char *name)
{
    auto v = find(data, name);
    return (v && v->kind == Value::STRING) ? v->s.c_str() : "";
}
long long obs_data_get_int(obs_data_t *data, const char *name)
{
    auto v = find(data, name);
    if(!v) return 0;
    return (v->kind == Value::INT) ? v->i : (v->kind == Value::DOUBLE) ? (long long)v->d : (v->kind == Value::BOOL) ? v->b : 0;
}
double obs_data_get_double(obs_data_t *data, const char *name)
{
    auto v = find(data, name);
    if(!v) return 0.0;
    return (v->kind == Value::DOUBLE) ? v->d : (v->kind == Value::INT) ? (double)v->i : 0.0;
}
bool obs_data_get_bool(obs_data_t *data, const char *name)
{
    auto v = find(data, name);
    if(!v) return false;
    return (v->kind == Value::BOOL) ? v->b : (v->kind == Value::INT) ? (v->i != 0) : false;
}
void obs_data_set_default_string(obs_data_t *data, const char *name, const char *val) { Value v; v.kind = Value::STRING; v.s = val; data->defaults[name] = v; }
void obs_data_set_default_int(obs_data_t *data, const char *name, long long val) { Value v; v.kind = Value::INT; v.i = val; data->defaults[name] = v; }
void obs_data_set_default_double(obs_data_t *data, const char *name, double val) { Value v; v.kind = Value::DOUBLE; v.d = val; data->defaults[name] = v; }
void obs_data_set_default_bool(obs_data_t *data, const char *name, bool val) { Value v; v.kind = Value::BOOL; v.b = val; data->defaults[name] = v; }

/* ---- properties ------------------------------------------------------------- */
struct obs_property { std::string name; bool visible = true; bool enabled = true; obs_property_modified_t modified = nullptr; };
struct obs_properties { std::map<std::string, std::unique_ptr<obs_property>> props; obs_property dummy; };
static obs_property_t *add_prop(obs_properties_t *props, const char *name)
{
    auto &p = props->props[name];
    if(!p) p = std::make_unique<obs_property>();
    p->name = name;
    return p.get();
}
obs_properties_t *obs_properties_create(void) { return new obs_properties(); }
void obs_properties_destroy(obs_properties_t *props) { delete props; }
obs_property_t *obs_properties_get(obs_properties_t *props, const char *property)
{
    auto it = props->props.find(property);
    return (it == props->props.end()) ? &props->dummy : it->second.get();
}
obs_property_t *obs_properties_add_bool(obs_properties_t *p, const char *n, const char *) { return add_prop(p, n); }
obs_property_t *obs_properties_add_int(obs_properties_t *p, const char *n, const char *, int, int, int) { return add_prop(p, n); }
obs_property_t *obs_properties_add_int_slider(obs_properties_t *p, const char *n, const char *, int, int, int) { return add_prop(p, n); }
obs_property_t *obs_properties_add_float_slider(obs_properties_t *p, const char *n, const char *, double, double, double) { return add_prop(p, n); }
obs_property_t *obs_properties_add_list(obs_properties_t *p, const char *n, const char *, obs_combo_type, obs_combo_format) { return add_prop(p, n); }
obs_property_t *obs_properties_add_color(obs_properties_t *p, const char *n, const char *) { return add_prop(p, n); }
obs_property_t *obs_properties_add_color_alpha(obs_properties_t *p, const char *n, const char *) { return add_prop(p, n); }
size_t obs_property_list_add_string(obs_property_t *, const char *, const char *) { return 0; }
void obs_property_list_item_disable(obs_property_t *, size_t, bool) {}
void obs_property_set_modified_callback(obs_property_t *p, obs_property_modified_t modified) { p->modified = modified; }
void obs_property_set_visible(obs_property_t *p, bool visible) { p->visible = visible; }
bool obs_property_visible(obs_property_t *p) { return p->visible; }
void obs_property_set_enabled(obs_property_t *p, bool enabled) { p->enabled = enabled; }
void obs_property_set_long_description(obs_property_t *, const char *) {}
void obs_property_int_set_suffix(obs_property_t *, const char *) {}
void obs_property_float_set_suffix(obs_property_t *, const char *) {}
void obs_property_int_set_limits(obs_property_t *, int, int, int) {}

/* ---- graphics ------------------------------------------------------------------ */
struct gs_vertex_buffer { gs_vb_data *data; };
struct gs_effect { int dummy; };
struct gs_effect_technique { int dummy; };
struct gs_effect_param { std::string name; };
static gs_effect_technique g_tech;
static thread_local std::map<std::string, gs_effect_param> g_params;
void obs_enter_graphics(void) {}
void obs_leave_graphics(void) {}
gs_vb_data *gs_vbdata_create(void) { return (gs_vb_data *)bzalloc(sizeof(gs_vb_data)); }
gs_vertbuffer_t *gs_vertexbuffer_create(gs_vb_data *data, uint32_t) { return new gs_vertex_buffer{data}; }
void gs_vertexbuffer_destroy(gs_vertbuffer_t *vb)
{
    if(vb == nullptr) return;
    if(vb->data != nullptr) {
        bfree(vb->data->points);
        if(vb->data->tvarray != nullptr) { bfree(vb->data->tvarray->array); bfree(vb->data->tvarray); }
        bfree(vb->data);
    }
    delete vb;
}
void gs_vertexbuffer_flush(gs_vertbuffer_t *) { g_draws.push_back(fakeobs::Draw{-1, 0, 0, {}}); } // logged: one flush per displayed channel, drawn or not
gs_vb_data *gs_vertexbuffer_get_data(const gs_vertbuffer_t *vb) { return vb->data; }
void gs_load_vertexbuffer(gs_vertbuffer_t *vb) { g_loaded_vb = vb; }
void gs_load_indexbuffer(gs_indexbuffer_t *) {}
void gs_draw(gs_draw_mode mode, uint32_t start, uint32_t num)
{
    fakeobs::Draw d{(int)mode, start, num, {}};
    if(g_loaded_vb != nullptr && g_loaded_vb->data != nullptr && g_loaded_vb->data->points != nullptr) {
        const size_t n = std::min<size_t>((size_t)start + num, g_loaded_vb->data->num);
        d.points.resize(n * 4);
        std::memcpy(d.points.data(), g_loaded_vb->data->points, n * 4 * sizeof(float));
    }
    g_draws.push_back(std::move(d));
}
gs_effect_t *gs_effect_create_from_file(const char *, char **) { return new gs_effect{0}; }
void gs_effect_destroy(gs_effect_t *effect) { delete effect; }
gs_technique_t *gs_effect_get_technique(const gs_effect_t *, const char *) { return &g_tech; }
gs_eparam_t *gs_effect_get_param_by_name(const gs_effect_t *, const char *name)
{
    auto &p = g_params[name ? name : ""];
    p.name = name ? name : "";
    return &p;
}
size_t gs_technique_begin(gs_technique_t *) { return 1; }
void gs_technique_end(gs_technique_t *) {}
bool gs_technique_begin_pass(gs_technique_t *, size_t) { return true; }
void gs_technique_end_pass(gs_technique_t *) {}
void gs_effect_set_bool(gs_eparam_t *p, bool v) { if(p) g_shader_vals[p->name] = {v ? 1.0f : 0.0f, 0, 0, 0}; }
void gs_effect_set_float(gs_eparam_t *p, float v) { if(p) g_shader_vals[p->name] = {v, 0, 0, 0}; }
void gs_effect_set_vec2(gs_eparam_t *p, const vec2 *v) { if(p && v) g_shader_vals[p->name] = {v->x, v->y, 0, 0}; }
void gs_effect_set_vec4(gs_eparam_t *p, const vec4 *v) { if(p && v) g_shader_vals[p->name] = {v->x, v->y, v->z, v->w}; }

/* ---- memory / log / clock / module ------------------------------------------------ */
void *bmalloc(size_t size) { return malloc(size ? size : 1); }
void *bzalloc(size_t size) { return calloc(1, size ? size : 1); }
void bfree(void *ptr) { free(ptr); }
void blog(int log_level, const char *format, ...)
{
    if(log_level > g_log_level)
        return;
    va_list args;
    va_start(args, format);
    vfprintf(stderr, format, args);
    fputc('\n', stderr);
    va_end(args);
}
uint64_t os_gettime_ns(void) { return t_clock_ns; }
const char *obs_module_text(const char *lookup_string) { return lookup_string; }
char *obs_module_file(const char *file)
{
    size_t n = strlen(file) + 1;
    auto p = (char *)bmalloc(n);
    memcpy(p, file, n);
    return p;
}

} // extern "C"
