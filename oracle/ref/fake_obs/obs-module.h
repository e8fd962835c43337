/*
 * fake_obs/obs-module.h -- TEST INFRASTRUCTURE, not product code.
 *
 * A headless stand-in for the parts of the libobs public API that the
 * reference plugin (phandasm/waveform v1.9.1) touches.  libobs is NOT vendored
 * under /root/reference and is not installed in this image, so the reference's
 * src/source.cpp + src/module.cpp cannot be compiled as-is; with this header
 * (and fake_obs.cpp behind it) they compile VERBATIM and run without OBS:
 * settings are a string->variant map, graphics calls are no-ops that keep a
 * vertex buffer in host memory, the clock is a settable counter and audio
 * sources are objects the harness pushes packets through.
 *
 * Every declaration below restates a public libobs declaration (names,
 * argument order and the constants the plugin reads); nothing is copied from
 * the reference repository -- it only *uses* these (call sites:
 * src/source.cpp:45-499 properties/callbacks, :501-674 get_settings,
 * :676-780 audio capture, :935-1075 graphics objects, :1778-1888).
 */
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants ------------------------------------------------------- */
#define MAX_AV_PLANES 8
#define MAX_AUDIO_CHANNELS 8
#define AUDIO_OUTPUT_FRAMES 1024

enum { LOG_ERROR = 100, LOG_WARNING = 200, LOG_INFO = 300, LOG_DEBUG = 400 };

#define OBS_SOURCE_VIDEO (1 << 0)
#define OBS_SOURCE_AUDIO (1 << 1)
#define OBS_SOURCE_CUSTOM_DRAW (1 << 3)

enum obs_source_type { OBS_SOURCE_TYPE_INPUT, OBS_SOURCE_TYPE_FILTER, OBS_SOURCE_TYPE_TRANSITION, OBS_SOURCE_TYPE_SCENE };
enum obs_icon_type { OBS_ICON_TYPE_UNKNOWN, OBS_ICON_TYPE_IMAGE, OBS_ICON_TYPE_COLOR, OBS_ICON_TYPE_SLIDESHOW,
                     OBS_ICON_TYPE_AUDIO_INPUT, OBS_ICON_TYPE_AUDIO_OUTPUT };
enum obs_combo_type { OBS_COMBO_TYPE_INVALID, OBS_COMBO_TYPE_EDITABLE, OBS_COMBO_TYPE_LIST, OBS_COMBO_TYPE_RADIO };
enum obs_combo_format { OBS_COMBO_FORMAT_INVALID, OBS_COMBO_FORMAT_INT, OBS_COMBO_FORMAT_FLOAT, OBS_COMBO_FORMAT_STRING, OBS_COMBO_FORMAT_BOOL };

enum speaker_layout {
    SPEAKERS_UNKNOWN, SPEAKERS_MONO, SPEAKERS_STEREO, SPEAKERS_2POINT1, SPEAKERS_4POINT0,
    SPEAKERS_4POINT1, SPEAKERS_5POINT1, SPEAKERS_7POINT1 = 8
};
enum audio_format {
    AUDIO_FORMAT_UNKNOWN, AUDIO_FORMAT_U8BIT, AUDIO_FORMAT_16BIT, AUDIO_FORMAT_32BIT, AUDIO_FORMAT_FLOAT,
    AUDIO_FORMAT_U8BIT_PLANAR, AUDIO_FORMAT_16BIT_PLANAR, AUDIO_FORMAT_32BIT_PLANAR, AUDIO_FORMAT_FLOAT_PLANAR
};
enum gs_draw_mode { GS_POINTS, GS_LINES, GS_LINESTRIP, GS_TRIS, GS_TRISTRIP };
#define GS_DYNAMIC (1 << 1)

/* ---- opaque handles --------------------------------------------------- */
typedef struct obs_source obs_source_t;
typedef struct obs_weak_source obs_weak_source_t;
typedef struct obs_data obs_data_t;
typedef struct obs_properties obs_properties_t;
typedef struct obs_property obs_property_t;
typedef struct obs_module obs_module_t;
typedef struct gs_effect gs_effect_t;
typedef struct gs_effect_technique gs_technique_t;
typedef struct gs_effect_param gs_eparam_t;
typedef struct gs_vertex_buffer gs_vertbuffer_t;
typedef struct gs_index_buffer gs_indexbuffer_t;
typedef struct audio_output audio_t;

/* ---- math types (libobs graphics/vec{2,3,4}.h shape) -------------------- */
struct vec2 { union { struct { float x, y; }; float ptr[2]; }; };
struct vec3 { union { struct { float x, y, z, w; }; float ptr[4]; }; };
struct vec4 { union { struct { float x, y, z, w; }; float ptr[4]; }; };

static inline void vec2_set(struct vec2 *d, float x, float y) { d->x = x; d->y = y; }
static inline void vec3_set(struct vec3 *d, float x, float y, float z) { d->x = x; d->y = y; d->z = z; d->w = 0.0f; }
static inline void vec4_set(struct vec4 *d, float x, float y, float z, float w) { d->x = x; d->y = y; d->z = z; d->w = w; }
static inline void vec3_copy(struct vec3 *d, const struct vec3 *s) { *d = *s; }
static inline void vec3_add(struct vec3 *d, const struct vec3 *a, const struct vec3 *b)
{ d->x = a->x + b->x; d->y = a->y + b->y; d->z = a->z + b->z; d->w = 0.0f; }

/* ---- audio ------------------------------------------------------------ */
struct audio_data { uint8_t *data[MAX_AV_PLANES]; uint32_t frames; uint64_t timestamp; };
struct obs_audio_info { uint32_t samples_per_sec; enum speaker_layout speakers; };
struct obs_video_info { uint32_t fps_num, fps_den, base_width, base_height, output_width, output_height; };
struct audio_output_info { const char *name; uint32_t samples_per_sec; enum audio_format format; enum speaker_layout speakers; };
struct audio_convert_info { uint32_t samples_per_sec; enum audio_format format; enum speaker_layout speakers; bool allow_clipping; };

static inline uint32_t get_audio_channels(enum speaker_layout speakers)
{
    switch(speakers) {
    case SPEAKERS_MONO: return 1; case SPEAKERS_STEREO: return 2; case SPEAKERS_2POINT1: return 3;
    case SPEAKERS_4POINT0: return 4; case SPEAKERS_4POINT1: return 5; case SPEAKERS_5POINT1: return 6;
    case SPEAKERS_7POINT1: return 8; default: return 0;
    }
}
/* libobs util_mul_div64 + audio-io.h helpers */
static inline uint64_t util_mul_div64(uint64_t num, uint64_t mul, uint64_t div)
{ const uint64_t rem = num % div; return (num / div) * mul + (rem * mul) / div; }
static inline uint64_t audio_frames_to_ns(size_t sample_rate, uint64_t frames) { return util_mul_div64(frames, 1000000000ULL, sample_rate); }
static inline uint64_t ns_to_audio_frames(size_t sample_rate, uint64_t ns) { return util_mul_div64(ns, sample_rate, 1000000000ULL); }

typedef void (*obs_source_audio_capture_t)(void *param, obs_source_t *source, const struct audio_data *audio_data, bool muted);
typedef void (*audio_output_callback_t)(void *param, size_t mix_idx, struct audio_data *data);
typedef bool (*obs_enum_proc_t)(void *param, obs_source_t *source);
typedef bool (*obs_property_modified_t)(obs_properties_t *props, obs_property_t *property, obs_data_t *settings);

bool obs_get_audio_info(struct obs_audio_info *oai);
bool obs_get_video_info(struct obs_video_info *ovi);
audio_t *obs_get_audio(void);
const struct audio_output_info *audio_output_get_info(const audio_t *audio);
bool audio_output_connect(audio_t *audio, size_t mix_idx, const struct audio_convert_info *conversion, audio_output_callback_t callback, void *param);
void audio_output_disconnect(audio_t *audio, size_t mix_idx, audio_output_callback_t callback, void *param);

/* ---- sources ---------------------------------------------------------- */
struct obs_source_info {
    const char *id;
    enum obs_source_type type;
    uint32_t output_flags;
    const char *(*get_name)(void *type_data);
    void *(*create)(obs_data_t *settings, obs_source_t *source);
    void (*destroy)(void *data);
    uint32_t (*get_width)(void *data);
    uint32_t (*get_height)(void *data);
    void (*get_defaults)(obs_data_t *settings);
    obs_properties_t *(*get_properties)(void *data);
    void (*update)(void *data, obs_data_t *settings);
    void (*activate)(void *data);
    void (*deactivate)(void *data);
    void (*show)(void *data);
    void (*hide)(void *data);
    void (*video_tick)(void *data, float seconds);
    void (*video_render)(void *data, gs_effect_t *effect);
    enum obs_icon_type icon_type;
};
void obs_register_source_s(const struct obs_source_info *info, size_t size);
#define obs_register_source(info) obs_register_source_s(info, sizeof(struct obs_source_info))

void obs_enum_sources(obs_enum_proc_t enum_proc, void *param);
uint32_t obs_source_get_output_flags(const obs_source_t *source);
const char *obs_source_get_name(const obs_source_t *source);
bool obs_source_showing(const obs_source_t *source);
void obs_source_release(obs_source_t *source);
obs_source_t *obs_get_source_by_name(const char *name);
obs_weak_source_t *obs_source_get_weak_source(obs_source_t *source);
obs_source_t *obs_weak_source_get_source(obs_weak_source_t *weak);
void obs_weak_source_release(obs_weak_source_t *weak);
void obs_source_add_audio_capture_callback(obs_source_t *source, obs_source_audio_capture_t callback, void *param);
void obs_source_remove_audio_capture_callback(obs_source_t *source, obs_source_audio_capture_t callback, void *param);

/* ---- settings --------------------------------------------------------- */
const char *obs_data_get_string(obs_data_t *data, const char *name);
long long obs_data_get_int(obs_data_t *data, const char *name);
double obs_data_get_double(obs_data_t *data, const char *name);
bool obs_data_get_bool(obs_data_t *data, const char *name);
void obs_data_set_default_string(obs_data_t *data, const char *name, const char *val);
void obs_data_set_default_int(obs_data_t *data, const char *name, long long val);
void obs_data_set_default_double(obs_data_t *data, const char *name, double val);
void obs_data_set_default_bool(obs_data_t *data, const char *name, bool val);

/* ---- properties (UI description; kept only so get_properties() runs) ---- */
obs_properties_t *obs_properties_create(void);
void obs_properties_destroy(obs_properties_t *props);
obs_property_t *obs_properties_get(obs_properties_t *props, const char *property);
obs_property_t *obs_properties_add_bool(obs_properties_t *props, const char *name, const char *description);
obs_property_t *obs_properties_add_int(obs_properties_t *props, const char *name, const char *description, int min, int max, int step);
obs_property_t *obs_properties_add_int_slider(obs_properties_t *props, const char *name, const char *description, int min, int max, int step);
obs_property_t *obs_properties_add_float_slider(obs_properties_t *props, const char *name, const char *description, double min, double max, double step);
obs_property_t *obs_properties_add_list(obs_properties_t *props, const char *name, const char *description, enum obs_combo_type type, enum obs_combo_format format);
obs_property_t *obs_properties_add_color(obs_properties_t *props, const char *name, const char *description);
obs_property_t *obs_properties_add_color_alpha(obs_properties_t *props, const char *name, const char *description);
size_t obs_property_list_add_string(obs_property_t *p, const char *name, const char *val);
void obs_property_list_item_disable(obs_property_t *p, size_t idx, bool disabled);
void obs_property_set_modified_callback(obs_property_t *p, obs_property_modified_t modified);
void obs_property_set_visible(obs_property_t *p, bool visible);
bool obs_property_visible(obs_property_t *p);
void obs_property_set_enabled(obs_property_t *p, bool enabled);
void obs_property_set_long_description(obs_property_t *p, const char *long_description);
void obs_property_int_set_suffix(obs_property_t *p, const char *suffix);
void obs_property_float_set_suffix(obs_property_t *p, const char *suffix);
void obs_property_int_set_limits(obs_property_t *p, int min, int max, int step);

/* ---- graphics (headless) ------------------------------------------------ */
struct gs_tvertarray { size_t width; void *array; };
struct gs_vb_data {
    size_t num; struct vec3 *points; struct vec3 *normals; struct vec3 *tangents;
    uint32_t *colors; size_t num_tex; struct gs_tvertarray *tvarray;
};
void obs_enter_graphics(void);
void obs_leave_graphics(void);
struct gs_vb_data *gs_vbdata_create(void);
gs_vertbuffer_t *gs_vertexbuffer_create(struct gs_vb_data *data, uint32_t flags);
void gs_vertexbuffer_destroy(gs_vertbuffer_t *vertbuffer);
void gs_vertexbuffer_flush(gs_vertbuffer_t *vertbuffer);
struct gs_vb_data *gs_vertexbuffer_get_data(const gs_vertbuffer_t *vertbuffer);
void gs_load_vertexbuffer(gs_vertbuffer_t *vertbuffer);
void gs_load_indexbuffer(gs_indexbuffer_t *indexbuffer);
void gs_draw(enum gs_draw_mode draw_mode, uint32_t start_vert, uint32_t num_verts);
gs_effect_t *gs_effect_create_from_file(const char *file, char **error_string);
void gs_effect_destroy(gs_effect_t *effect);
gs_technique_t *gs_effect_get_technique(const gs_effect_t *effect, const char *name);
gs_eparam_t *gs_effect_get_param_by_name(const gs_effect_t *effect, const char *name);
size_t gs_technique_begin(gs_technique_t *technique);
void gs_technique_end(gs_technique_t *technique);
bool gs_technique_begin_pass(gs_technique_t *technique, size_t pass);
void gs_technique_end_pass(gs_technique_t *technique);
void gs_effect_set_bool(gs_eparam_t *param, bool val);
void gs_effect_set_float(gs_eparam_t *param, float val);
void gs_effect_set_vec2(gs_eparam_t *param, const struct vec2 *val);
void gs_effect_set_vec4(gs_eparam_t *param, const struct vec4 *val);

/* ---- memory / log / clock / module ---------------------------------------- */
void *bmalloc(size_t size);
void *bzalloc(size_t size);
void bfree(void *ptr);
void blog(int log_level, const char *format, ...);
uint64_t os_gettime_ns(void);

#define MODULE_EXPORT __attribute__((visibility("default")))
#define MODULE_EXTERN extern "C"
const char *obs_module_text(const char *lookup_string);
char *obs_module_file(const char *file);
#define OBS_DECLARE_MODULE() \
    extern "C" MODULE_EXPORT uint32_t obs_module_ver(void) { return (32u << 24) | (0u << 16) | 4u; }
#define OBS_MODULE_USE_DEFAULT_LOCALE(module_name, default_locale)

#ifdef __cplusplus
}
/* module.cpp defines these without a prior declaration; libobs declares them extern "C". */
extern "C" {
MODULE_EXPORT bool obs_module_load(void);
MODULE_EXPORT void obs_module_unload(void);
MODULE_EXPORT const char *obs_module_name(void);
MODULE_EXPORT const char *obs_module_description(void);
}
#endif
