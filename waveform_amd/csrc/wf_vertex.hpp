// wf_vertex.hpp -- gfx950 kernel of the vertex fill (device code only; hipcc).
//
// What it replaces (reference): the loops of WAVSource::render_bars (src/source.cpp:1576-1659, the plain-bar branch incl.
// rounded caps) and WAVSource::render_curve (:1436-1461) that turn the pixel rows in m_interp_bufs into the vertex buffer
// handed to gs_draw.  Input: the rows the tick has just left in `bars` ([stream][display channel][num_bars], pixel y per bar
// or curve point).  Output per displayed channel: the vertices of that channel's draw call, x / y / z / w as libobs' vec3
// (z = w = 0).  Every float below is formed by the reference's own operations in its order (integer products before the
// conversion, the same additions), so the x coordinates are bit-identical and y differs only by what the bar differs.
// Stepped bars and the radial layout are not in wf_config and stay with the host.
#pragma once
#include <hip/hip_runtime.h>
#include "wf_fft_core.hpp"

namespace wf {

struct VertexArgs {
    const float *bars;     // [n_streams][disp_ch][num_bars]
    f4 *verts;             // [n_streams][disp_ch][per_row]
    const float *cap_xy;   // [cap_tris + 1][2] m_cap_verts (src/source.cpp:1293-1309), or nullptr
    uint32_t stream_base, stream_count, disp_ch;
    int num_bars, per_row, per_bar;
    int mode;              // 0: bars, 1: curve as a triangle strip, 2: curve as a line strip
    int bar_stride, bar_width;
    float cpos, bottom, channel_offset, cap_radius;
    int rounded, cap_tris;
    int bottom_caps;       // !m_stereo || m_channel_spacing > 0 (:1645)
    int bot_offset;        // (m_rounded_caps && !m_stereo) || m_channel_spacing > 0 (:1619)
};

__global__ __launch_bounds__(256) void vertex_fill_kernel(const VertexArgs a)
{
    const uint32_t row = blockIdx.x; // (stream - stream_base) * disp_ch + channel
    const uint32_t stream = a.stream_base + row / a.disp_ch, channel = row % a.disp_ch;
    const float *vals = a.bars + ((size_t)stream * a.disp_ch + channel) * a.num_bars;
    f4 *out = a.verts + ((size_t)stream * a.disp_ch + channel) * a.per_row;
    if(a.mode != 0) {
        // render_curve :1436-1461: x = the column (set once by update(), :1027-1038), y = the point (channel 1 mirrored at
        // `bottom`), and in the filled modes a second vertex per column on the channel's base line
        const float offset = channel ? -a.channel_offset : a.channel_offset;
        const float bot = a.cpos - offset;
        for(int i = (int)threadIdx.x; i < a.num_bars; i += (int)blockDim.x) {
            const float val = vals[i];
            const float y = channel ? a.bottom - val : val;
            if(a.mode == 2) {
                out[i] = f4{(float)i, y, 0.0f, 0.0f};
            } else {
                out[2 * i] = f4{(float)i, y, 0.0f, 0.0f};
                out[2 * i + 1] = f4{(float)i, bot, 0.0f, 0.0f};
            }
        }
        return;
    }
    // render_bars :1609-1657
    const int half = a.cap_tris / 2;
    for(int i = (int)threadIdx.x; i < a.num_bars; i += (int)blockDim.x) {
        float val = vals[i];
        const float x1 = (float)(i * a.bar_stride);
        const float x2 = x1 + (float)a.bar_width;
        float offset = (a.rounded ? a.cap_radius : 0.0f) + a.channel_offset;
        if(channel) {
            val = a.bottom - val;
            offset = -offset;
        }
        const float bot = a.bot_offset ? (a.cpos - offset) : a.cpos;
        f4 *v = out + (size_t)i * a.per_bar;
        v[0] = f4{x1, val, 0.0f, 0.0f};
        v[1] = f4{x2, val, 0.0f, 0.0f};
        v[2] = f4{x1, bot, 0.0f, 0.0f};
        v[3] = f4{x2, val, 0.0f, 0.0f};
        v[4] = f4{x1, bot, 0.0f, 0.0f};
        v[5] = f4{x2, bot, 0.0f, 0.0f};
        if(a.rounded) {
            int vp = 6;
            const float ccx = (float)(i * a.bar_stride) + a.cap_radius; // cap centre
            int start = channel ? 0 : half; // (non-radial: the half of the circle that faces away from the base line)
            for(int j = start; j < start + half; ++j, vp += 3) {
                v[vp] = f4{a.cap_xy[2 * j] + ccx, a.cap_xy[2 * j + 1] + val, 0.0f, 0.0f};
                v[vp + 1] = f4{a.cap_xy[2 * (j + 1)] + ccx, a.cap_xy[2 * (j + 1) + 1] + val, 0.0f, 0.0f};
                v[vp + 2] = f4{ccx, val, 0.0f, 0.0f};
            }
            if(a.bottom_caps) {
                const float ccy = a.cpos - offset;
                start = channel ? half : 0;
                for(int j = start; j < start + half; ++j, vp += 3) {
                    v[vp] = f4{a.cap_xy[2 * j] + ccx, a.cap_xy[2 * j + 1] + ccy, 0.0f, 0.0f};
                    v[vp + 1] = f4{a.cap_xy[2 * (j + 1)] + ccx, a.cap_xy[2 * (j + 1) + 1] + ccy, 0.0f, 0.0f};
                    v[vp + 2] = f4{ccx, ccy, 0.0f, 0.0f};
                }
            }
        }
    }
}

} // namespace wf
