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
    int mode;              // 0: bars, 1: curve as a triangle strip, 2: curve as a line strip, 3: stepped bars
    int bar_stride, bar_width;
    float cpos, bottom, channel_offset, cap_radius;
    int rounded, cap_tris;
    int bottom_caps;       // !m_stereo || m_channel_spacing > 0 (:1645)
    int radial;            // m_radial: full-circle fans (:1632-1633, :1646-1647)
    int bot_offset;        // (m_rounded_caps && !m_stereo) || m_channel_spacing > 0 (:1619)
    // stepped bars (mode 3, :1583-1607)
    int step_width, step_stride, max_steps;
    uint32_t *counts;      // [n_streams][disp_ch] vertices of the row's draw call (constant unless stepped)
};

__global__ __launch_bounds__(256) void vertex_fill_kernel(const VertexArgs a)
{
    const uint32_t row = blockIdx.x; // (stream - stream_base) * disp_ch + channel
    const uint32_t stream = a.stream_base + row / a.disp_ch, channel = row % a.disp_ch;
    const float *vals = a.bars + ((size_t)stream * a.disp_ch + channel) * a.num_bars;
    f4 *out = a.verts + ((size_t)stream * a.disp_ch + channel) * a.per_row;
    if(a.mode == 3) {
        // Stepped bars (:1583-1607): bar i gets one quad of m_step_verts per step j whose y = j * step_stride lies under the
        // bar's height; the quads are packed in bar order, so a bar's first vertex is the sum of its predecessors' counts.
        // Every thread takes a contiguous run of bars; the runs' totals are scanned through LDS.
        __shared__ uint32_t run_sum[256];
        const int per = (a.num_bars + (int)blockDim.x - 1) / (int)blockDim.x;
        const int b0 = (int)threadIdx.x * per, b1 = min(b0 + per, a.num_bars);
        auto steps_of = [&](float val) {
            const float maxheight = a.cpos - val - a.channel_offset;
            int n = 0;
            for(int j = 0; j < a.max_steps; ++j) {
                if((float)(j * a.step_stride) >= maxheight)
                    break;
                ++n;
            }
            return n;
        };
        uint32_t mine = 0;
        for(int i = b0; i < b1; ++i)
            mine += (uint32_t)steps_of(vals[i]);
        run_sum[threadIdx.x] = mine;
        __syncthreads();
        for(uint32_t d = 1; d < blockDim.x; d <<= 1) { // inclusive scan
            const uint32_t v = threadIdx.x >= d ? run_sum[threadIdx.x - d] : 0u;
            __syncthreads();
            run_sum[threadIdx.x] += v;
            __syncthreads();
        }
        size_t vp = 6u * (size_t)(run_sum[threadIdx.x] - mine);
        const float sw = (float)a.step_width, bw = (float)a.bar_width;
        for(int i = b0; i < b1; ++i) {
            const float val = vals[i];
            const float x = (float)(i * a.bar_stride);
            const int n = steps_of(val);
            for(int j = 0; j < n; ++j, vp += 6) {
                float y = (float)(j * a.step_stride);
                if(channel)
                    y = a.cpos + y + a.channel_offset;
                else
                    y = a.cpos - y - a.channel_offset - sw;
                out[vp] = f4{0.0f + x, 0.0f + y, 0.0f, 0.0f};
                out[vp + 1] = f4{bw + x, 0.0f + y, 0.0f, 0.0f};
                out[vp + 2] = f4{0.0f + x, sw + y, 0.0f, 0.0f};
                out[vp + 3] = f4{bw + x, 0.0f + y, 0.0f, 0.0f};
                out[vp + 4] = f4{0.0f + x, sw + y, 0.0f, 0.0f};
                out[vp + 5] = f4{bw + x, sw + y, 0.0f, 0.0f};
            }
        }
        if(threadIdx.x == blockDim.x - 1)
            a.counts[(size_t)stream * a.disp_ch + channel] = 6u * run_sum[threadIdx.x];
        return;
    }
    if(threadIdx.x == 0)
        a.counts[(size_t)stream * a.disp_ch + channel] = (uint32_t)a.per_row;
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
            const int fan = a.radial ? a.cap_tris : half; // radial: full circles; else the half that faces away from the base line
            int start = a.radial ? 0 : (channel ? 0 : half);
            for(int j = start; j < start + fan; ++j, vp += 3) {
                v[vp] = f4{a.cap_xy[2 * j] + ccx, a.cap_xy[2 * j + 1] + val, 0.0f, 0.0f};
                v[vp + 1] = f4{a.cap_xy[2 * (j + 1)] + ccx, a.cap_xy[2 * (j + 1) + 1] + val, 0.0f, 0.0f};
                v[vp + 2] = f4{ccx, val, 0.0f, 0.0f};
            }
            if(a.bottom_caps) {
                const float ccy = a.cpos - offset;
                start = a.radial ? 0 : (channel ? half : 0);
                for(int j = start; j < start + fan; ++j, vp += 3) {
                    v[vp] = f4{a.cap_xy[2 * j] + ccx, a.cap_xy[2 * j + 1] + ccy, 0.0f, 0.0f};
                    v[vp + 1] = f4{a.cap_xy[2 * (j + 1)] + ccx, a.cap_xy[2 * (j + 1) + 1] + ccy, 0.0f, 0.0f};
                    v[vp + 2] = f4{ccx, ccy, 0.0f, 0.0f};
                }
            }
        }
    }
}

} // namespace wf
