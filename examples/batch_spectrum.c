/*
 * batch_spectrum.c -- the C ABI of libwaveform_hip.so from plain C: a batch of stereo sources, audio in, ticks, bar heights
 * and spectra out.  What a host other than the OBS plugin (a server analysing many streams) writes; the plugin-side binding
 * is host/wav_source_hip.cpp.
 *
 *   gcc -std=c99 -Iinclude examples/batch_spectrum.c -Lwaveform_amd -lwaveform_hip -lm -o batch_spectrum
 *   LD_LIBRARY_PATH=waveform_amd ./batch_spectrum
 */
#include <stdio.h>
#include <stdlib.h>
#include "wf_hip.h"
#include "wf_synth.h"

#define STREAMS 64u
#define HOP 800u /* 48 kHz audio, 60 video frames per second */

int main(void)
{
    wf_config cfg;
    wf_config_defaults(&cfg); /* the plugin's get_defaults: FFT 4096, Hann, EMA 0.65, mono mixdown, ... */
    cfg.stereo = 1;
    cfg.bars = 1; /* 26 bars per channel from the default geometry */
    cfg.interp_mode = WF_INTERP_LANCZOS;

    wf_hip *h = NULL;
    int rc = wf_hip_create(&cfg, 0, STREAMS, 0, &h);
    if(rc != WF_HIP_OK) {
        fprintf(stderr, "wf_hip_create: %d (%s)\n", rc, wf_hip_last_error(NULL));
        return 1; /* e.g. WF_HIP_ERR_NO_DEVICE: there is no CPU path in the library */
    }
    const uint32_t ch = wf_hip_capture_channels(h), bars = wf_hip_num_bars(h), disp = wf_hip_display_channels(h);
    float *packet = (float *)malloc(sizeof(float) * STREAMS * ch * HOP);   /* [stream][channel][frame] */
    float *tops = (float *)malloc(sizeof(float) * STREAMS * disp * bars); /* pixel row of every bar's top */

    for(uint32_t tick = 0; tick < 120 && rc == WF_HIP_OK; ++tick) {
        for(uint32_t s = 0; s < STREAMS; ++s)
            for(uint32_t c = 0; c < ch; ++c)
                for(uint32_t i = 0; i < HOP; ++i)
                    packet[(s * ch + c) * HOP + i] = 0.25f * wf_synth_noise(1234u, s, c, (uint64_t)tick * HOP + i);
        rc = wf_hip_push_audio(h, 0, STREAMS, packet, HOP); /* capture_audio for every source */
        wf_hip_tick_params p = {1.0f / 60.0f, 0, 0.0f, 0, 0};
        if(rc == WF_HIP_OK)
            rc = wf_hip_tick(h, &p);                        /* tick_spectrum + the bar reduction, one kernel launch */
    }
    if(rc == WF_HIP_OK)
        rc = wf_hip_read(h, WF_HIP_OUT_BARS, 0, STREAMS, tops);
    if(rc != WF_HIP_OK)
        fprintf(stderr, "wf_hip: %d (%s)\n", rc, wf_hip_last_error(h));
    else
        printf("%s\nstream 0, left channel: bar 0 top at y = %.2f px, bar %u at y = %.2f px\n", wf_hip_kernel_name(h), tops[0],
               bars - 1, tops[bars - 1]);
    free(tops);
    free(packet);
    wf_hip_destroy(h);
    return rc == WF_HIP_OK ? 0 : 1;
}
