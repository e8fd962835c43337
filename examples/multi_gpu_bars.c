/*
 * multi_gpu_bars.c -- one batch of stereo sources over every GPU of the node, from plain C: the streams shard contiguously
 * (wf_hip_multi_create), every device ticks its shard on its own host thread inside the library, and after every tick the
 * bar heights of ALL streams are gathered onto EVERY device (RCCL over xGMI through a dlopen()ed librccl.so; peer copies
 * where it is absent) under the next tick -- BASELINE configs[4]'s exchange.  The combined picture is read from device 0.
 *
 *   gcc -std=c99 -Iinclude examples/multi_gpu_bars.c -Lwaveform_amd -lwaveform_hip -lm -o multi_gpu_bars
 *   LD_LIBRARY_PATH=waveform_amd ./multi_gpu_bars
 */
#include <stdio.h>
#include <stdlib.h>
#include "wf_hip.h"

#define STREAMS_PER_DEVICE 1024u
#define HOP 800u

int main(void)
{
    wf_config cfg;
    wf_config_defaults(&cfg);
    cfg.stereo = 1;
    cfg.bars = 1;
    cfg.interp_mode = WF_INTERP_LANCZOS;

    int n = wf_hip_device_count();
    if(n <= 0) {
        fprintf(stderr, "no HIP device: %d\n", (int)WF_HIP_ERR_NO_DEVICE);
        return 1; /* there is no CPU path in the library */
    }
    if(n > 64)
        n = 64;
    int devices[64];
    for(int i = 0; i < n; ++i)
        devices[i] = i;
    const uint32_t total = STREAMS_PER_DEVICE * (uint32_t)n;
    wf_hip_multi *m = NULL;
    int rc = wf_hip_multi_create(&cfg, devices, (uint32_t)n, total, 0, &m);
    if(rc != WF_HIP_OK) {
        fprintf(stderr, "wf_hip_multi_create: %d (%s)\n", rc, wf_hip_multi_last_error(NULL));
        return 1;
    }
    uint32_t first = 0, count = 0;
    wf_hip *shard0 = wf_hip_multi_shard(m, 0, NULL, &first, &count);
    const uint32_t bars = wf_hip_num_bars(shard0), disp = wf_hip_display_channels(shard0);
    float *all = (float *)malloc(sizeof(float) * total * disp * bars);
    wf_hip_tick_params p = {1.0f / 60.0f, 0, 0.0f, WF_HIP_TICK_NO_DECIBELS, 0}; /* bars only: the rows stay on nobody's bus */
    for(uint32_t tick = 0; tick < 120 && rc == WF_HIP_OK; ++tick) {
        /* every stream its own noise, generated on the devices (a live host calls wf_hip_multi_push_audio here) */
        rc = wf_hip_multi_push_synth(m, 0, total, 1234u, 0, (uint64_t)tick * HOP, HOP);
        if(rc == WF_HIP_OK)
            rc = wf_hip_multi_tick(m, &p);
        if(rc == WF_HIP_OK)
            rc = wf_hip_multi_allgather_bars(m); /* asynchronous: runs under the next tick */
    }
    if(rc == WF_HIP_OK)
        rc = wf_hip_multi_read_gathered(m, 0, all); /* device 0's copy of everybody's bars */
    if(rc != WF_HIP_OK)
        fprintf(stderr, "wf_hip_multi: %d (%s)\n", rc, wf_hip_multi_last_error(m));
    else
        printf("%u streams over %d device(s), transport %s; shard 0 = streams [%u, %u); stream %u, left channel, bar 0 top at y = %.2f px\n",
               total, n, wf_hip_multi_transport(m), first, first + count, total - 1, all[(size_t)(total - 1) * disp * bars]);
    free(all);
    wf_hip_multi_destroy(m);
    return rc == WF_HIP_OK ? 0 : 1;
}
