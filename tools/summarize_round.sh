#!/bin/bash
# tools/summarize_round.sh TAG -- after a `gpurun ... tools/gpu_round.sh TAG` has merged its raw rocprofv3 output into
# gpurun_out/prof/, turns every shape's passes into the committed summaries profiles/TAG_<shape>_{kernel_stats.csv,pmc.json}.
# The <shape> part is the tag bench.py's roofline objects look up (shape_list's last column).
set -u
TAG=$1
cd "$(dirname "$0")/.."
S() { # prof-dir-suffix shape-name kernel-match streams fft
  [ -d gpurun_out/prof/${TAG}_$1 ] || { echo "(no gpurun_out/prof/${TAG}_$1)"; return; }
  python tools/summarize_profile.py ${TAG}_$1 ${TAG}_$2 "$3" $4 $5 "$(head -c 300 gpurun_out/prof/${TAG}_$1/cmd.txt 2>/dev/null)" > /dev/null && echo "profiles/${TAG}_$2_pmc.json"
}
S cfg3 cfg3_n4096 spectrum_tick 4096 4096
S cfg3_8192streams cfg3_8192streams spectrum_tick 8192 4096
S cfg3_16384streams cfg3_16384streams spectrum_tick 16384 4096
S cfg4 cfg4_n16384_bars spectrum_tick 1024 16384
S cfg2 cfg2_batch spectrum_tick 256 2048
S cfg5shape cfg5shape_8192streams_barsonly spectrum_tick 8192 4096
S plugindefaults plugindefaults spectrum_tick 4096 4096
S n32768 n32768 spectrum_tick 512 32768
S n800 n800_mixed_radix spectrum_tick 8192 800
S n4160 n4160_mixed_radix spectrum_tick 2048 4160
S n65536 n65536 big_ 256 65536
S n16400 n16400 big_ 512 16400
S n48000 n48000 big_ 256 48000
S n48016 n48016_bluestein_rows big_ 256 48016
S n32000 n32000_mixed_radix_whole big_ 256 32000
S meter meter meter_tick 16384 7200
S wave wave waveform_tick 65536 800
