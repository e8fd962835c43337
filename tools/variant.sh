#!/bin/bash
# tools/variant.sh NAME "EXTRA FLAGS" -- a development build of the library into variants/lib_NAME.so (travels with gpurun,
# git-ignored); select it with WF_HIP_LIB=variants/lib_NAME.so.  Built with -DWF_DEV_BUILD: the WF_HIP_* environment overrides of the plan
# (WF_HIP_LANES, WF_HIP_SPLIT, WF_HIP_RING_PAD, WF_HIP_TLDS ...) exist in these builds only.  e.g. tools/variant.sh t4096 "-DWF_GEOM_ONLY=4096 -DWF_PHASE_TIMING"
set -e
NAME=$1; shift
cd "$(dirname "$0")/../waveform_amd/csrc"
mkdir -p ../../variants
make -s -j8 BUILD=../../build/variant_$NAME OUT=../../variants/lib_$NAME.so EXTRA="-DWF_DEV_BUILD $*" ../../variants/lib_$NAME.so
