// wf_dev_guard.hpp -- the release library carries no laboratory: measurement cuts (WF_EXP_*: kernels that end early or skip a phase,
// with WRONG results), the per-phase clock stamps (WF_PHASE_TIMING) and the test aids (wf_hip_debug_*) exist in development builds
// only (-DWF_DEV_BUILD: tools/variant.sh, libwaveform_hip_dev.so).  A stray -D on a release build stops here instead of shipping.
// tests/test_cpu_units.py checks that every WF_EXP_* macro in csrc/ is listed below and defaults to 0.
#pragma once
#ifndef WF_DEV_BUILD
#if defined(WF_PHASE_TIMING) || defined(WF_EXP_CUT_AT)
#error "WF_PHASE_TIMING / WF_EXP_CUT_AT need -DWF_DEV_BUILD: they change what the kernels compute or store"
#endif
#if defined(WF_EXP_NO_TAIL) && WF_EXP_NO_TAIL != 0
#error "WF_EXP_NO_TAIL needs -DWF_DEV_BUILD: the display phase is skipped, the bars come out wrong"
#endif
#endif
