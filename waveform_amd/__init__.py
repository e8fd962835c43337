"""waveform_amd -- MI355X (gfx950) implementation of phandasm/waveform's per-tick spectrum path.

The product is the C-ABI shared library ``libwaveform_hip.so`` (include/wf_hip.h); this
package is a thin ctypes binding used by the tests and by bench.py.  There is no CPU
implementation in here: if the library (or a gfx950 device) is missing, calls fail loudly.
"""
from .binding import (  # noqa: F401
    Config, SpectrumBatch, MultiBatch, PinnedBuffer, WfHipError, lib, library_path, device_count, db_min,
    WINDOW, TSMOOTH, INTERP, TICK_NO_DECIBELS,
)
