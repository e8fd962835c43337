"""GPU parity: libwaveform_hip.so (through its C ABI) against the reference itself
(oracle/_ref/libwfref.so = phandasm/waveform's own TUs + vendored FFTW, generic path),
same audio pushed into both, tick by tick.

Tolerance (BASELINE.json north_star): 1e-5 relative on the float32 dB outputs
("outputs match source_generic.cpp ... within 1e-5 relative float tolerance"); an
absolute floor of 1e-4 dB covers bins whose dB value is close to 0.
"""
import numpy as np
import pytest

import waveform_amd as wf
from oracle import wfref
from tools import synth

pytestmark = pytest.mark.gpu

from helpers import ref_settings, assert_db_close


def _run_pair(cfg, streams, ticks, hop, channels=2, seconds=1 / 60):
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    refs = [wfref.RefSource(ref_settings(cfg), channels=channels) for _ in range(streams)]
    worst = 0.0
    with wf.SpectrumBatch(cfg, streams) as b:
        assert b.capture_channels == refs[0].capture_channels
        for t in range(ticks):
            audio = synth.block(synth.DEFAULT_SEED, 0, streams, b.capture_channels, t * hop, hop)
            b.push_audio(audio)
            b.tick(seconds=seconds)
            got = b.decibels()
            for s, r in enumerate(refs):
                r.feed_and_tick(audio[s, 0], audio[s, 1] if b.capture_channels > 1 else None, seconds=seconds)
                nch = 2 if cfg.stereo else 1
                for c in range(nch):
                    want = r.decibels(c)
                    worst = max(worst, assert_db_close(got[s, c], want, f"N={cfg.fft_size} tick {t} stream {s} ch {c}", deep=True))
    return worst


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384])
def test_stereo_ema_slope(n):
    cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=1.0)
    _run_pair(cfg, streams=3, ticks=8, hop=800)


@pytest.mark.parametrize("n", [2048, 4096])
def test_mono_mixdown_tv_fast_peaks(n):
    cfg = wf.Config.defaults(fft_size=n, stereo=0, tsmoothing=wf.TSMOOTH["tvexponential"], fast_peaks=1, slope=0.5)
    _run_pair(cfg, streams=2, ticks=8, hop=800)


def test_no_smoothing_hann_2048():
    cfg = wf.Config.defaults(fft_size=2048, stereo=1, tsmoothing=wf.TSMOOTH["none"])
    _run_pair(cfg, streams=2, ticks=4, hop=800)
