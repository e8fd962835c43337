"""numpy restatement of include/wf_synth.h (counter-hash white noise, bit-identical)."""
from __future__ import annotations

import numpy as np

DEFAULT_SEED = 0x5741564546524D31
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def key(seed: int, stream: int, channel: int) -> np.uint64:
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + _GOLD * np.uint64((int(stream) << 1) + int(channel) + 1)
        return _mix64(np.array([s], dtype=np.uint64))[0]


def noise(seed: int, stream: int, channel: int, index0: int, count: int) -> np.ndarray:
    """float32[count]: wf_synth_noise(seed, stream, channel, index0 + i)"""
    with np.errstate(over="ignore"):
        k = key(seed, stream, channel)
        idx = np.arange(index0 + 1, index0 + 1 + count, dtype=np.uint64)
        z = _mix64(k + _GOLD * idx)
    kbits = (z >> np.uint64(40)).astype(np.uint32)
    return (kbits.astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)).astype(np.float32)


def block(seed: int, stream0: int, n_streams: int, channels: int, index0: int, count: int) -> np.ndarray:
    """float32[n_streams, channels, count]"""
    out = np.empty((n_streams, channels, count), np.float32)
    for s in range(n_streams):
        for c in range(channels):
            out[s, c] = noise(seed, stream0 + s, c, index0, count)
    return out
