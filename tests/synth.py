"""Seeded synthetic element PCM shared by oracle/gen_golden.py, the tests and bench.py.

Signals follow SURVEY.md §8(d): planar f32 in [-1, 1), 48 kHz.
"""
import numpy as np

FS = 48000


def gaussian(seed, channels, ns, sigma=0.15):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((channels, ns)).astype(np.float32) * np.float32(sigma)
    return np.clip(x, -1.0, np.float32(1.0 - 2.0**-24)).astype(np.float32)


def uniform(seed, channels, ns, amp=0.5):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-amp, amp, size=(channels, ns))).astype(np.float32)


def hot(seed, channels, ns, sigma=0.25, burst_amp=1.5, burst_len=240, burst_period=24000,
        burst_phase=3000):
    """'hot' programme: Gaussian noise plus a `burst_amp` burst of 5 ms every 0.5 s on every
    channel (alternating sign per sample so it survives any matrix), which drives the limiter
    through attack, hold and release.  Values are NOT clipped to [-1, 1): the limiter is the
    thing under test."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((channels, ns)).astype(np.float32) * np.float32(sigma)
    t = np.arange(ns)
    in_burst = ((t - burst_phase) % burst_period) < burst_len
    sign = np.where(t % 2 == 0, 1.0, -1.0).astype(np.float32)
    x[:, in_burst] += np.float32(burst_amp) * sign[in_burst]
    return x.astype(np.float32)


def quiet(seed, channels, ns, sigma=0.05):
    return gaussian(seed, channels, ns, sigma)
