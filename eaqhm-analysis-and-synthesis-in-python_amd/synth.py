"""Deterministic synthetic speech-like test signal.

Not part of the reference: the reference ships a single recording (SA19.WAV).  This
generator produces the "synthetic 16 kHz / 48 kHz speech of the named length" that
BASELINE.json's configs 4-5 ask for: a harmonic complex with a slowly moving f0 inside
the 'female' pitch limits of functions.py:101-103, slow per-partial amplitude
modulation, noise-only 100 ms margins (instants within analysisWindow*step samples of
either end are never analysed, functions.py:180) and a -45 dB noise floor so that no
analysed frame is digitally silent.
"""
import numpy as np

SEED = 20240608


def synth_speech(duration_s: float, fs: int) -> np.ndarray:
    """Float64 signal in [-0.25, 0.25]; `synth_speech_int16` quantises it like a 16-bit WAV."""
    n = int(round(duration_s * fs))
    rng = np.random.default_rng(SEED)
    t = np.arange(n) / fs
    f0 = 220.0 + 40.0 * np.sin(2 * np.pi * 0.31 * t) + 10.0 * np.sin(2 * np.pi * 1.7 * t)
    phi1 = 2 * np.pi * np.cumsum(f0) / fs
    kh = int((fs / 2 - 400) // 270.0)  # max f0 of the law above is 270 Hz
    theta = rng.uniform(0, 2 * np.pi, kh + 1)
    x = np.zeros(n)
    for k in range(1, kh + 1):
        x += k ** -1.2 * (1 + 0.3 * np.sin(2 * np.pi * (0.5 + 0.05 * k) * t + theta[k])) * np.cos(k * phi1 + theta[k])
    margin = int(round(0.1 * fs))
    ramp = int(round(0.02 * fs))
    gate = np.ones(n)
    gate[:margin] = 0.0
    gate[n - margin:] = 0.0
    r = 0.5 - 0.5 * np.cos(np.pi * (np.arange(ramp) + 0.5) / ramp)
    gate[margin:margin + ramp] = r
    gate[n - margin - ramp:n - margin] = r[::-1]
    steady = x[margin + ramp:n - margin - ramp]
    rms = np.sqrt(np.mean(steady ** 2)) if len(steady) else 1.0
    x = x * gate + rng.standard_normal(n) * rms * 10 ** (-45 / 20)
    return 0.25 * x / np.max(np.abs(x))


def synth_speech_int16(duration_s: float, fs: int) -> np.ndarray:
    return np.round(synth_speech(duration_s, fs) * 32767).astype(np.int16)
