"""Host-side prologue of the driver: everything eaQHMAnalysisAndSynthesis does once per file
before the adaptation loop (functions.py:86-161).  Runs on the CPU (NumPy/SciPy): it is < 1 % of
the reference's run time and is outside the accelerated hot path.
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view
from scipy.io import wavfile
from scipy.signal import ellip, filtfilt

from .structs import Frame

NORMALIZE = 32768  # misc.py:13


def read_signal(speech_file, fc=0):
    """functions.py:86-91: 16-bit mono WAV scaled by 1/32768, optional zero-phase elliptic high-pass."""
    fs, x = wavfile.read(speech_file)
    if np.ndim(x) != 1:
        raise ValueError("expected a mono .wav file (the reference assumes one channel)")
    s = np.asarray(x, dtype=np.float64) / NORMALIZE
    if fc > 0:
        s = elliptic(s, fs, fc, "highpass")
    return int(fs), s


def pitch_limits(gender):
    """functions.py:95-109 (the docstring there is wrong about 'female'; the code says 160-300 Hz)."""
    if isinstance(gender, tuple):
        return gender[0], gender[1]
    table = {"male": (70, 180), "female": (160, 300), "child": (300, 600)}
    return table.get(gender, (70, 500))


def elliptic(s, fs, fc, kind):
    """misc.py:167-182: 6th-order elliptic, 0.5 dB ripple, 60 dB stop band, filtfilt."""
    b, a = ellip(6, 0.5, 60, 2 * fc / fs, kind)
    return filtfilt(b, a, s)


def resample_track(track, times):
    """functions.py:644-680 (getLinear) for an ascending vector of query times: every column of
    `track` after the first is interpolated linearly; column 0 of the result is the query time.
    Queries before the first track time take the first row; a query past the last time is an error
    in the reference too."""
    track = np.asarray(track, dtype=np.float64)
    times = np.asarray(times, dtype=np.float64)
    tt = track[:, 0]
    lo = np.searchsorted(tt, times, side="right") - 1
    out = np.empty((len(times), track.shape[1]))
    out[:, 0] = times
    before = lo < 0
    lo_c = np.clip(lo, 0, len(tt) - 1)
    exact = tt[lo_c] == times
    hi_c = np.clip(lo_c + 1, 0, len(tt) - 1)
    if np.any(~before & ~exact & (lo_c == len(tt) - 1)):
        raise IndexError("pitch track is shorter than the signal")
    den = np.where(hi_c > lo_c, tt[hi_c] - tt[lo_c], 1.0)
    g = (times - tt[lo_c]) / den
    vals = track[lo_c, 1:] * (1 - g)[:, None] + track[hi_c, 1:] * g[:, None]
    vals[exact] = track[lo_c[exact], 1:]
    vals[before] = track[0, 1:]
    out[:, 1:] = vals
    return out


def _median_quirk(x, p=5):
    """misc.py:184-206 as written: (p-1)-tap 'median' whose output runs backwards in time (the
    Toeplitz rows are built from the flipped, edge-padded input).  Output values are in {0, .5, 1}."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    pad = (p - 1) // 2
    xp = np.concatenate((np.full(pad, x[0]), x, np.full(pad, x[-1])))
    row = np.arange(n)[:, None]
    col = np.arange(p - 1)[None, :]
    pick = np.where(col <= row, n - 1 - row + col, n + col - row)
    srt = np.sort(xp[pick], axis=1)
    mid = (p - 1) // 2
    return 0.5 * (srt[:, mid - 1] + srt[:, mid])


def voiced_unvoiced_frames(s, fs, gender):
    """functions.py:577-642: 30 ms energy windows every 5 ms on the 30 Hz high-passed signal and on
    its 1 / 1.5 kHz low-passed copy; thresholds -60 / 10 / -50 dB; smoothed flags.
    Returns (list[Frame], frame_step)."""
    s = elliptic(np.asarray(s, dtype=np.float64), fs, 30, "highpass")
    n = len(s)
    smooth = elliptic(s, fs, 1000 if gender == "male" else 1500, "lowpass")
    wlen = int(round(0.03 * fs))
    wlen += (wlen % 2 == 0)
    hop = int(round(0.005 * fs))
    half = (wlen - 1) // 2
    ti = np.arange(1, n, hop)
    speech = np.zeros(len(ti), dtype=bool)
    voiced = np.zeros(len(ti), dtype=bool)
    ok = (ti > half) & (ti < n - half)
    if ok.any():
        starts = ti[ok] - half - 1                     # window = s[t-half-1 : t+half]
        with np.errstate(divide="ignore"):
            e = 20 * np.log10(sliding_window_view(s, wlen)[starts].std(axis=1))
            es = 20 * np.log10(sliding_window_view(smooth, wlen)[starts].std(axis=1))
        sp = e > -60
        speech[ok] = sp
        voiced[ok] = sp & (e - es < 10) & (es > -50)
    speech_f = _median_quirk(speech, 5)
    voiced_f = _median_quirk(voiced, 5)
    frames = [Frame(int(t), a, b) for t, a, b in zip(ti, speech_f, voiced_f)]
    return frames, int(frames[1].ti - frames[0].ti)


def apply_full_waveform(frames, length, analysis_window_samples):
    """functions.py:139-146: away from the edges every frame becomes (speech and) voiced."""
    half = analysis_window_samples / 2
    for f in frames:
        if half < f.ti < length - half:
            if f.isSpeech and not f.isVoiced:
                f.isVoiced = True
            if (not f.isSpeech) and (not f.isVoiced):
                f.isSpeech = True
                f.isVoiced = True


def voiced_only_target(s, frames, frame_step):
    """functions.py:127-138: SRER target for fullWaveform=False."""
    out = np.zeros_like(s)
    run = []
    for f in frames:
        if f.isSpeech and f.isVoiced:
            run.append(f.ti)
        elif run:
            lo, hi = run[0] - frame_step, run[-1] + frame_step + 1
            out[lo:hi] = s[lo:hi]
            run = []
    return out
