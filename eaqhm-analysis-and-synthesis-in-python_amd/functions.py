"""Drop-in replacement for the reference's functions.py entry points, backed by libeaqhm_hip.so.

    eaQHMAnalysisAndSynthesis(speechFile, gender='other', step=15, maxAdpt=10, pitchPeriods=3,
                              analysisWindow=32, fullWaveform=True, fc=0, partials=0,
                              printPrompts=True, loadingScreen=True)
        -> (s_recon, SRER, DetComponents, endTime)                      [functions.py:35-418]
    iqhmLS_complexamps(s, f0range, window, fs)   -> (amplitudes, slopes) [functions.py:420-470]
    eaqhmLS_complexamps(s, am, fm, window, fs)   -> (amplitudes, slopes) [functions.py:472-535]
    phase_integr_interpolation(fm_recon, ph_recon, indices) -> pm_final  [functions.py:537-575]

Same names, argument meaning, defaults, return shapes and error behaviour (Python exceptions).
There is no CPU path: without the HIP library or a GPU every call raises `HipUnavailable`.
"""
from time import gmtime, strftime, time

import numpy as np

from . import prologue
from .engine import DeviceAnalysis, FramePlan
from .hip import Context
from .structs import Deterministic

_seam_ctx = {}


def _ctx(device_index=0):
    c = _seam_ctx.get(device_index)
    if c is None:
        c = _seam_ctx[device_index] = Context(device_index)
    c.bind_stream()
    return c


def _slot_arrays(active, *fields):
    """What misc.py:65-93 (arrayByIndex) returns when it is fed the (n,1)-shaped index and value arrays of
    functions.py:407-411, for every row of `active` and every array in `fields` at once: per row an object array of
    length (highest active slot + 1) whose active entries are shape-(1,) float64 arrays and whose other entries are
    the int 0.  The cells are views of one (cells, 1) block per field, scattered into one flat object
    array that is then cut per row: no Python-level loop over the ~2 M cells per field of a minute of speech.  The
    cyclic garbage collector is paused meanwhile — millions of fresh container objects otherwise trigger full
    collections whose cost grows with their number (measured: 5.3 s instead of 1.3 s for 6 M cells).

    active: (rows, Kmax) bool;  fields: (rows, Kmax) float64 each.  Returns one list of `rows` object arrays per field
    (a row without any active slot gets an empty one: the reference raises IndexError there, end() of an empty array)."""
    import gc
    rows, K = active.shape
    lens = np.where(active.any(axis=1), K - np.argmax(active[:, ::-1], axis=1), 0)
    offs = np.concatenate(([0], np.cumsum(lens)))
    r, k = np.nonzero(active)
    pos = offs[r] + k
    cuts = [(int(offs[i]), int(offs[i + 1])) for i in range(rows)]
    out = []
    was_on = gc.isenabled()
    gc.disable()
    try:
        for values in fields:
            flat = np.zeros(int(offs[-1]), dtype=object)         # int 0 everywhere, like numpy.zeros(dtype=object)
            if len(r):
                block = np.ascontiguousarray(values[r, k], dtype=np.float64).reshape(-1, 1)
                flat[pos] = np.fromiter(iter(block), dtype=object, count=len(block))
            out.append([flat[lo:hi] for lo, hi in cuts])
    finally:
        if was_on:
            gc.enable()
    return out


def _prepare(speechFile, gender, step, maxAdpt, pitchPeriods, analysisWindow, fullWaveform, fc, partials,
             pitch_track, device_index, track_budget_bytes="auto"):
    """functions.py:86-146: everything before the adaptation loop -> (plan, engine)."""
    fs, s = prologue.read_signal(speechFile, fc)                                 # functions.py:86-91
    length = len(s)
    f0min, f0max = prologue.pitch_limits(gender)                                 # functions.py:95-109
    if pitch_track is None:
        from .swipe import swipep
        pitch_track = swipep(s, fs, [f0min, f0max])                              # functions.py:111
    grid_t = np.arange(0, length - 1, round(fs * 5 / 1000)) / fs
    f0_grid = prologue.resample_track(pitch_track, grid_t)                       # functions.py:113
    frames, frame_step = prologue.voiced_unvoiced_frames(s, fs, gender)          # functions.py:125
    if fullWaveform:
        prologue.apply_full_waveform(frames, length, analysisWindow * step)      # functions.py:139-146
        target = s
    else:
        target = prologue.voiced_only_target(s, frames, frame_step)              # functions.py:127-138
    plan = FramePlan(length, fs, f0_grid, frames, frame_step, step, pitchPeriods, analysisWindow, partials)
    return plan, DeviceAnalysis(s, target, plan, f0min, maxAdpt, device_index=device_index,
                                track_budget_bytes=track_budget_bytes)


def eaQHMAnalysisAndSynthesis(speechFile: str, gender: str or tuple = 'other', step: int = 15,
                              maxAdpt: int = 10, pitchPeriods: int = 3, analysisWindow: int = 32,
                              fullWaveform: bool = True, fc: int = 0, partials: int = 0,
                              printPrompts: bool = True, loadingScreen: bool = True, *,
                              pitch_track=None, device_index: int = 0, track_budget_bytes="auto",
                              det_format: str = "structs", _return_engine: bool = False):
    """Adaptive quasi-harmonic analysis/resynthesis of a mono 16-bit .wav on an MI355X.

    Parameters and returns: exactly those of the reference (functions.py:38-82).  `loadingScreen` is
    accepted and ignored (there is no per-frame Python loop to show progress for).

    Keyword-only extensions (not in the reference):
      pitch_track   (T, >=2) array [time s, f0 Hz, ...] used instead of running SWIPE' — either the
                    1 ms track swipep() returns or an already resampled 5 ms grid
      device_index  which GPU of this process to use
      track_budget_bytes  long files: bytes the dense am/fm tracks (and their zero counts) may occupy on the device.
                    None keeps them resident for the whole file (the reference keeps seven (L, Kmax) arrays,
                    functions.py:159-171); with a budget the frames are worked off in time blocks whose tracks are
                    regenerated from the frame-centre records — same results, bit for bit.  "auto" (default): resident
                    while they take less than 40 % of the free device memory, else streamed (engine.auto_track_budget)
      det_format    "structs" (default): DetComponents is the reference's list of Deterministic objects
                    (functions.py:404-411).  "arrays": a dict of NumPy arrays instead — ti, isSpeech, isVoiced (per
                    instant), a0 (No_ti,), amplitudes / frange / pk (No_ti, Kmax; zero where a slot is inactive) — for
                    callers that do not want ~100 Python objects per analysis instant (building them takes four times
                    as long as the whole analysis of a minute of speech)
    """
    if det_format not in ("structs", "arrays"):
        raise ValueError("det_format must be 'structs' or 'arrays'")
    start = time()
    plan, eng = _prepare(speechFile, gender, step, maxAdpt, pitchPeriods, analysisWindow, fullWaveform, fc,
                         partials, pitch_track, device_index, track_budget_bytes)

    state = {"t": time()}

    def report(a, e):
        if printPrompts:                                                         # functions.py:390-392
            print('---- Adaptation No. {} ----\n'.format(a))
            print('\nSRER: {} dB in Adaptation No: {}'.format(e.SRER[a], a))
            print('Adaptation Time: {}\n'.format(strftime("%H:%M:%S", gmtime(time() - state["t"]))))
        state["t"] = time()

    eng.run(on_adaptation=report)
    fin = eng.final_arrays()
    det = pack_results(plan, fin) if det_format == "structs" else pack_arrays(plan, fin)
    end_time = time() - start
    if printPrompts:                                                             # functions.py:414-416
        print('Signal adapted to {} dB SRER'.format(round(max(eng.SRER), 6)))
        print('Total Time: {}\n\n'.format(strftime("%H:%M:%S", gmtime(end_time))))
    out = (fin["s_recon"], list(eng.SRER), det, end_time)
    return out + (eng,) if _return_engine else out


def eaQHMAnalysisAndSynthesisBatch(speechFiles, gender: str or tuple = 'other', step: int = 15,
                                   maxAdpt: int = 10, pitchPeriods: int = 3, analysisWindow: int = 32,
                                   fullWaveform: bool = True, fc: int = 0, partials: int = 0,
                                   printPrompts: bool = False, *, pitch_tracks=None, device_index: int = 0):
    """Several files in one go on one GPU (not in the reference).  Every file gets exactly the
    result eaQHMAnalysisAndSynthesis() gives it — the analyses are independent, each keeps its own stop rule — but
    their adaptation loops are interleaved on separate HIP streams (engine.run_interleaved), which keeps the GPU
    busy across the launch tails and host round trips of short files.  `gender` (and `pitch_tracks`) may be one
    value for all files or a list with one entry per file.  Returns a list of (s_recon, SRER, DetComponents,
    endTime) tuples in input order; endTime is the batch's wall time."""
    from .engine import run_interleaved
    start = time()
    files = list(speechFiles)
    genders = list(gender) if isinstance(gender, list) else [gender] * len(files)
    tracks = list(pitch_tracks) if pitch_tracks is not None else [None] * len(files)
    if len(genders) != len(files) or len(tracks) != len(files):
        raise ValueError("gender / pitch_tracks lists must have one entry per file")
    jobs = [_prepare(f, g, step, maxAdpt, pitchPeriods, analysisWindow, fullWaveform, fc, partials, t, device_index)
            for f, g, t in zip(files, genders, tracks)]

    def report(i, a, e):
        if printPrompts:
            print('[{}] SRER: {} dB in Adaptation No: {}'.format(files[i], e.SRER[a], a))

    run_interleaved([e for _, e in jobs], on_adaptation=report)
    out = []
    for plan, eng in jobs:
        fin = eng.final_arrays()
        out.append((fin["s_recon"], list(eng.SRER), pack_results(plan, fin)))
    end_time = time() - start
    return [o + (end_time,) for o in out]


def pack_results(plan, fin):
    """functions.py:325-329 + :404-411: one Deterministic per analysis instant."""
    centres = list((plan.ti - 1).astype(np.int64))                  # numpy.int64 scalars, like ti[i] - 1
    analysed = np.flatnonzero(plan.analysed)
    am = fin["am"][analysed]
    active = am != 0
    amps, freqs, phases = _slot_arrays(active, am, fin["fm"][analysed], fin["pk"][analysed])
    a0 = list(np.asarray(fin["a0"], dtype=np.float64)[analysed])
    in_bounds = plan.in_bounds
    det = [Deterministic(ti=centres[i], isSpeech=bool(in_bounds[i]), isVoiced=False) for i in range(plan.No_ti)]
    for j, i in enumerate(analysed):
        d = det[i]
        d.isVoiced = True
        d.a0 = a0[j]
        d.amplitudes = amps[j]
        d.frange = freqs[j]
        d.pk = phases[j]
    return det


def pack_arrays(plan, fin):
    """The content of functions.py:404-411 as plain arrays (det_format="arrays")."""
    voiced = np.asarray(plan.analysed, dtype=bool)
    out = dict(ti=(plan.ti - 1).astype(np.int64), isSpeech=np.asarray(plan.in_bounds, dtype=bool), isVoiced=voiced,
               a0=np.where(voiced, fin["a0"], 0.0))
    for name, key in (("amplitudes", "am"), ("frange", "fm"), ("pk", "pk")):
        arr = np.array(fin[key], dtype=np.float64)
        arr[fin["am"] == 0] = 0.0
        arr[~voiced] = 0.0
        out[name] = arr
    return out


def _column(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))


def _ls_explicit(s, am, fm, f0range, window, fs):
    import torch
    c = _ctx()
    dev = c.device
    s = _column(s)
    N = len(s)
    window = _column(window)
    if len(window) != N:
        raise ValueError("window and signal lengths differ")

    def up(a):
        return None if a is None else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)

    if fm is not None:
        am, fm = np.asarray(am, dtype=np.float64), np.asarray(fm, dtype=np.float64)
        if fm.shape != am.shape or fm.shape[0] != N:
            raise ValueError("am/fm must both be (len(s), K)")
        Kc = fm.shape[1]
    else:
        f0range = _column(f0range)
        Kc = len(f0range)
    out_a = torch.zeros(2 * Kc, dtype=torch.float64, device=dev)
    out_b = torch.zeros(2 * Kc, dtype=torch.float64, device=dev)
    c.ls_explicit(up(s), N, up(am), up(fm), up(f0range), Kc, up(window), fs, out_a, out_b)
    if c.ls_faults()[0]:                # functions.py:465 / :530: inv() of a singular normal matrix
        raise np.linalg.LinAlgError("Singular matrix")
    a = out_a.cpu().numpy().view(np.complex128).reshape(Kc, 1)
    b = out_b.cpu().numpy().view(np.complex128).reshape(Kc, 1)
    return a, b


def iqhmLS_complexamps(s, f0range, window, fs: int):
    """functions.py:420-470 on the GPU: returns (amplitudes, slopes), each (K, 1) complex128."""
    return _ls_explicit(s, None, None, f0range, window, fs)


def eaqhmLS_complexamps(s, am, fm, window, fs):
    """functions.py:472-535 on the GPU: returns (amplitudes, slopes), each (K, 1) complex128."""
    return _ls_explicit(s, am, fm, None, window, fs)


def phase_integr_interpolation(fm_recon, ph_recon, indices):
    """functions.py:537-575 on the GPU: phase interpolation by integration of the instantaneous frequency
    (`fm_recon` already in rad/sample, as the driver passes it) between the knots `indices`; returns the dense
    phase on indices[0]..indices[-1]."""
    import torch
    c = _ctx()
    om = _column(fm_recon)
    ph = _column(ph_recon)
    kn = np.ascontiguousarray(np.asarray(indices).reshape(-1), dtype=np.int32)
    if len(om) != len(ph):
        raise ValueError("fm_recon and ph_recon lengths differ")
    if len(kn) < 2 or np.any(np.diff(kn) <= 0) or kn[0] < 0 or kn[-1] >= len(om):
        raise ValueError("indices must be at least two ascending sample positions inside the arrays")
    dev = c.device
    out = torch.zeros(int(kn[-1] - kn[0] + 1), dtype=torch.float64, device=dev)
    c.phase_integrate(torch.as_tensor(om, device=dev), torch.as_tensor(ph, device=dev), torch.as_tensor(kn, device=dev),
                      len(kn), int(kn[0]), int(kn[-1]), out)
    return out.cpu().numpy()
