"""Diagnostic: BASELINE configs 4 and 5 at (near) full size on one GPU — 60 s @16 kHz and 20 s @48 kHz of the
synthetic signal, pitch grids from the reference's SWIPE' (tests/golden/prep_fixtures.npz)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from eaqhm_amd import prologue
from eaqhm_amd.engine import DeviceAnalysis, FramePlan
from eaqhm_amd.synth import synth_speech_int16
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "prep_fixtures.npz"))
for name, fs, dur, key, max_adpt in (("config 4", 16000, 60.0, "synth16k_60s_f0s_5ms", 3), ("config 5 (20 s)", 48000, 20.0, "synth48k_20s_f0s_5ms", 1)):
    s = synth_speech_int16(dur, fs) / 32768.0
    grid = prologue.resample_track(g[key], np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    eng = DeviceAnalysis(s, s, plan, 160, max_adpt)
    eng.run()                       # warm-up (allocations)
    eng.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    N = 2 * plan.frame_wl.astype(np.int64) + 1
    fl = bench.ls_flops(N, 2 * plan.frame_K.astype(np.int64) + 1).sum() * len(eng.SRER)
    print("%s: L=%d frames/adaptation=%d Kmax=%d adaptations=%d  %.1f ms  %.0f frames/s  ~%.1f TFLOP/s (a=0 sizes)  SRER %s"
          % (name, plan.L, plan.n_frames, plan.Kmax, len(eng.SRER), dt * 1e3, eng.n_ls_frames / dt, fl / dt / 1e12,
             [round(float(v), 6) for v in eng.SRER]))
    del eng
    torch.cuda.empty_cache()
