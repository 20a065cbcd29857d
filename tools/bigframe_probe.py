"""Diagnostic: LS-stage time on large frames (48 kHz synthetic excerpt = BASELINE config 5 in miniature)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eaqhm_amd import prologue
from eaqhm_amd.engine import DeviceAnalysis, FramePlan
from eaqhm_amd.synth import synth_speech_int16
from eaqhm_amd.swipe import swipep
import bench
fs = 48000
s = synth_speech_int16(2.0, fs) / 32768.0
track = swipep(s, fs, [160, 300])
grid = prologue.resample_track(track, np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
prologue.apply_full_waveform(frames, len(s), 480)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, 1)
eng.profile = True
eng.run()
t = eng.stage_times_ms()
N = 2 * plan.frame_wl.astype(np.int64) + 1
fl = bench.ls_flops(N, 2 * plan.frame_K.astype(np.int64) + 1).sum()
print("frames", plan.n_frames, "Kmax", plan.Kmax, "N", N.min(), N.max(), "Kc", (2 * plan.frame_K + 1).min(), (2 * plan.frame_K + 1).max())
print("LS ms", t["ls"], "post ms", t["post"], "SRER", eng.SRER)
print("a=0: %.0f frames/s, %.2f TFLOP/s algorithmic" % (plan.n_frames / t["ls"][0] * 1e3, fl / t["ls"][0] / 1e9))
