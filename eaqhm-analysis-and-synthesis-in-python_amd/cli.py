"""Headless replacement for the reference's GUI harness main.py:44-72 (no Tk dialog, no plots):

    python eaqhm_amd.py <file.wav> [--gender female] [--max-adpt 10] ...

prints the per-adaptation SRER lines in the reference's format (functions.py:391-392, :415-416) and writes
`<name>_reconstructed.wav` as float32 next to the input (main.py:72)."""
import argparse

import numpy as np
from scipy.io import wavfile

from .functions import eaQHMAnalysisAndSynthesis


def main(argv=None):
    ap = argparse.ArgumentParser(prog="eaqhm_amd", description="eaQHM analysis/resynthesis on MI355X")
    ap.add_argument("wav")
    ap.add_argument("--gender", default="other", help="male | female | child | other | fmin,fmax")
    ap.add_argument("--step", type=int, default=15)
    ap.add_argument("--max-adpt", type=int, default=10)
    ap.add_argument("--pitch-periods", type=int, default=3)
    ap.add_argument("--analysis-window", type=int, default=32)
    ap.add_argument("--voiced-only", action="store_true", help="fullWaveform=False")
    ap.add_argument("--fc", type=int, default=0)
    ap.add_argument("--partials", type=int, default=0)
    ap.add_argument("--no-write", action="store_true")
    ap.add_argument("--track-budget-mb", type=float, default=0.0,
                    help="long files: device memory for the dense tracks (streamed in time blocks, same results); 0 = automatic: "
                         "resident while they fit comfortably, streamed otherwise")
    a = ap.parse_args(argv)
    gender = a.gender
    if "," in gender:
        lo, hi = gender.split(",")
        gender = (float(lo), float(hi))
    s_recon, srer, det, t = eaQHMAnalysisAndSynthesis(
        a.wav, gender, step=a.step, maxAdpt=a.max_adpt, pitchPeriods=a.pitch_periods,
        analysisWindow=a.analysis_window, fullWaveform=not a.voiced_only, fc=a.fc, partials=a.partials,
        printPrompts=True, loadingScreen=False,
        track_budget_bytes=int(a.track_budget_mb * 2 ** 20) if a.track_budget_mb > 0 else "auto")
    if not a.no_write:
        fs, _ = wavfile.read(a.wav)
        out = a.wav[:len(a.wav) - 4] + "_reconstructed.wav"
        wavfile.write(out, fs, np.float32(s_recon))
        print("wrote", out)
    return 0
