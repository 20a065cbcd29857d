"""Parity point between the reference-generated fixtures and the full-size runs: the first `seconds` of a bench workload
analysed by the oracle (CPU restatement, pinned to the reference's fixtures) and by the HIP path, adaptations 0..maxAdpt:
    python tools/oracle_vs_hip.py <workload> <seconds> [maxAdpt] [threads]
The oracle is the checker here (tools/ and tests/ may use it; the product never does).  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
import eaqhm_oracle as O
from threadpoolctl import threadpool_limits
from eaqhm_amd import prologue
from eaqhm_amd.engine import DeviceAnalysis, FramePlan

wl, secs = sys.argv[1], float(sys.argv[2])
max_adpt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 16
fs, s, track = bench.load_signal(wl)
s = s[:int(secs * fs)]
gt = np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs
grid = prologue.resample_track(track, gt)
frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
prologue.apply_full_waveform(frames, len(s), 32 * 15)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, max_adpt)
t0 = time.time()
eng.run()
t_hip = time.time() - t0
fin = eng.final_arrays()
ti5, sp, vo, fstep_o = O.voiced_unvoiced_frames(s, fs, "female")
t0 = time.time()
with threadpool_limits(limits=threads):
    ref = O.analyse(s, fs, O.get_linear(track, gt), ti5, sp, vo, fstep_o, f0min=160, maxAdpt=max_adpt)
t_or = time.time() - t0
m = ref["am"] != 0
both = m & (fin["am"] != 0)
strong = both & (ref["am"] > 1e-6 * ref["am"].max())
out = {"workload": wl, "seconds": secs, "samples": len(s), "max_adpt": max_adpt, "ls_frames_per_adaptation": int(plan.n_frames),
       "Kmax": int(plan.Kmax), "srer_hip_db": [float(v) for v in eng.SRER], "srer_oracle_db": [float(v) for v in ref["SRER"]],
       "srer_abs_diff_db": [float(abs(a - b)) for a, b in zip(eng.SRER, ref["SRER"])],
       "s_recon_max_abs_diff": float(np.abs(fin["s_recon"] - ref["s_recon"]).max()),
       "acceptance_mask_agreement": float(np.mean((fin["am"] != 0) == m)),
       "am_max_rel_diff": float(np.abs(fin["am"][both] - ref["am"][both]).max() / ref["am"].max()),
       "fm_max_abs_diff_hz": float(np.abs(fin["fm"][both] - ref["fm"][both]).max()),
       "pk_max_abs_diff_rad_strong": float(np.abs(np.angle(np.exp(1j * (fin["pk"][strong] - ref["pk"][strong])))).max()),
       "hip_seconds": t_hip, "oracle_seconds": t_or, "oracle_threads": threads, "source_hash": bench.source_hash()}
print(json.dumps(out))
