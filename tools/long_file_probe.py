"""Long files on one MI355X: the 60 s synthetic workload tiled to `minutes` minutes, analysed with
the dense tracks streamed in time blocks under a byte budget (engine.DeviceAnalysis(track_budget_bytes=...)).
    python tools/long_file_probe.py <synth16k_60s|synth48k_60s> <minutes> <track budget MB | auto> [--also-resident] [--max-adpt N]
Prints one JSON line: frames/s of the adaptation loops, SRER list, time blocks, peak device memory of the engine's
buffers, and — with --also-resident — whether the resident run gives bit for bit the same SRER list and records."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from eaqhm_amd import prologue
from eaqhm_amd.engine import DeviceAnalysis, FramePlan

args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl, minutes = args[0], int(args[1])
budget = "auto" if args[2] == "auto" else int(float(args[2]) * 2 ** 20)
max_adpt = int(sys.argv[sys.argv.index("--max-adpt") + 1]) if "--max-adpt" in sys.argv else 1
fs, s1, track = bench.load_signal(wl)
grid1 = prologue.resample_track(track, np.arange(0, len(s1) - 1, round(fs * 5 / 1000)) / fs)
s = np.tile(s1, minutes)
# 5 ms pitch grid of the tiled signal: the one-minute grid repeated (times shifted); the last frame of a minute is held
n5 = len(np.arange(0, len(s) - 1, round(fs * 5 / 1000)))
per_min = len(s1) // round(fs * 5 / 1000)
f0 = np.concatenate([np.resize(grid1[:, 1], per_min) for _ in range(minutes)])
f0 = np.concatenate((f0, np.full(max(0, n5 - len(f0)), f0[-1])))[:n5]
grid = np.column_stack((np.arange(n5) * 0.005, f0))
t0 = time.time()
frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
prologue.apply_full_waveform(frames, len(s), 32 * 15)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
host_s = time.time() - t0
out = {"workload": wl, "minutes": minutes, "samples": int(plan.L), "fs": int(fs), "Kmax": int(plan.Kmax),
       "ls_frames_per_adaptation": int(plan.n_frames), "host_vuv_and_plan_s": host_s, "max_adpt": max_adpt}


def run(budget):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    eng = DeviceAnalysis(s, s, plan, 160, max_adpt, track_budget_bytes=budget)
    torch.cuda.synchronize()
    t0 = time.time()
    eng.run()
    torch.cuda.synchronize()
    dt = time.time() - t0
    res = {"seconds": dt, "frames_per_sec": eng.n_ls_frames / dt, "srer_db": [float(v) for v in eng.SRER],
           "time_blocks": len(eng.blocks), "track_bytes": int(eng.track_bytes()),
           "peak_engine_bytes": int(torch.cuda.max_memory_allocated() - base),
           "records_checksum": float(eng.records[1].sum()), "s_hat_checksum": float(eng.s_hat[1].sum())}
    del eng
    return res


out["streaming"] = run(budget)
out["streaming"]["budget_bytes"] = budget
out["dense_tracks_if_resident_bytes"] = int(2 * 8 * plan.Kmax * plan.L)
out["reference_seven_arrays_bytes"] = int(7 * 8 * plan.Kmax * plan.L)
if "--also-resident" in sys.argv:
    out["resident"] = run(None)
    out["bitwise_equal"] = (out["resident"]["srer_db"] == out["streaming"]["srer_db"] and
                            out["resident"]["records_checksum"] == out["streaming"]["records_checksum"] and
                            out["resident"]["s_hat_checksum"] == out["streaming"]["s_hat_checksum"])
print(json.dumps(out))
