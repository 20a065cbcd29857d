"""Diagnostic: where eaqhm_ls_tile_kernel spends its cycles (in-kernel s_memtime stamps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from eaqhm_amd.engine import DeviceAnalysis, FramePlan
wl = sys.argv[1] if len(sys.argv) > 1 else "synth16k_60s"
fs, s, grid, frames, fstep = bench.load_workload(wl)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, 5)
eng.ctx.set_option(2, 1)
eng.run()
d = eng.ctx.debug_read()
names = ["setup+A1", "build: barrier wait", "contraction", "(fact tail)", "backsubst", "record", "publish diag", "trsm", "update", "-", "diag_coop", "build: items (thread 0)", "build: rows (thread 0)"]
tot = sum(d[:13])
for n, v in zip(names, d):
    print("%-20s %12d cycles  %5.1f%%" % (n, v, 100.0 * v / max(tot, 1)))
print("workload", wl, "frames", eng.n_ls_frames, "cycles/frame", tot / max(1, eng.n_ls_frames), "SRER", [float(v) for v in eng.SRER])
