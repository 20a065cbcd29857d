"""Diagnostic: where eaqhm_ls_tile_kernel spends its cycles (in-kernel s_memtime stamps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from eaqhm_amd.engine import DeviceAnalysis, FramePlan
args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = args[0] if args else "synth16k_60s"
fs, s, grid, frames, fstep = bench.load_workload(wl)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, 5)
eng.ctx.set_option(2, 2 if "--diag" in sys.argv else 1)     # --diag: also time diag_D and the gaps (slows the kernel ~8 %)
eng.run()
d = eng.ctx.debug_read()
names = ["setup + slot preparation", "build: barrier wait", "contraction (a>=1) / closed-form fill (a=0)", "(end of factorisation)",
         "back substitution", "record", "wait at barrier A (the diagonal pipeline of another wave)", "panel (X = T W^H)",
         "trailing update (own tiles)", "-", "barrier C wait + next diagonal tile's update + diagonal role (when wave 0 has one)",
         "-", "build: basis rows (wave 0)"]
tot = sum(d[:9]) + d[10] + d[12]
for n, v in zip(names, d[:13]):
    print("%-90s %14d cycles  %5.1f%%" % (n, v, 100.0 * v / max(tot, 1)))
if d[13]:
    print("diag_D: %d tiles, %.0f cycles inside per tile, %.0f cycles from the end of one to the start of the next" % (d[13], d[9] / d[13], d[11] / max(d[14], 1)))
print("workload", wl, "frames", eng.n_ls_frames, "cycles/frame", tot / max(1, eng.n_ls_frames), "SRER", [float(v) for v in eng.SRER])
