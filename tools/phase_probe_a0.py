"""Diagnostic: where the adaptation-0 launch (closed-form Gramian, two real systems; eaqhm_ls_a0.h) spends its cycles —
in-kernel s_memtime stamps of thread 0:   python tools/phase_probe_a0.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eaqhm_amd.engine import DeviceAnalysis, FramePlan
wl = sys.argv[1] if len(sys.argv) > 1 else "synth16k_60s"
fs, s, grid, frames, fstep = bench.load_workload(wl)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, 5)
it = eng.adaptations()
next(it)
torch.cuda.synchronize()
eng.ctx.set_option(2, 1)
p, launches = eng.plan, 3
for r in range(launches):
    eng.ctx.ls_batch(0, eng.s, p.L, p.fs, eng.am_cur, eng.fm_cur, eng.track_t0, eng.track_len, p.Kmax, eng.frame_inst, eng.frame_c,
                     eng.frame_wl, eng.frame_f0, eng.frame_K, eng.ncol, eng.cols, eng.seeded, eng.any_seed, eng.nf, p.wl_max, 0,
                     p.f0_stale, eng.f0min, eng.records[0], None, None)
torch.cuda.synchronize()
d = eng.ctx.debug_read()
names = {0: "window + signal into LDS", 1: "Toeplitz tables", 2: "tile numbering + system fill (a0_entry)", 10: "stage: own diagonal role / wait", 8: "stage: trailing update", 6: "stage: wait at barrier A",
         7: "stage: panel", 3: "system fill + factorisation rest", 4: "back substitution", 5: "record"}
tot = sum(d[k] for k in names)
for k, n in names.items():
    print("%-40s %16d cycles %5.1f%%" % (n, d[k], 100.0 * d[k] / max(tot, 1)))
print("workload", wl, "frames", eng.nf, "cycles/frame", tot / (launches * eng.nf))
