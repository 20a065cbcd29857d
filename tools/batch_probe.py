"""Diagnostic: B copies of the SA19 workload one after another vs interleaved on B streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from eaqhm_amd.engine import DeviceAnalysis, FramePlan, run_interleaved
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fs, s, grid, frames, fstep = bench.load_workload("sa19")
engs = []
for _ in range(B):
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    engs.append(DeviceAnalysis(s, s, plan, 160, 5))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def sequential():
    for e in engs:
        e.reset(); e.run()


def interleaved():
    for e in engs:
        e.reset()
    run_interleaved(engs)


for name, fn in (("sequential", sequential), ("interleaved", interleaved)):
    fn()
    dt = min(timed(fn) for _ in range(3))
    nfr = sum(e.n_ls_frames for e in engs)
    print("%-12s %d files  %.2f ms  %.0f frames/s   SRER[-2] %s" % (name, B, dt * 1e3, nfr / dt, engs[-1].SRER[-2]))
