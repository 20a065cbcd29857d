"""A/B timing of the LS launch of adaptation 1 with two builds of the library on the SAME device state
(tracks of a real adaptation 0):   python tools/ls_ab_probe.py <alternative .so> [workload]
The alternative build may be a timing experiment that produces wrong numbers (nothing downstream is run)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from eaqhm_amd import hip
from eaqhm_amd.engine import DeviceAnalysis, FramePlan

alt, wl = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "synth16k_60s")
fs, s, grid, frames, fstep = bench.load_workload(wl)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, 5)
it = eng.adaptations()
next(it); next(it)          # adaptation 0 complete, adaptation 1 enqueued (frame_prep done for a = 1)
torch.cuda.synchronize()


def time_ls(ctx, reps=5):
    p = eng.plan
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    out = []
    for a in (0, 1):
        for r in range(reps + 1):
            if r == 1:
                ev[0].record()
            ctx.ls_batch(a, eng.s, p.L, p.fs, eng.am_cur, eng.fm_cur, p.Kmax, eng.frame_inst, eng.frame_c, eng.frame_wl,
                         eng.frame_f0, eng.frame_K, eng.ncol, eng.cols, eng.seeded, eng.any_seed, eng.nf, p.wl_max, a,
                         p.f0_stale, eng.f0min, eng.records[0], None, None)
        ev[1].record()
        torch.cuda.synchronize()
        out.append(ev[0].elapsed_time(ev[1]) / reps)
    return out


base = time_ls(eng.ctx)
hip._lib = None
hip.LIB_PATH = alt
ctx2 = hip.Context(0)
other = time_ls(ctx2)
print("workload %s, %d frames: default build  a=0 %.2f ms  a>=1 %.2f ms | %s  a=0 %.2f ms  a>=1 %.2f ms"
      % (wl, eng.nf, base[0], base[1], os.path.basename(alt), other[0], other[1]))
