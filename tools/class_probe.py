"""Time of the adaptation>=1 LS launch per size class (tile rows of the stacked system) on real tracks:
    python tools/class_probe.py [workload]
Runs adaptations 0 and 1 of the workload, then launches eaqhm_ls_batch on the frames of one class at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from eaqhm_amd.engine import DeviceAnalysis, FramePlan, ls_cost

wl = sys.argv[1] if len(sys.argv) > 1 else "synth16k_60s"
fs, s, grid, frames, fstep = bench.load_workload(wl)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, 5)
it = eng.adaptations()
next(it); next(it)          # adaptation 0 complete, adaptation 1 enqueued (frame_prep done for a = 1)
torch.cuda.synchronize()
p, K = plan, plan.Kmax
ncol = eng.ncol.cpu().numpy()
nt = (2 * (2 * ncol + 1) + 1 + 15) // 16
cols = eng.cols.view(-1, K)
N = 2 * p.frame_wl.astype(np.int64) + 1


def time_ls(sel, reps=3):
    idx = torch.as_tensor(sel, device=eng.s.device)
    tabs = [t[idx].contiguous() for t in (eng.frame_inst, eng.frame_c, eng.frame_wl, eng.frame_f0, eng.frame_K, eng.ncol)]
    cc = cols[idx].contiguous().view(-1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps + 1):
        if r == 1:
            ev[0].record()
        eng.ctx.ls_batch(1, eng.s, p.L, p.fs, eng.am_cur, eng.fm_cur, eng.track_t0, eng.track_len, K, tabs[0], tabs[1],
                         tabs[2], tabs[3], tabs[4], tabs[5], cc, eng.seeded, eng.any_seed, len(sel), p.wl_max, 1,
                         p.f0_stale, eng.f0min, eng.records[0], None, None)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


tot = time_ls(np.arange(eng.nf))
print("workload %s: %d frames, whole launch %.2f ms" % (wl, eng.nf, tot))
acc = 0.0
for c in sorted(set(nt.tolist())):
    sel = np.flatnonzero(nt == c)
    if len(sel) < 256:
        continue
    ms = time_ls(sel)
    acc += ms
    fl = float(ls_cost(N[sel], 2 * ncol[sel] + 1).sum())
    cyc = ms * 1e-3 * 2.4e9 * eng.ctx.n_cu / len(sel)
    print("  %2d tile rows: %6d frames  %7.2f ms  %6.0f k cycles/frame/CU (at 2.4 GHz)  %5.1f TFLOP/s = %.3f of 78.6"
          % (c, len(sel), ms, cyc / 1e3, fl / ms / 1e9, fl / ms / 1e9 / 78.6))
print("sum of the class launches %.2f ms" % acc)
