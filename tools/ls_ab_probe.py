"""A/B timing of the LS launch of adaptation 1 with two builds of the library on the SAME device state
(tracks of a real adaptation 0):   python tools/ls_ab_probe.py <alternative .so> [workload]
The alternative build may be a timing experiment that produces wrong numbers (nothing downstream is run)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from eaqhm_amd import hip
from eaqhm_amd.engine import DeviceAnalysis, FramePlan

alts, wl = sys.argv[1].split(","), (sys.argv[2] if len(sys.argv) > 2 else "synth16k_60s")
fs, s, grid, frames, fstep = bench.load_workload(wl)
secs = float(os.environ.get("EAQHM_PROBE_SECONDS", "0"))      # optional: only the first seconds of the workload
if secs > 0:
    s = s[:int(secs * fs)]
    grid = grid[:len(np.arange(0, len(s) - 1, round(fs * 5 / 1000)))]
    from eaqhm_amd import prologue
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, 5)
it = eng.adaptations()
next(it); next(it)          # adaptation 0 complete, adaptation 1 enqueued (frame_prep done for a = 1)
torch.cuda.synchronize()


def time_ls(ctx, reps=(2 if fs > 16000 else 5)):
    p = eng.plan
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    out = []
    for a in (0, 1):
        for r in range(reps + 1):
            if r == 1:
                ev[0].record()
            ctx.ls_batch(a, eng.s, p.L, p.fs, eng.am_cur, eng.fm_cur, eng.track_t0, eng.track_len, p.Kmax, eng.frame_inst, eng.frame_c, eng.frame_wl,
                         eng.frame_f0, eng.frame_K, eng.ncol, eng.cols, eng.seeded, eng.any_seed, eng.nf, p.wl_max, a,
                         p.f0_stale, eng.f0min, eng.records[0], None, None)
        ev[1].record()
        torch.cuda.synchronize()
        out.append(ev[0].elapsed_time(ev[1]) / reps)
    return out


base = time_ls(eng.ctx)
print("workload %s, %d frames: default build  a=0 %.2f ms  a>=1 %.2f ms" % (wl, eng.nf, base[0], base[1]))
for alt in alts:
    hip._lib = None
    hip.LIB_PATH = alt
    ctx2 = hip.Context(0)
    other = time_ls(ctx2)
    print("   %-40s a=0 %.2f ms  a>=1 %.2f ms" % (os.path.basename(alt), other[0], other[1]))
