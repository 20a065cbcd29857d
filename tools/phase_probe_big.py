"""Diagnostic: where eaqhm_ls_mfma_kernel (large frames) spends its cycles in an adaptation >= 1 launch — in-kernel
s_memtime stamps of thread 0 of every workgroup:   python tools/phase_probe_big.py [workload] [seconds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from eaqhm_amd import prologue
from eaqhm_amd.engine import DeviceAnalysis, FramePlan
wl = sys.argv[1] if len(sys.argv) > 1 else "synth48k_60s"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
fs, s, grid, frames, fstep = bench.load_workload(wl)
s = s[:int(secs * fs)]
grid = grid[:len(np.arange(0, len(s) - 1, round(fs * 5 / 1000)))]
frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
prologue.apply_full_waveform(frames, len(s), 32 * 15)
plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
eng = DeviceAnalysis(s, s, plan, 160, 5)
it = eng.adaptations()
next(it); next(it)          # adaptation 0 complete, adaptation 1 enqueued (slot lists of a = 1; the scratch has its final size)
torch.cuda.synchronize()
eng.ctx.set_option(2, 1)    # stamps on, counters cleared
launches, p = 2, eng.plan
for r in range(launches):
    eng.ctx.ls_batch(1, eng.s, p.L, p.fs, eng.am_cur, eng.fm_cur, eng.track_t0, eng.track_len, p.Kmax, eng.frame_inst, eng.frame_c,
                     eng.frame_wl, eng.frame_f0, eng.frame_K, eng.ncol, eng.cols, eng.seeded, eng.any_seed, eng.nf, p.wl_max, 1,
                     p.f0_stale, eng.f0min, eng.records[0], None, None)
torch.cuda.synchronize()
d = eng.ctx.debug_read()
names = ["set-up, slot preparation, clears", "basis build of a chunk (to its barrier)", "contraction of a chunk (to its barrier)",
         "accumulators -> system tiles", "factorisation: block update (to its barrier)", "factorisation: in-block update + tile load",
         "factorisation: diagonal tile (barrier, diag_coop, W store)", "factorisation: panel tiles", "factorisation: end-of-column barrier",
         "back substitution", "record"]
tot = sum(d[:11])
for n, v in zip(names, d[:11]):
    print("%-62s %16d cycles  %5.1f%%" % (n, v, 100.0 * v / max(tot, 1)))
print("   of which: in-block j-loop %d, tile load + combine %d (slot 5 above = the latter); barrier in front of diag_coop %d" % (d[14], d[5], d[15]))
print("workload", wl, secs, "s, frames", eng.nf, "launches", launches, "cycles/frame", tot / max(1, launches * eng.nf))
