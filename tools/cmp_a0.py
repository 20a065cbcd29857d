"""Diagnostic: adaptation-0 records, fallback kernel (variant 2) against the tile kernel (variant 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, collections
import bench
from eaqhm_amd.engine import DeviceAnalysis, FramePlan
fs, s, grid, frames, fstep = bench.load_workload("sa19")
out = {}
for var in (2, 3):
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    eng = DeviceAnalysis(s, s, plan, 160, 5)
    eng.ctx.set_option(1, var)
    eng.ls_stage(0)
    torch.cuda.synchronize()
    out[var] = eng.records[0].cpu().numpy().copy()
K = np.asarray(plan.frame_K); inst = np.asarray(plan.frame_inst)
r2, r3 = out[2], out[3]
nan_rows = np.flatnonzero(~np.isfinite(r2).all(axis=1))
d = np.abs(r2 - r3).max(axis=1)
bad = np.flatnonzero(~(d <= 1e-6 * (1 + np.abs(r3).max(axis=1))))
print("rows", len(r2), "nan rows", len(nan_rows), "bad rows", len(bad))
imap = {int(i): q for q, i in enumerate(inst)}
tot = collections.Counter(int(k) for k in K); badk = collections.Counter(int(K[imap[int(b)]]) for b in bad if int(b) in imap)
print("K: bad/total", {k: (badk.get(k, 0), tot[k]) for k in sorted(tot)})
for b in bad[:5]:
    q = imap[int(b)]
    print("row", b, "K", K[q], "got", r2[b][:4], "ref", r3[b][:4])
