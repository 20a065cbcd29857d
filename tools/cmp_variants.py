"""Diagnostic: records of adaptation 1 from the tile kernel (variant 3) against the fallback kernel (variant 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from eaqhm_amd.engine import DeviceAnalysis, FramePlan
fs, s, grid, frames, fstep = bench.load_workload("sa19")
out = {}
for var in (2, 3):
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    eng = DeviceAnalysis(s, s, plan, 160, 5)
    eng.ctx.set_option(1, var)
    eng.ls_stage(0); eng.post_stage(0)
    eng.ls_stage(1)
    torch.cuda.synchronize()
    out[var] = (eng.records[0].cpu().numpy().copy(), eng.ncol.cpu().numpy().copy(), eng.frame_inst.cpu().numpy().copy())
    fm = eng.fm_cur.cpu().numpy().reshape(plan.Kmax, -1); cols = eng.cols.cpu().numpy().reshape(-1, plan.Kmax)
    fc = eng.frame_c.cpu().numpy(); fwl = eng.frame_wl.cpu().numpy()
r2, ncol, inst = out[2]
r3 = out[3][0]
d = np.abs(r2 - r3).max(axis=1)
bad = np.flatnonzero(d > 1e-6 * (1 + np.abs(r2).max(axis=1)))
print("bad rows", len(bad), "of", len(r2))
imap = {int(i): q for q, i in enumerate(inst)}
for b in bad[:40]:
    q = imap.get(int(b), -1)
    n = int(ncol[q]) if q >= 0 else -1
    print(b, "frame", q, "n", n, "nt", (2 * (2 * n + 1) + 1 + 15) // 16, "maxdiff", d[b])
nt_all = (2 * (2 * ncol + 1) + 1 + 15) // 16
badset = set(int(b) for b in bad)
import collections
tot = collections.Counter(int(x) for x in nt_all)
badc = collections.Counter(); zero = collections.Counter()
for q, i in enumerate(inst):
    if int(i) in badset:
        badc[int(nt_all[q])] += 1
        if not r3[int(i)].any():
            zero[int(nt_all[q])] += 1
print("total by nt", dict(tot)); print("bad by nt", dict(badc)); print("all-zero rows by nt", dict(zero))
K = (r2.shape[1] - 1) // 3
for b in bad[:6]:
    dd = np.abs(r2[b] - r3[b]); w = np.flatnonzero(dd > 1e-6 * (1 + np.abs(r2[b])))
    print("row", b, "n", int(ncol[imap[int(b)]]), "wrong entries", len(w), "am slots", [int(x) for x in w if x < K][:50], "fm slots", [int(x - K) for x in w if K <= x < 2 * K][:50])
    print("   am ref", r2[b][:8], "\n   am got", r3[b][:8])

gap = np.zeros(len(inst), bool)
for q in range(len(inst)):
    n = int(ncol[q]); c = int(fc[q]); wl = int(fwl[q])
    gap[q] = bool((fm[cols[q, :n], c - wl:c + wl + 1] == 0).any())
isbad = np.array([int(i) in badset for i in inst])
for ntv in (8, 9, 10, 11, 12):
    m = nt_all == ntv
    print("nt", ntv, "frames", m.sum(), "gappy", (gap & m).sum(), "bad", (isbad & m).sum(), "bad&gappy", (isbad & gap & m).sum())
