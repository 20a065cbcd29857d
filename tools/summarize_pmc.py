"""Per-kernel means of the counter passes written by tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE / SQ_*)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "?"
head = sys.argv[3] if len(sys.argv) > 3 else "unknown"
res = defaultdict(dict)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # counter -> kernel -> dispatch -> value
    for fn in files:
        for row in csv.DictReader(open(fn)):
            acc[row["Counter_Name"]][row["Kernel_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, kerns in acc.items():
        for kern, disp in kerns.items():
            if not kern.startswith("eaqhm_"):
                continue
            key = name + ("_KB_mean_per_launch" if name in ("FETCH_SIZE", "WRITE_SIZE") else "_mean_per_launch")
            res[kern][key] = sum(disp.values()) / len(disp)
            res[kern]["launches"] = len(disp)
for kern, d in res.items():
    f = d.get("FETCH_SIZE_KB_mean_per_launch", 0.0) * 1024.0
    w = d.get("WRITE_SIZE_KB_mean_per_launch", 0.0) * 1024.0
    d["hbm_bytes_per_launch_raw"] = f + w
    d["hbm_bytes_per_launch_fetch_doubled"] = 2.0 * f + w
    busy, mfma = d.get("SQ_BUSY_CYCLES_mean_per_launch"), d.get("SQ_VALU_MFMA_BUSY_CYCLES_mean_per_launch")
    wave = d.get("SQ_WAVE_CYCLES_mean_per_launch")
    if wave:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if d.get(k + "_mean_per_launch") is not None:
                d[k + "_share_of_wave_cycles"] = d[k + "_mean_per_launch"] / wave
    if busy and mfma:
        d["mfma_busy_over_sq_busy"] = mfma / busy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
res["source_hash"] = bench.source_hash()      # bench.py withholds roofline.traffic when the tree's hash differs
res["git_head"] = head
res["workload"] = workload
res["_note"] = ("rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), bench.py --workload %s --steps 1 "
                "--warmup 0; FETCH/WRITE unit KB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of "
                "wide (16 B/lane) coalesced reads, other widths uncalibrated -> 'fetch_doubled' is the upper estimate "
                "used as roofline.traffic; SQ_* are summed over the chip's SQs as rocprofv3 reports them" % workload)
print(json.dumps(res, indent=1))
