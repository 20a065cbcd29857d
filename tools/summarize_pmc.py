"""Per-kernel means of the FETCH_SIZE / WRITE_SIZE counter passes written by tools/profile_round.sh."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
res = defaultdict(dict)
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))   # kernel -> dispatch -> summed value
    for fn in files:
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] != name:
                continue
            acc[row["Kernel_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for kern, disp in acc.items():
        if not kern.startswith("eaqhm_"):
            continue
        res[kern][name + "_KB_mean_per_launch"] = sum(disp.values()) / len(disp)
        res[kern]["launches"] = len(disp)
for kern, d in res.items():
    f = d.get("FETCH_SIZE_KB_mean_per_launch", 0.0) * 1024.0
    w = d.get("WRITE_SIZE_KB_mean_per_launch", 0.0) * 1024.0
    d["hbm_bytes_per_launch_raw"] = f + w
    d["hbm_bytes_per_launch_fetch_doubled"] = 2.0 * f + w
res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0 "
                "(SA19, maxAdpt=5, 6 adaptations); counter unit KB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports "
                "half the bytes of wide (16 B/lane) coalesced reads, other widths uncalibrated -> 'fetch_doubled' is "
                "the upper estimate used as roofline.traffic")
print(json.dumps(res, indent=1))
