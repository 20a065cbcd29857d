#!/usr/bin/env python3
"""Adds the derived matrix-pipe figures to <dir>/pmc_hbm_traffic.json from <dir>/kernel_stats.csv.
   tools/derive_mfma.py profiles/r02_synth16k_60s [kernel]"""
import csv
import json
import sys

d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "eaqhm_ls_tile_kernel"
avg_s = None
for row in csv.DictReader(open(d + "/kernel_stats.csv")):
    if row["Name"] == kern:
        avg_s = float(row["AverageNs"]) * 1e-9
j = json.load(open(d + "/pmc_hbm_traffic.json"))
k = j[kern]
n_mfma = k["SQ_INSTS_MFMA_mean_per_launch"]
busy = k["SQ_VALU_MFMA_BUSY_CYCLES_mean_per_launch"]
k["derived"] = {
    "avg_launch_s_from_kernel_stats": avg_s,
    "mfma_instructions_per_launch": n_mfma,
    "mfma_flops_issued_per_launch": n_mfma * 2048.0,       # v_mfma_f64_16x16x4_f64: 16*16*4 FMA = 2048 flop
    "mfma_busy_cycles_check": "SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = %.1f (v_mfma_f64_16x16x4_f64: 64 cycles per SIMD)" % (busy / n_mfma),
    "mfma_pipe_utilisation": busy / (avg_s * 2.4e9 * 1024),
    "mfma_pipe_utilisation_note": "busy SIMD-cycles / (launch duration x 2.4 GHz x 1024 SIMDs); nominal clock, so a lower bound "
                                  "if the chip clocks below 2.4 GHz under FP64 load",
}
json.dump(j, open(d + "/pmc_hbm_traffic.json", "w"), indent=1)
print(kern, k["derived"])
