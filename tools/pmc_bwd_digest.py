#!/usr/bin/env python3
"""Digest of tools/pmc_bwd_widths.sh: per input width W the backward kernel's duration and its L2 <-> memory counters, and the
differences against the first width -> profiles/r03_bwd_906_vs_896.json.  usage: tools/pmc_bwd_digest.py W [W ...]"""
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"workload": "backward of config A, gather form: grad [256,3,196,320] -> [256,3,438,W] fp32 (tools/workload.py bwdw:W)", "widths": {}}
for w in sys.argv[1:]:
    d = os.path.join(root, "gpurun_out", "bwdw", w)
    rec = {}
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "fused_f32_nchw_up" in row["Name"] or "fused" in row["Name"]:
                if "avg_us" not in rec or int(row["Calls"]) > rec["calls"]:
                    rec.update({"kernel": row["Name"][:120], "calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3})
    for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "fused" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            rec[k] = sum(v) / len(v)
    alg_out = 256 * 3 * 438 * int(w) * 4
    alg_in = 256 * 3 * 196 * 320 * 4
    rec["algorithmic_write_bytes"] = alg_out
    rec["algorithmic_read_bytes"] = alg_in
    if "avg_us" in rec:
        rec["algorithmic_GBs"] = round((alg_out + alg_in) / rec["avg_us"] / 1e3, 1)
    if "WRITE_SIZE" in rec:
        rec["write_bytes_over_algorithmic (WRITE_SIZE KiB)"] = round(rec["WRITE_SIZE"] * 1024 / alg_out, 4)
    if "FETCH_SIZE" in rec:
        rec["fetch_bytes_over_algorithmic (FETCH_SIZE KiB x2)"] = round(2 * rec["FETCH_SIZE"] * 1024 / alg_in, 4)
    if rec.get("TCC_EA0_WRREQ_sum"):
        rec["bytes_per_write_request (algorithmic)"] = round(alg_out / rec["TCC_EA0_WRREQ_sum"], 1)
        if "TCC_EA0_WRREQ_64B_sum" in rec:
            rec["share_of_64B_write_requests"] = round(rec["TCC_EA0_WRREQ_64B_sum"] / rec["TCC_EA0_WRREQ_sum"], 4)
    out["widths"][w] = rec
path = os.path.join(root, "profiles", "r03_bwd_906_vs_896.json")
json.dump(out, open(path, "w"), indent=1)
print(path)
for w, rec in out["widths"].items():
    print(w, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in rec.items() if k not in ("kernel",)})
