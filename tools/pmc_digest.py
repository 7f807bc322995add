#!/usr/bin/env python3
"""Distil gpurun_out/pmc_<tag>/ (tools/pmc_any.sh) into a committed summary: profiles/<out>.json with the averaged counters of the
dominant kernel, its rocprofv3 --kernel-trace --stats line, and the derived numbers DESIGN.md quotes (HBM traffic with the guide's
gfx950 FETCH_SIZE x2 correction, ratio to the algorithmic bytes, LDS conflict share, instructions per wave).
The summary is stamped with the fingerprint of the kernel's sources (interpolate_antialiasing_amd/_lib.py KERNEL_SOURCES; bench.py
compares it with the tree it runs from and prints "traffic_stale") and with the register / LDS figures of the profiled kernel read
from the built code objects (csrc/*.o).
usage: tools/pmc_digest.py TAG KERNEL_SUBSTRING OUT_NAME ALG_BYTES [BATCH] [VARIANT]"""
import csv
import glob
import json
import os
import sys

tag, kern, out_name, alg = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
batch = int(sys.argv[5]) if len(sys.argv) > 5 else None
variant = sys.argv[6] if len(sys.argv) > 6 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", f"pmc_{tag}")
res = {}
for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k] = {"avg_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
stats = None
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row["Name"] and (stats is None or int(row["Calls"]) > stats["calls"]):  # the longest (warm-clock) run
            stats = {"kernel": row["Name"][:160], "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                     "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
g = lambda k: res.get(k, {}).get("avg_per_dispatch")
derived = {}
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    fetch_b, write_b = 2.0 * g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
    derived.update({"hbm_read_bytes (FETCH_SIZE KiB x2, guide's gfx950 correction)": int(fetch_b), "hbm_write_bytes (WRITE_SIZE KiB)": int(write_b),
                    "hbm_traffic_bytes": int(fetch_b + write_b), "algorithmic_bytes": int(alg),
                    "traffic_over_algorithmic": round((fetch_b + write_b) / alg, 4)})
if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
    derived["lds_conflict_share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
if g("SQ_WAVES"):
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if g(k) is not None:
            derived[k.lower().replace("sq_insts_", "") + "_per_wave"] = round(g(k) / g("SQ_WAVES"), 1)
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if g(k) is not None:
            derived[k.lower() + "_share_of_wave_cycles"] = round(g(k) / g("SQ_WAVE_CYCLES"), 4)
if stats and alg:
    derived["algorithmic_GBs_at_profiled_avg"] = round(alg / stats["avg_ns"], 1)
if g("GRBM_GUI_ACTIVE") and stats:
    derived["effective_clock_GHz (GRBM_GUI_ACTIVE / 8 / duration)"] = round(g("GRBM_GUI_ACTIVE") / 8.0 / stats["avg_ns"], 3)
if g("GRBM_GUI_ACTIVE") and g("SQ_ACTIVE_INST_VALU") is not None:
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; a SIMD issues one VALU instruction at a time, so over the chip's 1024
    # SIMDs and the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs) this is the share of time the vector ALUs were busy
    derived["valu_busy_share (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles))"] = round(g("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0), 4)
if g("GRBM_GUI_ACTIVE") and g("SQ_ACTIVE_INST_SCA") is not None:
    # a SIMD issues at most one scalar (SALU / SMEM) instruction every 4 cycles (the CU's scalar unit serves its 4 SIMDs in turn)
    derived["scalar_busy_share (SQ_ACTIVE_INST_SCA x 4 / (1024 SIMDs x cycles))"] = round(g("SQ_ACTIVE_INST_SCA") * 4.0 / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0), 4)
if g("GRBM_GUI_ACTIVE") and g("SQ_LDS_IDX_ACTIVE") is not None:
    derived["lds_busy_share (SQ_LDS_IDX_ACTIVE / (256 CUs x cycles))"] = round(g("SQ_LDS_IDX_ACTIVE") / (256.0 * g("GRBM_GUI_ACTIVE") / 8.0), 4)


def code_object_figures(demangled_name):
    """{vgpr_count, sgpr_count, lds_bytes, scratch_bytes} of the kernel whose demangled name starts like `demangled_name`, from the
    code objects embedded in csrc/*.o (None when the ROCm LLVM tools or the objects are missing)."""
    import re
    import shutil
    import subprocess
    import tempfile

    llvm = "/opt/rocm/lib/llvm/bin"
    objdump, readelf = (os.path.join(llvm, t) for t in ("llvm-objdump", "llvm-readelf"))
    filt = os.path.join(llvm, "llvm-cxxfilt") if os.path.exists(os.path.join(llvm, "llvm-cxxfilt")) else shutil.which("c++filt")
    if not (os.path.exists(objdump) and os.path.exists(readelf) and filt):
        return None
    def norm(n):  # name with its template arguments, without return type, parameter list and white space
        n = n.replace("(anonymous namespace)", "@anon@").replace("void ", "")
        return re.sub(r"\s+", "", n.split("(")[0]).rstrip(">")

    want = norm(demangled_name)
    for obj in sorted(glob.glob(os.path.join(root, "interpolate_antialiasing_amd", "csrc", "*.o"))):
        with tempfile.TemporaryDirectory() as td:
            shutil.copy(obj, os.path.join(td, "x.o"))
            if subprocess.run([objdump, "--offloading", "x.o"], cwd=td, capture_output=True).returncode != 0:
                continue
            for co in [f for f in os.listdir(td) if "gfx950" in f]:
                notes = subprocess.run([readelf, "--notes", os.path.join(td, co)], capture_output=True, text=True).stdout
                for blk in notes.split("- .agpr_count")[1:]:
                    m = re.search(r"\.name:\s+(\S+)", blk)
                    if not m:
                        continue
                    dem = subprocess.run([filt, m.group(1)], capture_output=True, text=True).stdout.strip()
                    have = norm(dem)
                    if not (have.startswith(want) or want.startswith(have)):  # (one side may print trailing default arguments)
                        continue
                    num = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1)) if re.search(r"\." + k + r":\s+(\d+)", blk) else None
                    return {"object": os.path.basename(obj), "vgpr_count": num("vgpr_count"), "sgpr_count": num("sgpr_count"),
                            "static_lds_bytes": num("group_segment_fixed_size"), "scratch_bytes": num("private_segment_fixed_size")}
    return None


out = {"workload_tag": tag, "kernel_substring": kern, "kernel_trace_stats": stats, "derived": derived}
if variant:
    sys.path.insert(0, root)
    from interpolate_antialiasing_amd import _lib

    out["variant"] = variant
    out["source_fingerprint"] = _lib.source_fingerprint(variant)
if stats:
    try:
        out["code_object"] = code_object_figures(stats["kernel"])
    except Exception as e:  # never lose a summary over the decoration
        out["code_object"] = {"error": str(e)[:200]}
if batch:
    out["batch"] = batch
out.update(res)
path = os.path.join(root, "profiles", out_name + ".json")
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(derived))
