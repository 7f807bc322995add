import csv, glob, sys, collections, json
tag = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "fused"
res = collections.OrderedDict()
import os
files = {}
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    k = f.split("/")[2]
    if k not in files or os.path.getmtime(f) > os.path.getmtime(files[k]): files[k] = f
for f in [files[k] for k in sorted(files)]:
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k] = {"avg_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
print(json.dumps(res, indent=1))
