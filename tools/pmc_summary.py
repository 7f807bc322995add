import csv, glob, sys, collections, json
tag = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "fused"
res = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k] = {"avg_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
print(json.dumps(res, indent=1))
