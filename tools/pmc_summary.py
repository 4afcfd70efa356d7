"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
# one pass = one directory; gpurun_out accumulates over calls, so only the newest file of each pass counts
paths = []
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if files:
        paths.append(max(files, key=os.path.getmtime))
for path in paths:
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")[:60]
            acc[name][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
for name, ctrs in sorted(acc.items(), key=lambda kv: -sum(len(v) for v in kv[1].values())):
    if not any(k in name for k in ("gram_streamk", "colnorm", "chol_step", "gemm_ops", "burg_prox", "fw_")):
        continue
    print("==", name)
    for c, vals in sorted(ctrs.items()):
        print("   %-32s n=%4d mean=%.6g" % (c, len(vals), sum(vals) / len(vals)))
