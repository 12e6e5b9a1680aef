"""Summarises rocprofv3 --pmc CSVs: per-kernel mean of every counter over dispatches of trace_kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            short = "trace_kernel" if "trace_kernel" in k else k.split("(")[0][-40:]
            acc[short][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
out = {}
for kern, ctrs in acc.items():
    out[kern] = {}
    for name, vals in ctrs.items():
        per = defaultdict(float)
        for d, v in vals:
            per[d] += v  # sum over dimensions (XCC / SE instances) of one dispatch
        out[kern][name] = sum(per.values()) / len(per)
        out[kern]["_dispatches_" + name] = len(per)
t = out.get("trace_kernel", {})
print(json.dumps({k: v for k, v in t.items() if not k.startswith("_")}, indent=1))
with open(os.path.join(root, "summary.json"), "w") as fh:
    json.dump(out, fh, indent=1)
