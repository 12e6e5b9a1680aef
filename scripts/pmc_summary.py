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
            short = "trace_kernel" if ("trace_kernel" in k or "trace_block_kernel" in k) and "true>(" not in k.split("DevParams")[0][-12:] else k.split("(")[0][-40:]
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

# HBM-side traffic per launch of the tracing kernel, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE are in KB, from separate passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a
# wide coalesced stream (x2); this kernel's reads are 4-8 B gathers, for which the factor is
# uncalibrated, so both raw and corrected values are kept.
if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
    traffic = {"fetch_kb_raw": t["FETCH_SIZE"], "write_kb": t["WRITE_SIZE"],
               "hbm_bytes_per_launch": (2.0 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0,
               "hbm_bytes_per_launch_uncorrected": (t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0}
    with open(os.path.join(root, "traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
