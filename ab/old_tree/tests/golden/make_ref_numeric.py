"""Generates tests/golden/ref_numeric.json from the REFERENCE's own
src/numericUtilities.f95, compiled where it lies under /root/reference by
`make -C oracle ref` into oracle/_ref/ref_numeric (our harness:
oracle/ref_numeric_harness.f90).  Run in the build container only; the JSON
(bit patterns of float32 outputs and integer indices) is what is committed."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_numeric")], text=True)
rec = {"lobatto_mus": {}, "lobatto_w": {}, "legendre": {}, "findindex": [], "findcdf": []}
for line in out.splitlines():
    t = line.split()
    if not t:
        continue
    if t[0] in ("lobatto_mus", "lobatto_w", "legendre"):
        rec[t[0]][t[1]] = [int(x) for x in t[2:]]
    elif t[0] == "findindex":
        rec["findindex"].append([int(x) for x in t[1:]])  # guess, value bits, real, double, mixed
    elif t[0] == "findcdf":
        rec["findcdf"].append([int(x) for x in t[1:]])
rec["_source"] = "reference src/numericUtilities.f95 built with amdflang -O2 (oracle/Makefile target ref)"
with open(os.path.join(ROOT, "tests", "golden", "ref_numeric.json"), "w") as f:
    json.dump(rec, f)
print("wrote ref_numeric.json", {k: len(v) for k, v in rec.items() if k != "_source"})
