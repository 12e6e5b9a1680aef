"""Generates tests/golden/ref_surface.json from the REFERENCE's own src/surfaceProperties.f95
(+ ErrorMessages.f95, numericUtilities.f95), compiled where they lie under /root/reference by
`make -C oracle ref` into oracle/_ref/ref_surface (our harness: oracle/ref_surface_harness.f90).
Run in the build container only; the JSON (bit patterns of the float64 positions and of the float32
reflectance computeSurfaceReflectance returned) is what is committed."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_surface")], text=True)
surfaces = {}
for line in out.splitlines():
    t = line.split()
    if not t:
        continue
    if t[0] == "surface":
        cur = surfaces.setdefault(t[1], {"numX": int(t[2]), "numY": int(t[3]), "xedge": [], "yedge": [], "points": []})
    elif t[0] in ("xedge", "yedge"):
        cur[t[0]].append(int(t[1]))
    elif t[0] == "point":
        surfaces[t[1]]["points"].append([int(t[2]), int(t[3]), int(t[4])])
rec = {"surfaces": surfaces,
       "_source": "reference src/surfaceProperties.f95 (computeSurfaceReflectance :119-147, makePeriodic :211-230, "
                  "findIndex of numericUtilities.f95) built with amdflang -O2 (oracle/Makefile target ref); patch (i, j) "
                  "carries reflectance real(i + 10 j)/100."}
with open(os.path.join(ROOT, "tests", "golden", "ref_surface.json"), "w") as f:
    json.dump(rec, f)
print("wrote ref_surface.json", {k: len(v["points"]) for k, v in surfaces.items()})
