"""Summarises rocprofv3 --pmc CSVs: per-launch mean of every counter over the dispatches of the tracing kernel
(trace_kernel / trace_block_kernel, the non-instrumented instantiation), plus the derived figures bench.py's
`roofline.valu` object and `roofline.traffic` are built from.

    python scripts/pmc_summary.py <dir with pass*/ subdirectories> [workload photons_per_launch fetch_correction]

With the three extra arguments the derived record is also merged into profiles/pmc_shipped.json under `workload`.
fetch_correction: 2.0 where the kernel's HBM-side reads are wide coalesced streams (the guide's gfx950 correction:
FETCH_SIZE counts 64 B per 128-B request), 1.0 where they are 4-byte gathers (uncalibrated there; raw)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcbrat3d_amd.build import sources_sha256  # noqa: E402

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
files = []
for d in sorted(glob.glob(os.path.join(root, "pass*"))):  # (a pass directory may hold the files of earlier runs: the newest one counts)
    cand = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if cand:
        files.append(max(cand, key=os.path.getmtime))
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            tracing = ("trace_kernel" in k or "trace_block_kernel" in k)
            # the instrumented instantiation (DEBUG: third template argument of trace_block_kernel, fifth of trace_kernel) is not the product
            targs = [a.strip() for a in k.split("<", 1)[1].split(">(")[0].split(",")] if tracing and "<" in k else []
            debug = tracing and len(targs) > 4 and (targs[2] if "trace_block_kernel" in k else targs[4]) == "true"
            short = "trace_kernel" if tracing and not debug else k.split("(")[0][-40:]
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

VALU_CYCLES = {"plain": 2.3, "F64": 4.3, "CVT": 4.25, "INT64": 4.5, "TRANS": 8.2}


def valu_cycles(g):
    """SIMD cycles the launch's vector instructions occupy: every class at its measured issue cost, the rest as plain."""
    f64 = (g("SQ_INSTS_VALU_ADD_F64") or 0) + (g("SQ_INSTS_VALU_MUL_F64") or 0) + (g("SQ_INSTS_VALU_FMA_F64") or 0)
    trans = (g("SQ_INSTS_VALU_TRANS_F32") or 0) + (g("SQ_INSTS_VALU_TRANS_F64") or 0)
    i64, cvt = g("SQ_INSTS_VALU_INT64") or 0, g("SQ_INSTS_VALU_CVT") or 0
    plain = g("SQ_INSTS_VALU") - f64 - trans - i64 - cvt
    return (VALU_CYCLES["plain"] * plain + VALU_CYCLES["F64"] * f64 + VALU_CYCLES["CVT"] * cvt + VALU_CYCLES["INT64"] * i64 +
            VALU_CYCLES["TRANS"] * trans)


if len(sys.argv) >= 5 and t:
    workload, ppl, corr = sys.argv[2], int(float(sys.argv[3])), float(sys.argv[4])
    g = lambda k: t.get(k)  # noqa: E731
    rec = {"photons_per_launch": ppl,
           "valu_insts_per_launch": g("SQ_INSTS_VALU"), "salu_insts_per_launch": g("SQ_INSTS_SALU"),
           "lds_insts_per_launch": g("SQ_INSTS_LDS"), "waves_per_launch": g("SQ_WAVES"),
           # lanes that do work per VALU instruction: thread-cycles / (64 x wave-level active cycles)
           "lane_occupancy": g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") else None,
           # wave lifetime: issuing / parked in s_waitcnt / ready but not issued (the three are disjoint and add up to ~1)
           "wave_time_issuing": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
           "wave_time_waiting": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
           "wave_time_issue_stall": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
           "lds_bank_conflict_ratio": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None,
           "l2_requests_per_launch": g("TCC_REQ_sum"), "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") is not None else None,
           "l2_misses_per_launch": g("TCC_MISS_sum"), "global_atomics_per_launch": g("TCC_ATOMIC_sum"),
           # instruction classes (a pass of their own) and what they cost the SIMD: cycles per wave64 instruction measured on
           # this part with 8 waves per SIMD issuing independent instructions (scripts/valu_rates.hip, profiles/r03_valu_rates.txt):
           # plain f32 / int32 2.3, f64 add / mul / fma 4.3, conversions 4.25 (f64 <-> f32; the counter also holds cheaper ones),
           # 64-bit integer (v_mad_u64_u32: Philox) 4.5, transcendental (rcp, log, sqrt, sin) 8.2
           "valu_class_insts_per_launch": {k: g("SQ_INSTS_VALU_" + k) for k in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F32", "TRANS_F64", "INT64", "INT32", "CVT")}
           if g("SQ_INSTS_VALU_INT64") is not None else None,
           "valu_simd_cycles_per_launch": valu_cycles(g) if g("SQ_INSTS_VALU_INT64") is not None else None,
           "fetch_kb_raw": g("FETCH_SIZE"), "write_kb": g("WRITE_SIZE"), "fetch_correction": corr,
           "hbm_bytes_per_launch": (corr * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0 if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None else None,
           # which kernels these counters belong to: bench.py compares with the sources beside it and marks the record stale on mismatch
           "kernel_sources_sha256": sources_sha256(),
           "source": "rocprofv3 --pmc, separate passes (scripts/pmc_profile.sh), %s; launch = %d photons, fixed event threshold" % (os.path.basename(os.path.normpath(root)), ppl)}
    shipped = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_shipped.json")
    cur = {}
    if os.path.exists(shipped):
        with open(shipped) as fh:
            cur = json.load(fh)
    cur[workload] = rec
    with open(shipped, "w") as fh:
        json.dump(cur, fh, indent=1)
    print("merged", workload, "into", shipped)
