"""Copies one workload's bench line, rocprofv3 kernel stats and PMC summary from gpurun_out/ into profiles/ and
refreshes profiles/pmc_traffic.json.  usage: collect_profiles.py <step|landsat> <tag, e.g. v10>"""
import glob
import json
import shutil
import sys

w, tag = sys.argv[1], sys.argv[2]
name = {"step": "i3rcStepCloud", "landsat": "landsatLike128"}[w]
long = {"step": "stepcloud", "landsat": "landsat"}[w]
d = json.load(open("gpurun_out/bench_%s_%s.json" % (w, tag)))
r = d["roofline"]
print(w, "value %.4g" % d["value"], "frac %.4f" % r["frac"], "kernel ms %.3f" % r["kernel_ms_per_launch"], "thr", d["config"]["event_threshold"])
shutil.copy("gpurun_out/bench_%s_%s.json" % (w, tag), "profiles/r01_bench_%s_%s.json" % (long, tag))
ks = sorted(glob.glob("gpurun_out/stats_%s_%s/*/*_kernel_stats.csv" % (w, tag)))
shutil.copy(ks[-1], "profiles/r01_%s_%s_kernel_stats.csv" % (long, tag))
print(open(ks[-1]).read().splitlines()[1])
txt = open("gpurun_out/pmc_%s_%s/summary.txt" % (w, tag)).read()
pm = json.loads(txt[txt.index("{"):txt.rindex("}") + 1])
json.dump(pm, open("profiles/r01_pmc_%s_%s.json" % (w, tag), "w"), indent=1)
t = json.load(open("profiles/pmc_traffic.json"))
f, wr = pm["FETCH_SIZE"], pm["WRITE_SIZE"]
t[name] = {"fetch_kb_raw": f, "write_kb": wr, "hbm_bytes_per_launch": f * 1024 * 2 + wr * 1024,
           "hbm_bytes_per_launch_uncorrected": (f + wr) * 1024, "photons_per_launch": 10000000,
           "source": "profiles/r01_pmc_%s_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; launch = 1e7 photons, "
                     "fixed event threshold)" % (w, tag)}
json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
          "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_ATOMIC_sum", "FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES"):
    print("  %-24s %.4g" % (k, pm.get(k, float("nan"))))
