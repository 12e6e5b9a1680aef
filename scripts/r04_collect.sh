#!/bin/bash
# round 4: copies what `bash scripts/r04.sh profiles parity` left under gpurun_out/r04/ into profiles/r04_* and refreshes
# profiles/pmc_shipped.json (run here, after the gpurun call has merged its output back)
set -eu
cd "$(dirname "$0")/.."
P=gpurun_out/r04/prof; Q=gpurun_out/r04/parity
cp $P/bench_stepcloud.json profiles/r04_bench_stepcloud.json
cp $P/bench_landsat.json profiles/r04_bench_landsat.json
cp $P/bench_lw.json profiles/r04_bench_config4_lw.json
cp $P/bench_radar.json profiles/r04_bench_radar.json
cp $P/bench_stepcloud_facebyface.json profiles/r04_bench_stepcloud_facebyface.json
cp $P/bench_stepcloud_rccl1.json profiles/r04_bench_stepcloud_rccl1rank.json
newest() { ls -t $1/*/*_kernel_stats.csv | head -1; }
cp "$(newest $P/stats_step)" profiles/r04_stepcloud_kernel_stats.csv
cp "$(newest $P/stats_landsat)" profiles/r04_landsat_kernel_stats.csv
cp "$(newest $P/stats_lw)" profiles/r04_config4_lw_kernel_stats.csv
python scripts/pmc_summary.py $P/pmc_step i3rcStepCloud 1e7 2.0 > /dev/null
python scripts/pmc_summary.py $P/pmc_landsat landsatLike128 1e7 1.0 > /dev/null
# config 4: one wavelength's launch (6.25e6 photons as 100 batches); bench.py scales by photons, so the record reads per 1e8-photon step
python scripts/pmc_summary.py $P/pmc_lw homogLW20x16 6.25e6 1.0 > /dev/null
python - <<'PY'
import json
for w in ("step", "landsat", "lw"):
    d = json.load(open("gpurun_out/r04/prof/pmc_%s/summary.json" % w))
    json.dump({k: v for k, v in d.get("trace_kernel", {}).items() if not k.startswith("_")}, open("profiles/r04_pmc_%s.json" % w, "w"), indent=1)
PY
if [ -d $Q ]; then
  cp $Q/parity_stepcloud.json profiles/r04_parity_stepcloud_1e9.json
  cp $Q/parity_landsat.json profiles/r04_parity_landsat_1e9.json
  cp $Q/parity_radar.json profiles/r04_parity_radar_4e8.json
fi
head -2 profiles/r04_stepcloud_kernel_stats.csv profiles/r04_landsat_kernel_stats.csv profiles/r04_config4_lw_kernel_stats.csv | cut -c1-160
git status --short profiles | head -30
