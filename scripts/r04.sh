#!/bin/bash
# round 4: steps run on the GPU box, selected by name (one script instead of one file per gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash scripts/r04.sh tests poison base'
# steps: tests bounds poison poison_all base bench bench_lw lw_ab rates soak profiles benches parity ab
# Every step writes under gpurun_out/r04/ with a time stamp in the file name, so that a failing log is never overwritten.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
ts() { date +%m%d_%H%M%S; }

step_tests() {  # the whole GPU suite, one process
  local log=$OUT/tests_$(ts).log
  timeout -k 10 900 python -m pytest tests -q -m gpu -x ${PYTEST_ARGS:-} > $log 2>&1; local rc=$?
  tail -5 $log; grep -n "FAILED\|Timeout\|^E  " $log | head -20 | cut -c1-300
  echo "tests rc=$rc log=$log"; return $rc
}
step_bounds() {  # the tests that drive the kernels into their loop bounds
  local log=$OUT/bounds_$(ts).log
  timeout -k 10 600 python -m pytest tests/test_gpu_bounds.py -v -s -m gpu > $log 2>&1; local rc=$?
  grep -n "PASSED\|FAILED\|Timeout\|^E  \|legacy ties" $log | head -40 | cut -c1-300; tail -3 $log
  echo "bounds rc=$rc log=$log"; return $rc
}
step_poison() {  # the 60-seed in-process radiance run ONCE under the audit build (LDS and fresh device memory filled with 0xff)
  local log=$OUT/poison_$(ts).log
  MCBRAT_LIB=$ROOT/ab/libmcbrat_poison.so MCBRAT_FLIGHT_FUZZ=60 timeout -k 10 600 python -m pytest tests/test_gpu_intensity.py -v -m gpu > $log 2>&1; local rc=$?
  grep -c PASSED $log; grep -n "FAILED\|Timeout\|^E  " $log | head -20 | cut -c1-300; tail -2 $log
  echo "poison rc=$rc log=$log"; return $rc
}
step_poison_all() {  # the rest of the suite under the audit build
  local log=$OUT/poison_all_$(ts).log
  MCBRAT_LIB=$ROOT/ab/libmcbrat_poison.so timeout -k 10 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_intensity.py > $log 2>&1; local rc=$?
  tail -5 $log; grep -n "FAILED\|Timeout\|^E  " $log | head -20 | cut -c1-300
  echo "poison_all rc=$rc log=$log"; return $rc
}
step_base() {  # development timing of the three workloads (not the judged bench)
  local log=$OUT/base_$(ts).log
  { timeout -k 10 200 python scripts/quick_bench.py --case step --thr 16 --reps 5 --counters
    for lib in ab/libmcbrat_r03.so mcbrat3d_amd/libmcbrat_hip.so; do  # (the round-3 library, where a copy was kept for the A/B: git show 01a0af7, build, copy to ab/)
      [ -f $ROOT/$lib ] || continue
      echo "== $lib"
      MCBRAT_LIB=$ROOT/$lib MCBRAT_LIB_OLD=1 timeout -k 10 200 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 20 --reps 3
      MCBRAT_LIB=$ROOT/$lib MCBRAT_LIB_OLD=1 timeout -k 10 200 python scripts/quick_bench.py --case radar --ppb 1000000 --batches 100 --thr 24 --reps 3
    done; } > $log 2>&1
  grep "^==\|^opts\|^case\|lanes/phase" $log | cut -c1-330
  echo "base log=$log"
}
step_bench() {  # the judged line (step cloud + 128x128x64 secondary)
  local f=$OUT/bench_stepcloud_$(ts)
  python bench.py ${BENCH_ARGS:-} > $f.json 2> $f.err; local rc=$?
  python - "$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; s=d.get("secondary") or {}
print("value %.4g ms/step %.3f frac %.3f touched %.3f kernel %s" % (d["value"], d["ms_per_step"], r["frac"], r["frac_on_touched_bytes"], r["kernel"][:40]))
print("bad", d["config"].get("bad_photons"), "parity ok", (d.get("parity") or {}).get("within_thresholds"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
if s: print("secondary %.4g frac %.3f thr %s strong %s" % (s["value"], s["roofline_frac"], s["event_threshold"], json.dumps(s.get("strong_scaling",{}).get("projected_speedup"))))
PY
  echo "bench rc=$rc out=$f.json"; return $rc
}
step_bench_lw() {  # config 4
  local f=$OUT/bench_lw_$(ts)
  python bench.py --workload homogLW20x16 --steps 5 --warmup 1 ${BENCH_ARGS:-} > $f.json 2> $f.err; local rc=$?
  tail -3 $f.err
  python - "$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("LW value %.4g ms/step %.3f frac %.3f kernel_ms %.3f launches %s" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_per_launch"], d["config"]["kernel_launches_per_step"]))
print("bad", d["config"].get("bad_photons"), "parity", json.dumps({k:v for k,v in (d.get("parity") or {}).items() if k.startswith(("max_","within","z_domain"))}), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
  echo "bench_lw rc=$rc out=$f.json"; return $rc
}
step_soak() {  # the random differential tests with more seeds (SOAK seeds each, default 400), one log per test file
  local n=${SOAK:-400}
  for t in test_gpu_flight test_gpu_parity test_gpu_layer_skip test_gpu_block_walk test_gpu_intensity test_gpu_tunings; do
    local log=$OUT/soak_${t}_$(ts).log
    MCBRAT_FLIGHT_FUZZ=$n timeout -k 10 1000 python -m pytest tests/$t.py -q -m gpu -k "random" -p no:cacheprovider > $log 2>&1; local rc=$?
    echo "soak $t ($n seeds) rc=$rc: $(tail -1 $log)"; grep -n "FAILED\|Timeout\|^E  " $log | head -8 | cut -c1-250
    [ $rc -ne 0 ] && [ $rc -ne 1 ] && return $rc   # (a failed assertion goes on to the next file; a hang or crash stops the step)
  done
  return 0
}
step_benches() {  # the bench lines of the record; run AFTER profiles/pmc_shipped.json has been regenerated from this library's counters
                 # (profiles -> scripts/r04_collect.sh -> benches -> collect again), or they carry `counters_stale`
  local P=$OUT/prof; mkdir -p $P
  python bench.py --pipelined-extra > $P/bench_stepcloud.json 2> $P/bench_stepcloud.err; echo "bench step rc=$?"
  python bench.py --workload landsatLike128 --steps 5 --warmup 1 > $P/bench_landsat.json 2> $P/bench_landsat.err; echo "bench landsat rc=$?"
  python bench.py --workload homogLW20x16 --steps 5 --warmup 1 > $P/bench_lw.json 2> $P/bench_lw.err; echo "bench lw rc=$?"
  python bench.py --workload radarLike128 --steps 3 --warmup 1 --no-cpu-baseline > $P/bench_radar.json 2> $P/bench_radar.err; echo "bench radar rc=$?"
  python bench.py --block-walk 0 --no-cpu-baseline --no-secondary > $P/bench_stepcloud_facebyface.json 2> $P/bench_fbf.err; echo "bench fbf rc=$?"
  BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-secondary --steps 20 > $P/bench_stepcloud_rccl1.json 2> $P/bench_rccl1.err; echo "bench rccl rc=$?"
}
step_profiles() {  # the measurement record of the shipped kernels: rocprofv3 kernel stats, PMC passes (-> profiles/r04_*); then `benches`
  local P=$OUT/prof; mkdir -p $P
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_step -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --event-threshold 16 > $P/stats_step.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_landsat -- python3 $ROOT/bench.py --workload landsatLike128 --steps 5 --warmup 1 --no-cpu-baseline --event-threshold 20 > $P/stats_landsat.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_lw -- python3 $ROOT/bench.py --workload homogLW20x16 --steps 3 --warmup 1 --no-cpu-baseline > $P/stats_lw.log 2>&1 )
  bash scripts/pmc_profile.sh step r04/prof/pmc_step --thr 16 > $P/pmc_step.log 2>&1
  bash scripts/pmc_profile.sh landsat r04/prof/pmc_landsat --thr 20 > $P/pmc_landsat.log 2>&1
  bash scripts/pmc_profile.sh lw r04/prof/pmc_lw > $P/pmc_lw.log 2>&1
  find $P -name "*kernel_stats.csv" | xargs -n1 head -3 | cut -c1-200
  echo profiles done
}
step_parity() {  # large-sample parity records of the shipped kernels against the oracle's MT mode (-> profiles/r04_parity_*)
  local P=$OUT/parity; mkdir -p $P
  python bench.py --cpu-photons-per-core 20000000 --parity-photons 1000000000 --no-secondary > $P/parity_stepcloud.json 2> $P/parity_stepcloud.err; echo "step rc=$?"
  python bench.py --workload landsatLike128 --steps 3 --warmup 1 --parity-photons 1000000000 --cpu-photons-per-core 20000000 > $P/parity_landsat.json 2> $P/parity_landsat.err; echo "landsat rc=$?"
  python bench.py --workload radarLike128 --steps 3 --warmup 1 --parity-photons 400000000 --cpu-photons-per-core 16000000 > $P/parity_radar.json 2> $P/parity_radar.err; echo "radar rc=$?"
  python - $P <<'PY'
import json,sys
for w in ("stepcloud","landsat","radar"):
    d=json.loads(open(sys.argv[1]+"/parity_%s.json"%w).read().strip().splitlines()[-1]); p=d.get("parity") or {}
    print(w, "%.4g"%d["value"], "bad", d["config"].get("bad_photons"), json.dumps({k:v for k,v in p.items() if not isinstance(v,(list,dict))})[:700])
PY
}
step_rates() {  # issue cost of vector AND scalar instruction classes (scripts/valu_rates.hip, built into ab/valu_rates)
  local log=$OUT/valu_rates_$(ts).log
  timeout -k 10 120 $ROOT/ab/valu_rates > $log 2>&1; local rc=$?
  grep -c "waves/SIMD" $log; echo "rates rc=$rc log=$log"; return $rc
}
step_lw_ab() {  # config 4: the wide plan (tallies in LDS, one workgroup of 1024 lanes per CU) against the shared plan (global atomics)
  for wide in 1 0; do
    local f=$OUT/bench_lw_wide${wide}_$(ts)
    MCBRAT_WIDE=$wide python bench.py --workload homogLW20x16 --steps 5 --warmup 1 --no-cpu-baseline > $f.json 2> $f.err; local rc=$?
    python - "$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("LW value %.4g ms/step %.3f kernel_ms %.3f kernel %s bad %s" % (d["value"], d["ms_per_step"], r["kernel_ms_per_launch"], r["kernel"][:60], d["config"].get("bad_photons")))
PY
    echo "bench_lw wide=$wide rc=$rc out=$f.json"
  done
}
step_pmc_lw() {  # counters of config 4's kernel (one wavelength's launch), wide plan and without
  bash scripts/pmc_profile.sh lw r04/pmc_lw > $OUT/pmc_lw.log 2>&1
  MCBRAT_WIDE=0 bash scripts/pmc_profile.sh lw r04/pmc_lw_nowide > $OUT/pmc_lw_nowide.log 2>&1
  tail -3 $OUT/pmc_lw.log | cut -c1-200
}
step_lw_quick() {  # config 4: bench line without the CPU leg, and one wavelength's launch alone
  local f=$OUT/bench_lw_quick_$(ts)
  python bench.py --workload homogLW20x16 --steps 5 --warmup 1 --no-cpu-baseline > $f.json 2> $f.err; local rc=$?
  python - "$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("LW value %.4g ms/step %.3f kernel_ms %.3f kernel %s bad %s" % (d["value"], d["ms_per_step"], r["kernel_ms_per_launch"], r["kernel"][:200], d["config"].get("bad_photons")))
PY
  THR=16 LTHR=32 timeout -k 10 120 python scripts/lw_probe.py 62500 100 2>&1 | grep "counters False" | tail -1
  echo "lw_quick rc=$rc out=$f.json"
}
step_lw_sweep() {  # config 4, one wavelength's launch (6.25e6 photons as 100 batches): thresholds of the wide plan's kernel; stamps
  local log=$OUT/lw_sweep_$(ts).log
  for thr in 4 8 16 32; do for lthr in 16 32 48 64; do
    echo "== THR=$thr LTHR=$lthr" >> $log
    THR=$thr LTHR=$lthr timeout -k 10 120 python scripts/lw_probe.py 62500 100 2>&1 | grep "counters False" | tail -1 >> $log
  done; done
  echo "== stamps" >> $log
  MCBRAT_LIB=$ROOT/ab/libmcbrat_stamps.so timeout -k 10 120 python scripts/lw_probe.py 62500 100 >> $log 2>&1
  grep -A1 "^== " $log | grep -v "^--" | paste - - | cut -c1-160
  grep "stamp\|lanes per" $log | cut -c1-200
  echo "lw_sweep log=$log"
}
step_ab() {  # A/B of libraries in ab/: scripts/r04.sh ab <case> <thr> lib1 lib2 ...  (case: step | landsat | radar)
  local case=$1 thr=$2; shift 2
  local log=$OUT/ab_${case}_$(ts).log
  local args="--case $case --thr $thr --reps 3"
  [ "$case" != step ] && args="$args --ppb 1000000 --batches 100"
  for lib in "$@"; do
    echo "== $lib" >> $log
    MCBRAT_LIB=$ROOT/ab/$lib python scripts/quick_bench.py $args ${QB_ARGS:-} >> $log 2>&1
  done
  grep "^==\|^case\|lanes/phase" $log | cut -c1-330
  echo "ab log=$log"
}

rc=0
while [ $# -gt 0 ]; do
  s=$1; shift
  case $s in
    ab) step_ab "$@"; break ;;
    *) step_$s || { rc=$?; echo "step $s failed (rc=$rc): stopping"; break; } ;;
  esac
done
exit $rc
