#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r01}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# profiled / series runs: the headline workload alone, steady state only (no cold region, no other workloads).  The chip needs
# ~20-60 ms of work to reach its steady clocks: 110 launches of the headline (0.83 ms each) are past that, 110 launches of the
# short workloads are not (rfft 0.2 ms, pconv 0.07 ms: rocprofv3's per-kernel average would be an average over the ramp), so
# those run longer
STEPS=100; WARM=10
case "$*" in
  *"--workload rfft131072"*) ;;
  *"--workload rfft"*) STEPS=1000; WARM=300 ;;
  *"--workload pconv"*) STEPS=3000; WARM=1000 ;;
esac
ARGS="--steps $STEPS --warmup $WARM --no-cpu-baseline --no-bandwidth --no-other-workloads --no-cold $*"
# the driver's command line first (20 steps, 5 warm-up, yardsticks included) ...
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $* > "$OUT/bench_driver_cmdline.json" 2> "$OUT/bench_unprofiled.err" || exit 1
echo "== driver command line"; cat "$OUT/bench_driver_cmdline.json"
python3 bench.py $ARGS > "$OUT/bench_unprofiled.json" 2>> "$OUT/bench_unprofiled.err" || exit 1
echo "== unprofiled"; cat "$OUT/bench_unprofiled.json"
# ... and the per-launch series of both command lines (an event after every launch: a run of its own), plus a
# cold start without the full-size guard (the chip's start-up clock ramp)
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bandwidth --no-other-workloads --series-out "$OUT/series_steps20.txt" $* > /dev/null 2>> "$OUT/bench_unprofiled.err"
python3 bench.py $ARGS --series-out "$OUT/series_steps$STEPS.txt" > /dev/null 2>> "$OUT/bench_unprofiled.err"
python3 bench.py --steps 60 --warmup 0 --settle-ms 0 --no-cpu-baseline --no-bandwidth --no-other-workloads --no-selfcheck --series-out "$OUT/series_cold_start.txt" $* > /dev/null 2>> "$OUT/bench_unprofiled.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 2; }
echo "== stats done"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  D="$OUT/pmc_$(echo $C | tr ' ' '_')"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D" -- python3 bench.py $ARGS > "$D.log" 2>&1 || { echo "pmc $C failed"; tail -3 "$D.log"; }
  echo "== pmc $C done"
done
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D="$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D" -- python3 bench.py $ARGS > "$D.log" 2>&1 || { echo "pmc $C failed"; tail -3 "$D.log"; }
  echo "== pmc $C done"
done
python3 tools/pmc_summary.py "$OUT" || true
