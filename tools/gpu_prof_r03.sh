#!/bin/bash
# Round-3 measurement set (runs on the GPU box): bench lines, rocprofv3 kernel-trace summaries, PMC passes.
# Everything lands under gpurun_out/r03/; tools/collect_r03.py then writes the files kept under profiles/.
mkdir -p gpurun_out/r03
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03
export TMPDIR=/tmp
run_bench() {  # name, args...
  local name=$1; shift
  timeout -k 10 400 python bench.py "$@" > $OUT/$name.log 2>&1
  local rc=$?
  tail -n 1 $OUT/$name.log > $OUT/$name.json
  echo "bench $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
prof() {  # name, args...
  local name=$1; shift
  cd /tmp
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python3 $REPO/bench.py "$@" > $OUT/prof_$name.log 2>&1
  local rc=$?
  cd $REPO
  echo "rocprof $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
pmc() {  # name, counters..., then -- args
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  cd /tmp
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py "$@" > $OUT/pmc_$name.log 2>&1
  local rc=$?
  cd $REPO
  echo "pmc $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
if [ "${STAGE:-all}" = "all" ] || [ "$STAGE" = "bench" ]; then
run_bench bench_default --steps 20 --warmup 5
run_bench bench_fw --config 3 --workload fw --steps 400 --warmup 20
run_bench bench_fw_away --config 3 --steps 300 --warmup 20
run_bench bench_cfg4 --config 4 --steps 100 --warmup 10 --no-steady --no-cpu-baseline
run_bench bench_cfg4_single --config 4 --instances-per-gpu 1 --steps 100 --warmup 10 --no-steady --no-cpu-baseline
run_bench bench_cfg4_threads --config 4 --host-threads --steps 100 --warmup 10 --no-steady --no-cpu-baseline
run_bench bench_abpg --workload abpg --steps 20 --warmup 5 --no-cpu-baseline
run_bench bench_bpg --workload bpg --steps 20 --warmup 5 --no-cpu-baseline
run_bench bench_cfg1 --m 80 --n 200 --workload bpg --steps 500 --warmup 50 --no-cpu-baseline --no-steady
run_bench bench_cfg5_share --m 8192 --n 32768 --workload abpg --steps 5 --warmup 2 --no-cpu-baseline --no-steady --no-variants
run_bench bench_cfg5_full --config 5 --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline
run_bench bench_cfg4_64 --config 4 --instances-per-gpu 64 --steps 50 --warmup 10 --no-steady --no-cpu-baseline
run_bench bench_fw_away_exact --config 3 --steps 300 --warmup 20 --logdet-refresh 1 --no-cpu-baseline
run_bench bench_lingram_steady --linear-gram --steps 20 --warmup 5 --no-cpu-baseline
run_bench bench_memo_steady --memo-values --steps 20 --warmup 5 --no-cpu-baseline
run_bench bench_poisson --workload poisson_abpg --steps 50 --warmup 5
fi
if [ "${STAGE:-all}" = "all" ] || [ "$STAGE" = "prof" ] || [ "$STAGE" = "pmc" ]; then
prof default --steps 20 --warmup 5 --no-variants --no-steady --no-cpu-baseline --no-overlap
fi
if [ "${STAGE:-all}" = "all" ] || [ "$STAGE" = "prof" ]; then
prof steady --steps 20 --warmup 5 --no-variants --no-cpu-baseline
prof fw --config 3 --workload fw --steps 400 --warmup 20 --no-cpu-baseline
prof fw_away --config 3 --steps 300 --warmup 20 --no-cpu-baseline
prof cfg4 --config 4 --steps 100 --warmup 10 --no-steady --no-cpu-baseline
fi
if [ "${STAGE:-all}" = "all" ] || [ "$STAGE" = "pmc" ]; then
A="--steps 4 --warmup 2 --no-variants --no-steady --no-cpu-baseline --no-overlap"
pmc fetch FETCH_SIZE -- $A
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $A
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $A
fi
python3 tools/collect_r03.py $OUT > $OUT/collect.log 2>&1
tail -n 30 $OUT/collect.log
