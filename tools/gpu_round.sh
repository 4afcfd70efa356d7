#!/bin/bash
# Runs on the GPU box: parity tests, then (only if the tests did not time out) a short bench,
# then (PROFILE=1) a rocprofv3 kernel trace of the same bench command.
mkdir -p gpurun_out
REPO=$(pwd)
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q --timeout 600 ${PYTEST_ARGS:-} > gpurun_out/tests.log 2>&1
rc=$?
tail -n ${TAIL:-40} gpurun_out/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TESTS TIMED OUT rc=$rc"; exit $rc; fi
echo "tests rc=$rc"
else rc=0; fi
timeout -k 10 ${BENCH_TIMEOUT:-400} python bench.py ${BENCH_ARGS:---steps 5 --warmup 2 --no-cpu-baseline} > gpurun_out/bench.log 2>&1
brc=$?
tail -n 3 gpurun_out/bench.log
echo "bench rc=$brc"
if [ $brc -eq 124 ] || [ $brc -eq 137 ]; then exit $brc; fi
if [ "${PROFILE:-0}" = "1" ]; then
  export TMPDIR=/tmp
  cd /tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -- python3 $REPO/bench.py ${BENCH_ARGS:---steps 5 --warmup 2 --no-cpu-baseline} > $REPO/gpurun_out/prof.log 2>&1
  echo "rocprof rc=$?"
  cd $REPO
  find gpurun_out/prof -name "*kernel_stats*.csv" | head -3
  for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -20 $f; done
fi
exit $rc
