#!/bin/bash
# Round 3, GPU call F: memoisation test, config-4 bench lines again, memoised steady state, then the rocprofv3 stages
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "memoized or overlapped_value or lockstep" > $OUT/tests_f.log 2>&1; echo "tests exit $?" >> $OUT/tests_f.log
tail -3 $OUT/tests_f.log
grep -q "tests exit 0" $OUT/tests_f.log || exit 1
for spec in "bench_cfg4 --config 4 --steps 100 --warmup 10 --no-steady --no-cpu-baseline" \
            "bench_cfg4_64 --config 4 --instances-per-gpu 64 --steps 50 --warmup 10 --no-steady --no-cpu-baseline" \
            "bench_memo_steady --memo-values --steps 20 --warmup 5 --no-cpu-baseline"; do
  set -- $spec; name=$1; shift
  timeout -k 10 300 python bench.py "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log | cut -c1-400; exit 1; }
  tail -n 1 $OUT/$name.log > $OUT/$name.json
  echo "bench $name ok"
done
STAGE=prof bash tools/gpu_prof_r03.sh || exit 1
STAGE=pmc bash tools/gpu_prof_r03.sh
