#!/bin/bash
# Round 3, GPU call J: the parity suite on HEAD once more, and a whole 1000-iteration run with and without memoisation
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests_j.log 2>&1; echo "tests exit $?" >> $OUT/tests_j.log
tail -4 $OUT/tests_j.log
grep -q "tests exit 0" $OUT/tests_j.log || exit 1
timeout -k 10 200 python tools/long_run_check.py $OUT/long_run.json > $OUT/long_run.log 2>&1 || { tail -5 $OUT/long_run.log; exit 1; }
cat $OUT/long_run.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
