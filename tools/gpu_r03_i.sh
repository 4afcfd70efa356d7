#!/bin/bash
# Round 3, GPU call I: whole parity suite, then the default bench line and the steady-state kernel trace (reset kernel at 512 threads)
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03
REPO=$(pwd)
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests_i.log 2>&1; echo "tests exit $?" >> $OUT/tests_i.log
tail -4 $OUT/tests_i.log
grep -q "tests exit 0" $OUT/tests_i.log || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.log 2>&1 || { tail -5 $OUT/bench_default.log | cut -c1-400; exit 1; }
tail -n 1 $OUT/bench_default.log > $OUT/bench_default.json
python -c "import json;d=json.load(open('$OUT/bench_default.json'));print('default',d['value'],d['steady_state']['value'],d['overlap_variant']['value'],d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_steady -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $REPO/$OUT/prof_steady.log 2>&1
echo "rocprof steady rc=$?"
