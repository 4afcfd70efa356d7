#!/bin/bash
# Round 3, GPU call E: the whole parity suite on the kernels as they stand, schedule A/B once more, quick bench lines
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03/tests_e.log 2>&1; echo "tests exit $?" >> gpurun_out/r03/tests_e.log
tail -6 gpurun_out/r03/tests_e.log
grep -q "tests exit 0" gpurun_out/r03/tests_e.log || exit 1
timeout -k 10 200 python tools/kern_sched.py --rounds 3 --out gpurun_out/r03/kern_sched2.json > gpurun_out/r03/kern_sched2.log 2>&1 || { tail -20 gpurun_out/r03/kern_sched2.log; exit 1; }
grep -A4 "_median" gpurun_out/r03/kern_sched2.log
timeout -k 10 200 python tools/lingram_prefix.py gpurun_out/r03/lingram_prefix.json > gpurun_out/r03/lingram_prefix.log 2>&1 || { tail -20 gpurun_out/r03/lingram_prefix.log; exit 1; }
cat gpurun_out/r03/lingram_prefix.log
timeout -k 10 300 python bench.py > gpurun_out/r03/bench_default_e.json 2> gpurun_out/r03/bench_default_e.err || { tail -20 gpurun_out/r03/bench_default_e.err; exit 1; }
cat gpurun_out/r03/bench_default_e.json
