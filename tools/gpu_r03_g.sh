#!/bin/bash
# Round 3, GPU call G: column-blocked Gram product for long rows -- A/B, parity of everything that touches it, config-5 bench
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03
timeout -k 10 400 python tools/gram_partition.py --flags 8,0 --shapes 2048x131072,2048x262144,8192x262144 --out $OUT/gram_blocked.json > $OUT/gram_blocked.log 2>&1 || { tail -20 $OUT/gram_blocked.log; exit 1; }
grep -o "'shape.*rel_gap_g': [0-9.e-]*\|baseline_ms.*" $OUT/gram_blocked.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5 or m8192 or logical_shards or library_side or unaligned or poisson_large or runs_are_bitwise" > $OUT/tests_g.log 2>&1; echo "tests exit $?" >> $OUT/tests_g.log
tail -3 $OUT/tests_g.log
grep -q "tests exit 0" $OUT/tests_g.log || exit 1
timeout -k 10 400 python bench.py --config 5 --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_full.log 2>&1 || { tail -5 $OUT/bench_cfg5_full.log | cut -c1-300; exit 1; }
tail -n 1 $OUT/bench_cfg5_full.log > $OUT/bench_cfg5_full.json
cut -c1-900 $OUT/bench_cfg5_full.json
