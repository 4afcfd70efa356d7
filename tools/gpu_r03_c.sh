#!/bin/bash
# Round 3, GPU call C: Gram ablation table, stream-K partition A/B at m >= 4096, batch limit test, config-4 bench lines
mkdir -p gpurun_out/r03
timeout -k 10 200 python tools/ablate_gram.py gpurun_out/r03/ablate_gram.json > gpurun_out/r03/ablate_gram.log 2>&1 || { tail -20 gpurun_out/r03/ablate_gram.log; exit 1; }
cat gpurun_out/r03/ablate_gram.json
timeout -k 10 400 python tools/gram_partition.py --out gpurun_out/r03/gram_partition.json > gpurun_out/r03/gram_partition.log 2>&1 || { tail -20 gpurun_out/r03/gram_partition.log; exit 1; }
grep shape gpurun_out/r03/gram_partition.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch_size_limit or lockstep or logical_shards or config5_shard or m8192" > gpurun_out/r03/tests_c.log 2>&1; echo "tests exit $?" >> gpurun_out/r03/tests_c.log
tail -4 gpurun_out/r03/tests_c.log
timeout -k 10 200 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r03/bench_cfg4.json 2> gpurun_out/r03/bench_cfg4.err || { tail -20 gpurun_out/r03/bench_cfg4.err; exit 1; }
cat gpurun_out/r03/bench_cfg4.json
timeout -k 10 300 python bench.py --config 4 --instances-per-gpu 64 --no-cpu-baseline > gpurun_out/r03/bench_cfg4_64.json 2> gpurun_out/r03/bench_cfg4_64.err || { tail -20 gpurun_out/r03/bench_cfg4_64.err; exit 1; }
cat gpurun_out/r03/bench_cfg4_64.json
timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline > gpurun_out/r03/bench_fw_away.json 2> gpurun_out/r03/bench_fw_away.err || { tail -20 gpurun_out/r03/bench_fw_away.err; exit 1; }
cat gpurun_out/r03/bench_fw_away.json
