#!/bin/bash
# Round 3, GPU call A: new parity tests (config 5 at its size, m = 8192 golden, default-path long horizons, FW ring),
# the FW-away log-det mode comparison, and the config-5 bench at full size on one GPU.
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
  -k "m8192 or config5 or overlapped_value or 1000_iterations or gain_300 or past_first_retries or fw or housing or gives_up" \
  > gpurun_out/r03/tests_a.log 2>&1; echo "tests exit $?" >> gpurun_out/r03/tests_a.log
tail -5 gpurun_out/r03/tests_a.log
grep -q "tests exit 0" gpurun_out/r03/tests_a.log || exit 1
timeout -k 10 300 python tools/fw_away_modes.py --out gpurun_out/r03/fw_modes.json > gpurun_out/r03/fw_modes.log 2>&1 || { tail -20 gpurun_out/r03/fw_modes.log; exit 1; }
tail -3 gpurun_out/r03/fw_modes.log
timeout -k 10 400 python bench.py --config 5 --gpus 1 --steps 5 --warmup 2 > gpurun_out/r03/bench_cfg5_full.json 2> gpurun_out/r03/bench_cfg5_full.err || { tail -20 gpurun_out/r03/bench_cfg5_full.err; exit 1; }
cat gpurun_out/r03/bench_cfg5_full.json
