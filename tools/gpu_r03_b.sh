#!/bin/bash
# Round 3, GPU call B: FW-away log-det forms test, schedule A/B of the Gram / gradient kernels
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "logdet_forms or fw_trajectories or large_fw" \
  > gpurun_out/r03/tests_b.log 2>&1; echo "tests exit $?" >> gpurun_out/r03/tests_b.log
tail -5 gpurun_out/r03/tests_b.log
timeout -k 10 300 python tools/kern_sched.py --out gpurun_out/r03/kern_sched.json > gpurun_out/r03/kern_sched.log 2>&1 || { tail -20 gpurun_out/r03/kern_sched.log; exit 1; }
tail -40 gpurun_out/r03/kern_sched.log
