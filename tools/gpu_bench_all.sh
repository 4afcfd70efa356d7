#!/bin/bash
# Full round measurement on the GPU box: tests, default bench (with CPU baseline), other workloads.
mkdir -p gpurun_out
REPO=$(pwd)
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/tests.log 2>&1
rc=$?
tail -n 6 gpurun_out/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TESTS TIMED OUT"; exit $rc; fi
run() { # name, args
  timeout -k 10 600 python bench.py $2 > gpurun_out/bench_$1.log 2>&1
  b=$?; echo "bench $1 rc=$b"; tail -n 1 gpurun_out/bench_$1.log | cut -c1-1600
  if [ $b -eq 124 ] || [ $b -eq 137 ]; then exit $b; fi
}
run default ""
run abpg "--workload abpg --steps 20 --warmup 3 --no-cpu-baseline"
run bpg "--workload bpg --steps 20 --warmup 3 --no-cpu-baseline"
run fw "--workload fw --steps 200 --warmup 10 --cpu-iters 5"
run fw_away "--workload fw_away --steps 40 --warmup 5 --cpu-iters 3"
run cfg4 "--m 512 --n 8192 --workload abpg --steps 100 --warmup 10 --no-cpu-baseline"
run cfg1 "--m 80 --n 200 --workload bpg --steps 500 --warmup 20 --no-cpu-baseline"
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_fw -- python3 $REPO/bench.py --workload fw --steps 100 --warmup 5 --no-cpu-baseline > $REPO/gpurun_out/prof_fw.log 2>&1
echo "rocprof fw rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_fwa -- python3 $REPO/bench.py --workload fw_away --steps 20 --warmup 3 --no-cpu-baseline > $REPO/gpurun_out/prof_fwa.log 2>&1
echo "rocprof fw_away rc=$?"
cd $REPO
for d in prof_fw prof_fwa; do f=$(find gpurun_out/$d -name "*kernel_stats*.csv" | head -1); echo "== $d"; head -12 $f | cut -c1-200; done
exit $rc
