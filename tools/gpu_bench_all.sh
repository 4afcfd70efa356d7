#!/bin/bash
# Full round measurement on the GPU box: tests, default bench (with CPU baseline), other workloads.
mkdir -p gpurun_out
REPO=$(pwd)
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/tests.log 2>&1
rc=$?
tail -n 6 gpurun_out/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TESTS TIMED OUT"; exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
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
run poisson "--workload poisson_abpg --steps 50 --warmup 5"
run cfg4x8 "--m 512 --n 8192 --workload abpg --steps 100 --warmup 10 --no-cpu-baseline --instances-per-gpu 8"
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_main -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $REPO/gpurun_out/prof_main.log 2>&1
echo "rocprof main rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-variants > $REPO/gpurun_out/pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_write -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-variants > $REPO/gpurun_out/pmc_write.log 2>&1
echo "pmc write rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_sq -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-variants > $REPO/gpurun_out/pmc_sq.log 2>&1
echo "pmc sq rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_fw -- python3 $REPO/bench.py --workload fw --steps 100 --warmup 5 --no-cpu-baseline > $REPO/gpurun_out/prof_fw.log 2>&1
echo "rocprof fw rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_fwa -- python3 $REPO/bench.py --workload fw_away --steps 20 --warmup 3 --no-cpu-baseline > $REPO/gpurun_out/prof_fwa.log 2>&1
echo "rocprof fw_away rc=$?"
cd $REPO
python3 tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary_final.txt 2>&1
grep -A12 "gram_streamk_glds" gpurun_out/pmc_summary_final.txt | head -16
for d in prof_main prof_fw prof_fwa; do f=$(find gpurun_out/$d -name "*kernel_stats*.csv" | head -1); echo "== $d"; head -12 $f | cut -c1-200; done
exit $rc
