#!/bin/bash
# rocprofv3 kernel trace + the PMC passes of the headline bench command only (no variant segments).
mkdir -p gpurun_out
REPO=$(pwd)
rm -rf gpurun_out/prof_main gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_main -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $REPO/gpurun_out/prof_main.log 2>&1
rc=$?; echo "rocprof main rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-variants > $REPO/gpurun_out/pmc_fetch.log 2>&1
rc=$?; echo "pmc fetch rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_write -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-variants > $REPO/gpurun_out/pmc_write.log 2>&1
rc=$?; echo "pmc write rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_sq -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-variants > $REPO/gpurun_out/pmc_sq.log 2>&1
echo "pmc sq rc=$?"
cd $REPO
tail -n 1 gpurun_out/prof_main.log | cut -c1-400
python3 tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary_final.txt 2>&1
grep -A12 "gram_streamk_glds" gpurun_out/pmc_summary_final.txt | head -14
f=$(ls -t $(find gpurun_out/prof_main -name "*kernel_stats*.csv") | head -1); head -8 $f | cut -c1-90,290-420
