#!/bin/bash
# PMC passes (each its own rocprofv3 run, kernel-trace only) for the bench command.
mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $REPO/gpurun_out/counters_list.txt 2>&1
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline}"
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $line --output-format csv -d $REPO/gpurun_out/pmc$i -- python3 $REPO/bench.py $ARGS > $REPO/gpurun_out/pmc$i.log 2>&1
  rc=$?
  echo "pmc pass $i ($line) rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done <<< "${PMC_SETS}"
cd $REPO
python3 tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary.txt 2>&1
cat gpurun_out/pmc_summary.txt | head -80
