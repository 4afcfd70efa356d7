#!/bin/bash
# Round 3, GPU call H: parallel away-step search, long-row Gram test, then the FW bench lines
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fw or housing or gives_up or kyinit or expo_abda" > $OUT/tests_h.log 2>&1; echo "tests exit $?" >> $OUT/tests_h.log
tail -4 $OUT/tests_h.log
grep -q "tests exit 0" $OUT/tests_h.log || exit 1
for spec in "bench_fw_away --config 3 --steps 300 --warmup 20" "bench_fw_away_exact --config 3 --steps 300 --warmup 20 --logdet-refresh 1 --no-cpu-baseline" "bench_fw --config 3 --workload fw --steps 400 --warmup 20"; do
  set -- $spec; name=$1; shift
  timeout -k 10 300 python bench.py "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log | cut -c1-400; exit 1; }
  tail -n 1 $OUT/$name.log > $OUT/$name.json
  python -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',round(d['value'],1),d['roofline']['frac'],d['whole_step']['frac_of_hbm_peak'])"
done
