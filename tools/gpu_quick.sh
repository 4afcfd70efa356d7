#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for a in "--steps 10 --warmup 3 --no-cpu-baseline"; do
timeout -k 10 300 python bench.py $a 2>&1 | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['config']['workload'][:40], 'value=%.1f ms/step=%.3f'%(d['value'],d['ms_per_step']), {k:round(v['ms_avg'],4) for k,v in d['kernels'].items() if v['ms_avg']}, 'frac=%.3f'%d.get('roofline',{'frac':0})['frac'])"
done
exit $rc
