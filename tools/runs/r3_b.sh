#!/bin/bash
out=gpurun_out/r3_b; mkdir -p $out
./tools/kbench_batch > $out/kbench_batch.txt 2>&1; cat $out/kbench_batch.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open('$out/bench.json')); r=d['roofline']
print(round(d['ms_per_step'],2), 'frac', round(r['frac'],3), 'whole', round(r['whole_step_frac'],3), r['kernel'], {k: round(v['avg_ms']*1e3,1) for k,v in r['per_kernel'].items()})
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullfield.py tests/test_gpu_recompute.py tests/test_gpu_tiling.py -q -x > $out/tests.log 2>&1; echo "tests rc $?"; tail -n 3 $out/tests.log
