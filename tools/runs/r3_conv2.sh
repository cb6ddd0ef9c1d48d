#!/bin/bash
# round 3: second tiling of the real-space propagator (k_conv2) — parity with the first, then timing of both
out=gpurun_out/r3_conv2; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_conv.py -q -x > $out/tests.log 2>&1; rc=$?; echo "conv tests rc $rc"; tail -n 5 $out/tests.log
[ $rc = 0 ] || exit 1
for t in 2 2; do
  BDOF_CONV_TILING=$t timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_t$t.json 2> $out/bench_t$t.err || exit 1
  python - <<PY
import json
d=json.load(open('$out/bench_t$t.json')); r=d['roofline']
print('tiling $t: ms_per_step', round(d['ms_per_step'],2), 'frac', round(r['frac'],3), 'whole', round(r.get('whole_step_frac',0),3), {k: round(v['avg_ms']*1e3,1) for k,v in r['per_kernel'].items()})
PY
done
BDOF_STREAMS=1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_1stream.json 2> $out/bench_1stream.err || exit 1
python - <<PY
import json
d=json.load(open('$out/bench_1stream.json')); r=d['roofline']
print('one stream: ms_per_step', round(d['ms_per_step'],2), {k: round(v['avg_ms']*1e3,1) for k,v in r['per_kernel'].items()})
PY
