#!/bin/bash
# round 3: sub-batch streams of the per-slice sweeps on the final kernels (round 1 measured 1 / 2 / 4: 77.0 / 67.7 / 66.1 ms)
out=gpurun_out/r3_streams; mkdir -p $out
for n in 2 3 4 5; do
  BDOF_STREAMS=$n python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $out/bench_$n.json 2> $out/bench_$n.err || exit 1
  python - <<PY
import json
d=json.load(open('$out/bench_$n.json')); print('streams $n: ms_per_step', round(d['ms_per_step'],2), 'whole', round((d.get('roofline') or {}).get('whole_step_frac') or 0,3))
PY
done
