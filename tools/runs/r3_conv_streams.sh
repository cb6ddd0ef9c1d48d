#!/bin/bash
# round 3: the real-space path's Adam step against the number of sub-batch streams
out=gpurun_out/r3_conv_streams; mkdir -p $out
for s in 1 2 3 4; do
  BDOF_STREAMS=$s timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --propagator conv > $out/bench_s$s.json 2> $out/bench_s$s.err || exit 1
  python -c "import json; d=json.load(open('$out/bench_s$s.json')); print('streams $s: ms_per_step', round(d['ms_per_step'],2))"
done
