#!/bin/bash
# round 3: non-temporal loads of the read-once operands (modulation-factor rows in A / A', tape rows in A')
out=gpurun_out/r3_nt; mkdir -p $out
for v in default nt1 nt2 default; do
  if [ $v = default ]; then unset BDOF_LIB; else export BDOF_LIB=$PWD/beyond_dof_amd/libbdof_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err || exit 1
  python - <<PY
import json
d=json.load(open('$out/bench_$v.json')); pk=d['roofline']['per_kernel']
print('$v: ms_per_step', round(d['ms_per_step'],2), {k: round(v['avg_ms']*1e3,1) for k,v in pk.items()})
PY
done
