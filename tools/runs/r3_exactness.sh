#!/bin/bash
# round 3: which forward transforms have to be exact for G18, and what each variant costs on the bench
set -o pipefail
out=gpurun_out/r3_exact
mkdir -p $out
python tools/gpu_diag_g17.py g17 > $out/diag_g17.log 2>&1 || exit 1
BDOF_NO_F64_DET=1 python tools/gpu_diag_g17.py g17 > $out/diag_g17_nof64det.log 2>&1 || exit 1
for v in default fastfwd exA exB; do
  if [ $v = default ]; then unset BDOF_LIB; else export BDOF_LIB=$PWD/beyond_dof_amd/libbdof_$v.so; fi
  python -m pytest tests/test_gpu_fullfield.py -q -s -k "reference_loop" > $out/tests_$v.log 2>&1
  echo "variant $v: pytest rc $?" >> $out/summary.log
  grep -h "stats" $out/tests_$v.log >> $out/summary.log
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err || exit 1
  python - <<PY >> $out/summary.log
import json
d=json.load(open('$out/bench_$v.json'))
pk=d['roofline']['per_kernel']
print('  bench $v: ms_per_step', round(d['ms_per_step'],2), {k: round(v['avg_ms']*1e3,1) for k,v in pk.items()})
PY
done
cat $out/summary.log
