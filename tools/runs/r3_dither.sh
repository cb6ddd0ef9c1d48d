#!/bin/bash
# round 3: dithered twiddle tables (plain float32 transforms, a different rounding of the tables per slice) against the hi + lo tables
set -o pipefail
out=gpurun_out/r3_dither
mkdir -p $out
rm -f $out/summary.log
run() {   # name lib dither
  name=$1
  if [ -n "$2" ]; then export BDOF_LIB=$PWD/beyond_dof_amd/$2; else unset BDOF_LIB; fi
  if [ -n "$3" ]; then export BDOF_TW_DITHER=$3; else unset BDOF_TW_DITHER; fi
  echo "== $name" >> $out/summary.log
  python tools/gpu_check_cfg3_depth.py > $out/depth_$name.log 2>&1 || { echo "depth failed" >> $out/summary.log; return 1; }
  tail -4 $out/depth_$name.log >> $out/summary.log
  python -m pytest tests/test_gpu_fullfield.py -q -s -k "reference_loop" > $out/tests_$name.log 2>&1
  echo "pytest rc $?" >> $out/summary.log
  grep -h "stats" $out/tests_$name.log >> $out/summary.log
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_$name.json 2> $out/bench_$name.err || return 1
  python - <<PY >> $out/summary.log
import json
d=json.load(open('$out/bench_$name.json'))
pk=d['roofline']['per_kernel']
print('  bench $name: ms_per_step', round(d['ms_per_step'],2), {k: round(v['avg_ms']*1e3,1) for k,v in pk.items()})
PY
}
run ${1:-fast_d64} libbdof_fast.so ${2:-64} && run fast libbdof_fast.so ""
cat $out/summary.log
