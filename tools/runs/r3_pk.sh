#!/bin/bash
# round 3: complex add / sub as packed instructions (-DBDOF_PK_ADD) — LDS-resident kernel (cfg5) and streaming kernels (cfg3)
out=gpurun_out/r3_pk; mkdir -p $out
for v in default pk default pk; do
  if [ $v = default ]; then unset BDOF_LIB; else export BDOF_LIB=$PWD/beyond_dof_amd/libbdof_$v.so; fi
  python tools/bench_ptycho.py 72 20 5 > $out/pty_$v.txt 2>&1 || exit 1
  echo "$v cfg5: $(tail -n 1 $out/pty_$v.txt)"
  python tools/bench_ptycho.py 64 20 5 > $out/pty64_$v.txt 2>&1 || exit 1
  echo "$v 64^2: $(tail -n 1 $out/pty64_$v.txt)"
done
export BDOF_LIB=$PWD/beyond_dof_amd/libbdof_pk.so
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_pk.json 2> $out/bench_pk.err || exit 1
python - <<PY
import json
d=json.load(open('$out/bench_pk.json')); pk=d['roofline']['per_kernel']
print('pk cfg3: ms_per_step', round(d['ms_per_step'],2), {k: round(v['avg_ms']*1e3,1) for k,v in pk.items()})
PY
python -m pytest tests/test_gpu_resident.py tests/test_gpu_ptycho.py -q -x > $out/tests_pk.log 2>&1; echo "pk tests rc $?"; tail -n 2 $out/tests_pk.log
