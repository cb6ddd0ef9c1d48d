#!/bin/bash
# round 3: dithered taps of the real-space propagator against one nearest-rounded set (BDOF_TW_DITHER=0)
out=gpurun_out/r3_convdither; mkdir -p $out; rm -f $out/summary.log
for v in dither one; do
  if [ $v = one ]; then export BDOF_TW_DITHER=0; else unset BDOF_TW_DITHER; fi
  python -m pytest tests/test_gpu_fullfield.py tests/test_gpu_ptycho.py -q -s -k "directional" > $out/tests_$v.log 2>&1
  echo "== $v: pytest rc $?" >> $out/summary.log
  grep -h -A3 "^G2\|^.G2\|^FG2" $out/tests_$v.log >> $out/summary.log
done
cat $out/summary.log
