#!/bin/bash
out=gpurun_out/r3_t6
mkdir -p $out
for v in z1 z2 z3 z4 z5; do
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_$v.so python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle" > $out/$v.log 2>&1; echo "$v rc $?"; tail -n 1 $out/$v.log
done
