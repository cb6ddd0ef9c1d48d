#!/bin/bash
out=gpurun_out/r3_t5
mkdir -p $out
for v in v3 v4 resfast; do
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_$v.so python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle" > $out/$v.log 2>&1; echo "$v rc $?"; tail -n 2 $out/$v.log
done
