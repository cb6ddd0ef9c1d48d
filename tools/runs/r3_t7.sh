#!/bin/bash
out=gpurun_out/r3_t7
mkdir -p $out
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_z6.so python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle" > $out/z6.log 2>&1; echo "z6 rc $?"; tail -n 12 $out/z6.log
