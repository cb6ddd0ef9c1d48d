#!/bin/bash
out=gpurun_out/r3_t8
mkdir -p $out
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_wp.so python -m pytest tests/test_gpu_resident.py -q -s -k "sizes_vs_oracle" > $out/wp.log 2>&1; echo "wp rc $?"; tail -n 3 $out/wp.log
