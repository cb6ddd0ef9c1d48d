#!/bin/bash
out=gpurun_out/r3_t1
mkdir -p $out
python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle and 128" > $out/t128_default.log 2>&1; echo "default rc $?"
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_resfast.so python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle and 128" > $out/t128_resfast.log 2>&1; echo "resfast rc $?"
grep -h "passed\|failed\|FAILED" $out/t128_default.log $out/t128_resfast.log
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_resfast.so python -m pytest tests -q -m gpu > $out/gpu_tests_resfast.log 2>&1; echo "suite (resfast) rc $?"; tail -n 8 $out/gpu_tests_resfast.log
