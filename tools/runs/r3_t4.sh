#!/bin/bash
out=gpurun_out/r3_t4
mkdir -p $out
python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle" > $out/a.log 2>&1; echo "default rc $?"; tail -n 2 $out/a.log
BDOF_NO_F64_DET=1 python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle" > $out/b.log 2>&1; echo "NO_F64_DET rc $?"; tail -n 2 $out/b.log
HIP_LAUNCH_BLOCKING=1 python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle" > $out/c.log 2>&1; echo "LAUNCH_BLOCKING rc $?"; tail -n 2 $out/c.log
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_resfast.so python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle" > $out/d.log 2>&1; echo "resfast rc $?"; tail -n 2 $out/d.log
