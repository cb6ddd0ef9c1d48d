#!/bin/bash
set -o pipefail
out=gpurun_out/r3_res
mkdir -p $out
python tools/gpu_diag_g17.py g17 > $out/diag_g17.log 2>&1 || exit 1
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_resfast.so python tools/gpu_diag_g17.py g17 > $out/diag_g17_resfast.log 2>&1 || exit 1
python tools/bench_ptycho.py 72 20 5 > $out/ptycho72.log 2>&1 || exit 1
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_resfast.so python tools/bench_ptycho.py 72 20 5 > $out/ptycho72_resfast.log 2>&1 || exit 1
python tools/bench_ptycho.py 64 20 5 > $out/ptycho64.log 2>&1 || exit 1
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_resfast.so python tools/bench_ptycho.py 64 20 5 > $out/ptycho64_resfast.log 2>&1 || exit 1
tail -n 3 $out/*.log
python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -n 15 $out/gpu_tests.log
