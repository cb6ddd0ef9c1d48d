#!/bin/bash
out=gpurun_out/r3_t12
mkdir -p $out
python -m pytest tests/test_gpu_resident.py tests/test_gpu_generic.py tests/test_gpu_ptycho.py tests/test_gpu_probe.py tests/test_gpu_tiling.py tests/test_gpu_convergence.py tests/test_gpu_conv.py -q -s > $out/tests.log 2>&1; echo "tests rc $?"
grep -h "stats\|convergence\|cfg4 tiles\|tile 512\|passed\|failed\|^FAILED" $out/tests.log | cut -c1-400
python tools/gpu_diag_g17.py g14 > $out/diag_g14.log 2>&1; cat $out/diag_g14.log
