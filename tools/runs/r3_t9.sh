#!/bin/bash
out=gpurun_out/r3_t9
mkdir -p $out
BDOF_LIB=$PWD/beyond_dof_amd/libbdof_dbg.so python -m pytest tests/test_gpu_generic.py -q -s -k "sizes_vs_oracle and 72-72 and numpy_skip_last-None" > $out/dbg.log 2>&1; echo "dbg rc $?"
python -m pytest tests/test_gpu_generic.py -q -s -k "sizes_vs_oracle and 72-72 and numpy_skip_last-None" > $out/cur.log 2>&1; echo "cur rc $?"
grep -A3 "dbg72" $out/dbg.log | head -30; tail -n 3 $out/dbg.log $out/cur.log
