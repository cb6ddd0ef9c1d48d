#!/bin/bash
out=gpurun_out/r3_t2
mkdir -p $out
python -m pytest tests/test_gpu_resident.py -q > $out/a.log 2>&1; echo "resident file alone rc $?"; tail -n 3 $out/a.log
python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle" > $out/b.log 2>&1; echo "sizes_vs_oracle alone rc $?"; tail -n 3 $out/b.log
python -m pytest tests/test_gpu_resident.py -q -k "sizes_vs_oracle and (96 or 128)" > $out/c.log 2>&1; echo "96+128 rc $?"; tail -n 3 $out/c.log
for f in bilinear conv fullfield generic parity parity_ff probe ptycho recompute; do
python -m pytest tests/test_gpu_$f.py tests/test_gpu_resident.py -q -k "not resident or (sizes_vs_oracle and 128)" > $out/p_$f.log 2>&1; echo "$f + 128 rc $?"; tail -n 1 $out/p_$f.log
done
