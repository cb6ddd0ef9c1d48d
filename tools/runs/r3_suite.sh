#!/bin/bash
out=gpurun_out/r3_suite
mkdir -p $out
python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -n 12 $out/gpu_tests.log
python tools/bench_ptycho.py 72 20 5 > $out/ptycho72.log 2>&1; tail -n 1 $out/ptycho72.log
python tools/bench_ptycho.py 64 20 5 > $out/ptycho64.log 2>&1; tail -n 1 $out/ptycho64.log
