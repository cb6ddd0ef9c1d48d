#!/bin/bash
out=gpurun_out/r3_full
mkdir -p $out
python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -n 6 $out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/runs/r3_profiles.sh
