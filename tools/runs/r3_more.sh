#!/bin/bash
R=$PWD; out=$R/gpurun_out/r3_more; mkdir -p $out
./tools/kbench_batch > $out/kbench_batch.txt 2>&1; cat $out/kbench_batch.txt
python bench.py --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
BDOF_FORCE_COMM=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rccl_1rank.json 2> $out/bench_rccl_1rank.err; echo "rccl 1 rank rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_propagator.json 2> $out/bench_conv.err; echo "conv rc $?"
python - <<PY
import json
for f in ('bench_default','bench_rccl_1rank','bench_conv_propagator'):
    try:
        d=json.load(open('$out/%s.json'%f)); r=d['roofline']
        print(f, round(d['ms_per_step'],2), 'frac', round(r['frac'],3), 'whole', round(r['whole_step_frac'],3), r['kernel'], 'exch', d['config']['exchange'], round(d['config']['exchange_ms'],2), {k: round(v['avg_ms']*1e3,1) for k,v in r['per_kernel'].items()})
    except Exception as e: print(f, 'failed', e)
PY
python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullfield.py -q > $out/tests.log 2>&1; echo "tests rc $?"; tail -n 3 $out/tests.log
export TMPDIR=/tmp
export BDOF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $out/p_conv -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_under_rocprof.json 2> $out/p_conv.err && python tools/rocpd_summary.py stats $(find $out/p_conv -name "*.db" | head -1) $out/kernel_stats_conv.csv
unset BDOF_STREAMS
rm -rf $out/p_conv
CFG4_ORACLE_SLICES=1024 CFG4_GRAD=1 python tools/bench_cfg4.py 4096 1024 $out/cfg4_tiled_vs_whole_field.json > $out/cfg4.log 2>&1; echo "cfg4 rc $?"; tail -n 25 $out/cfg4.log
