#!/bin/bash
# round 3, second session, final state: GPU suite, smoke, rocprofv3 kernel statistics (FFT path 1 stream / 2 streams, real-space path), bench lines
R=$PWD; out=$R/gpurun_out/r3_final2; mkdir -p $out; export TMPDIR=/tmp
python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -n 4 $out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export BDOF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $out/p_1s -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof_1stream.json 2> $out/p_1s.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_1s -name "*.db" | head -1) $out/kernel_stats_1stream.csv && cp $out/kernel_stats_1stream.csv profiles/r03_kernel_stats_1stream.csv
rocprofv3 --kernel-trace --stats -d $out/p_conv -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_under_rocprof_1stream.json 2> $out/p_conv.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_conv -name "*.db" | head -1) $out/kernel_stats_conv.csv
unset BDOF_STREAMS
rocprofv3 --kernel-trace --stats -d $out/p_2s -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/p_2s.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_2s -name "*.db" | head -1) $out/kernel_stats.csv
echo "profiles done"
python bench.py > $out/bench_default_flags.json 2> $out/bench_default_flags.err; echo "bench (no flags) rc $?"
python bench.py --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_propagator.json 2> $out/bench_conv.err; echo "conv rc $?"
BDOF_CONV_TILING=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_propagator_first_tiling.json 2> $out/bench_conv1.err; echo "conv (first tiling) rc $?"
rm -rf $out/p_1s $out/p_2s $out/p_conv
python - <<PY
import json
for f in ('bench_default','bench_default_flags','bench_conv_propagator','bench_conv_propagator_first_tiling'):
    try:
        d=json.load(open('$out/%s.json'%f)); r=d.get('roofline') or {}
        print(f, round(d['ms_per_step'],2), round(d['value']), 'frac', r.get('frac') and round(r['frac'],3), 'whole', r.get('whole_step_frac') and round(r['whole_step_frac'],3), r.get('kernel'), 'rocprof', r.get('avg_launch_ms_rocprof'), 'traffic', r.get('traffic_bytes_per_launch'), {k: round(v['avg_ms']*1e3,1) for k,v in (r.get('per_kernel') or {}).items()})
    except Exception as e: print(f, 'failed', e)
PY
