#!/bin/bash
# round 3, final state: GPU suite, rocprofv3 kernel statistics + PMC traffic (copied into profiles/ BEFORE the bench reads them), bench lines
R=$PWD; out=$R/gpurun_out/r3_final; mkdir -p $out; export TMPDIR=/tmp
python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -n 4 $out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export BDOF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $out/p_1s -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof_1stream.json 2> $out/p_1s.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_1s -name "*.db" | head -1) $out/kernel_stats_1stream.csv && cp $out/kernel_stats_1stream.csv profiles/r03_kernel_stats_1stream.csv
rocprofv3 --pmc FETCH_SIZE -d $out/p_fetch -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $out/p_fetch.json 2> $out/p_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $out/p_write -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $out/p_write.json 2> $out/p_write.err || exit 1
python tools/pmc_summary.py $(find $out/p_fetch -name "*.db" | head -1) $(find $out/p_write -name "*.db" | head -1) $out/pmc_traffic_bench.json > $out/pmc_traffic.txt && cp $out/pmc_traffic_bench.json profiles/r03_pmc_traffic_bench.json
unset BDOF_STREAMS
rocprofv3 --kernel-trace --stats -d $out/p_2s -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/p_2s.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_2s -name "*.db" | head -1) $out/kernel_stats.csv
echo "profiles done"
python bench.py > $out/bench_default_flags.json 2> $out/bench_default_flags.err; echo "bench (no flags) rc $?"
python bench.py --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
BDOF_FORCE_COMM=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rccl_1rank.json 2> $out/bench_rccl_1rank.err; echo "rccl 1 rank rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_propagator.json 2> $out/bench_conv.err; echo "conv rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --recompute > $out/bench_recompute.json 2> $out/bench_recompute.err; echo "recompute rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --size 256 --angles-per-gpu 50 --n-theta 50 > $out/bench_cfg2.json 2> $out/bench_cfg2.err; echo "cfg2 rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --rotation bilinear > $out/bench_bilinear.json 2> $out/bench_bilinear.err; echo "bilinear rc $?"
python tools/bench_ptycho.py 72 20 5 > $out/ptycho_cfg5_bench.txt 2>&1; tail -n 1 $out/ptycho_cfg5_bench.txt
rocprofv3 --kernel-trace --stats -d $out/p_pty -o r -- python3 tools/bench_ptycho.py 72 20 5 > $out/p_pty.txt 2> $out/p_pty.err && python tools/rocpd_summary.py stats $(find $out/p_pty -name "*.db" | head -1) $out/ptycho_cfg5_kernel_stats.csv
rm -rf $out/p_1s $out/p_2s $out/p_fetch $out/p_write $out/p_pty
python - <<PY
import json
for f in ('bench_default','bench_default_flags','bench_rccl_1rank','bench_conv_propagator','bench_recompute','bench_cfg2','bench_bilinear'):
    try:
        d=json.load(open('$out/%s.json'%f)); r=d.get('roofline') or {}
        print(f, round(d['ms_per_step'],2), round(d['value']), 'frac', r.get('frac') and round(r['frac'],3), 'whole', r.get('whole_step_frac') and round(r['whole_step_frac'],3), r.get('kernel'), 'rocprof', r.get('avg_launch_ms_rocprof'), 'traffic', r.get('traffic_bytes_per_launch'), d['config']['exchange'], round(d['config']['exchange_ms'],2), {k: round(v['avg_ms']*1e3,1) for k,v in (r.get('per_kernel') or {}).items()})
    except Exception as e: print(f, 'failed', e)
PY
