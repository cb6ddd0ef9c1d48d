#!/bin/bash
# round 3: the bench lines, rocprofv3 kernel statistics and PMC traffic behind the numbers of DESIGN §5 (one MI355X)
R=$PWD
out=$R/gpurun_out/r3_prof
mkdir -p $out
export TMPDIR=/tmp
cd $R
python bench.py --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err || exit 1
echo "bench default done"; python - <<PY
import json; d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['whole_step_frac'], {k: round(v['avg_ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()}, d['cpu_baseline']['value'])
PY
export BDOF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $out/p_1s -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof_1stream.json 2> $out/p_1s.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_1s -name "*.db" | head -1) $out/kernel_stats_1stream.csv
unset BDOF_STREAMS
rocprofv3 --kernel-trace --stats -d $out/p_2s -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/p_2s.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_2s -name "*.db" | head -1) $out/kernel_stats.csv
echo "kernel stats done"
export BDOF_STREAMS=1
rocprofv3 --pmc FETCH_SIZE -d $out/p_fetch -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $out/p_fetch.json 2> $out/p_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $out/p_write -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $out/p_write.json 2> $out/p_write.err || exit 1
unset BDOF_STREAMS
python tools/pmc_summary.py $(find $out/p_fetch -name "*.db" | head -1) $(find $out/p_write -name "*.db" | head -1) $out/pmc_traffic_bench.json > $out/pmc_traffic.txt
echo "pmc done"
BDOF_FORCE_COMM=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rccl_1rank.json 2> $out/bench_rccl_1rank.err; echo "rccl 1 rank rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_propagator.json 2> $out/bench_conv.err; echo "conv rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --recompute > $out/bench_recompute.json 2> $out/bench_recompute.err; echo "recompute rc $?"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --size 256 --angles-per-gpu 50 --n-theta 50 > $out/bench_cfg2.json 2> $out/bench_cfg2.err; echo "cfg2 rc $?"
python tools/bench_ptycho.py 72 20 5 > $out/ptycho_cfg5_bench.txt 2>&1; tail -n 1 $out/ptycho_cfg5_bench.txt
rm -rf $out/p_1s $out/p_2s $out/p_fetch $out/p_write
ls $out
