#!/bin/bash
# The measurements behind a round's numbers, in one place (one MI355X): bench lines, rocprofv3 kernel statistics (single-stream =
# the roofline leg's twin; two streams = the timed region), PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes), the
# secondary workloads.  Summaries land in gpurun_out/<tag>_prof and are copied into profiles/ with the round tag; every summary
# carries the build id of the library it was taken on (beyond_dof_amd._lib.build_id) and bench.py only quotes it on a match.
#   usage: tools/profile_round.sh r04 [quick]
tag=${1:-r04}; quick=$2
R=$PWD; out=$R/gpurun_out/${tag}_prof; mkdir -p $out $R/profiles; export TMPDIR=/tmp
cd $R
export BDOF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $out/p_1s -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof_1stream.json 2> $out/p_1s.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_1s -name "*.db" | head -1) $out/kernel_stats_1stream.csv && cp $out/kernel_stats_1stream.csv profiles/${tag}_kernel_stats_1stream.csv && cp $out/kernel_stats_1stream.build.json profiles/${tag}_kernel_stats_1stream.build.json
echo "1-stream kernel stats done"
rocprofv3 --pmc FETCH_SIZE -d $out/p_fetch -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $out/p_fetch.json 2> $out/p_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $out/p_write -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $out/p_write.json 2> $out/p_write.err || exit 1
python tools/pmc_summary.py $(find $out/p_fetch -name "*.db" | head -1) $(find $out/p_write -name "*.db" | head -1) $out/pmc_traffic_bench.json > $out/pmc_traffic.txt && cp $out/pmc_traffic_bench.json profiles/${tag}_pmc_traffic_bench.json
echo "pmc done"
unset BDOF_STREAMS
rocprofv3 --kernel-trace --stats -d $out/p_2s -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/p_2s.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_2s -name "*.db" | head -1) $out/kernel_stats.csv && cp $out/kernel_stats.csv profiles/${tag}_kernel_stats.csv && cp $out/kernel_stats.build.json profiles/${tag}_kernel_stats.build.json
rm -rf $out/p_1s $out/p_2s $out/p_fetch $out/p_write
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench (no flags) rc $?"; cp $out/bench_default.json profiles/${tag}_bench_default.json
python bench.py --steps 20 --warmup 5 > $out/bench_default_steps20.json 2> $out/bench_default_steps20.err; echo "bench 20 steps rc $?"; cp $out/bench_default_steps20.json profiles/${tag}_bench_default_steps20.json
if [ -z "$quick" ]; then
  BDOF_FORCE_COMM=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rccl_1rank.json 2> $out/bench_rccl_1rank.err; echo "rccl 1 rank rc $?"; cp $out/bench_rccl_1rank.json profiles/${tag}_bench_rccl_1rank.json
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_propagator.json 2> $out/bench_conv.err; echo "conv rc $?"; cp $out/bench_conv_propagator.json profiles/${tag}_bench_conv_propagator.json
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --recompute > $out/bench_recompute.json 2> $out/bench_recompute.err; echo "recompute rc $?"; cp $out/bench_recompute.json profiles/${tag}_bench_tape_free.json
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --size 256 --angles-per-gpu 50 --n-theta 50 > $out/bench_cfg2.json 2> $out/bench_cfg2.err; echo "cfg2 rc $?"; cp $out/bench_cfg2.json profiles/${tag}_bench_cfg2.json
  python tools/bench_ptycho.py 72 20 5 > $out/ptycho_cfg5_bench.txt 2>&1; tail -n 1 $out/ptycho_cfg5_bench.txt; cp $out/ptycho_cfg5_bench.txt profiles/${tag}_ptycho_cfg5_bench.txt
  CFG4_GRAD=1 python tools/bench_cfg4.py 4096 1024 $out/cfg4.json > $out/cfg4.txt 2>&1; tail -n 10 $out/cfg4.txt; cp $out/cfg4.json profiles/${tag}_cfg4_tiled.json
fi
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d = json.load(open(f)); r = d.get('roofline') or {}
        print(os.path.basename(f), round(d['ms_per_step'], 2), round(d['value']), 'frac', r.get('frac') and round(r['frac'], 3), 'whole', r.get('whole_step_frac') and round(r['whole_step_frac'], 3),
              r.get('kernel'), 'rocprof', r.get('avg_launch_ms_rocprof'), 'traffic', r.get('traffic_bytes_per_launch'), {k: round(v['avg_ms'] * 1e3, 1) for k, v in (r.get('per_kernel') or {}).items()})
    except Exception as e:
        print(f, 'failed', e)
PY
