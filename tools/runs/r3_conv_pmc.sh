#!/bin/bash
# round 3: where the waves of the real-space kernels are (SQ counters), both tilings; true traffic and rocprofv3 stats of the second
R=$PWD; out=$R/gpurun_out/r3_conv_pmc; mkdir -p $out; export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
export BDOF_STREAMS=1
for t in 1 2; do
  export BDOF_CONV_TILING=$t
  timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $out/sq_t$t -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --propagator conv > $out/sq_t$t.json 2> $out/sq_t$t.err || exit 1
  python tools/pmc_sq_summary.py $(find $out/sq_t$t -name "*counter_collection.csv" | head -1) $out/pmc_sq_conv_tiling$t.json k_conv
done
export BDOF_CONV_TILING=2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/p_fetch -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --propagator conv > $out/p_fetch.json 2> $out/p_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/p_write -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --propagator conv > $out/p_write.json 2> $out/p_write.err || exit 1
python tools/pmc_summary.py $(find $out/p_fetch -name "*.db" | head -1) $(find $out/p_write -name "*.db" | head -1) $out/pmc_traffic_conv.json | grep -i conv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/p_ks -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --propagator conv > $out/bench_conv_under_rocprof_1stream.json 2> $out/p_ks.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_ks -name "*.db" | head -1) $out/kernel_stats_conv.csv && head -6 $out/kernel_stats_conv.csv
rm -rf $out/sq_t1 $out/sq_t2 $out/p_fetch $out/p_write $out/p_ks
