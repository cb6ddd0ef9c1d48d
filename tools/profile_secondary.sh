#!/bin/bash
# rocprofv3 kernel statistics of the secondary workloads (cfg5 on the LDS-resident engine, cfg3 through the real-space propagator) and
# the calibration / experiment log of tools/kbench.  usage: tools/profile_secondary.sh r04     (after tools/profile_round.sh)
tag=${1:-r04}
R=$PWD; out=$R/gpurun_out/${tag}_prof2; mkdir -p $out $R/profiles; export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $out/p_cfg5 -o r -- python3 tools/bench_ptycho.py 72 20 5 > $out/ptycho_under_rocprof.txt 2> $out/p_cfg5.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_cfg5 -name "*.db" | head -1) $out/ptycho_cfg5_kernel_stats.csv && cp $out/ptycho_cfg5_kernel_stats.csv profiles/${tag}_ptycho_cfg5_kernel_stats.csv && cp $out/ptycho_cfg5_kernel_stats.build.json profiles/${tag}_ptycho_cfg5_kernel_stats.build.json
export BDOF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $out/p_conv -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --propagator conv > $out/conv_under_rocprof.json 2> $out/p_conv.err || exit 1
python tools/rocpd_summary.py stats $(find $out/p_conv -name "*.db" | head -1) $out/kernel_stats_conv.csv && cp $out/kernel_stats_conv.csv profiles/${tag}_kernel_stats_conv.csv && cp $out/kernel_stats_conv.build.json profiles/${tag}_kernel_stats_conv.build.json
unset BDOF_STREAMS
rm -rf $out/p_cfg5 $out/p_conv
(cd tools && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o kbench kbench.hip 2>/dev/null; timeout -k 10 200 ./kbench 25 50 2) > $out/kbench_ceilings.txt 2>&1; cp $out/kbench_ceilings.txt profiles/${tag}_kbench_ceilings.txt
head -4 $out/ptycho_cfg5_kernel_stats.csv | cut -c1-150; head -4 $out/kernel_stats_conv.csv | cut -c1-150; tail -4 $out/kbench_ceilings.txt
