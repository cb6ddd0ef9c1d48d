#!/bin/bash
# round 3: `python bench.py --gpus 2` at cfg3's FULL size (512^3, 25 angles per rank) on one GPU through RcclComm and the library's
# collectives, librccl replaced by tests/rccl_stub: the true buffer sizes of the N-rank path (1 GiB gradient, slabs of 1 / 8 / 16 / 32).
out=gpurun_out/r3_rehearse_full; mkdir -p $out
/opt/rocm/bin/hipcc -O2 -std=c++17 -shared -fPIC -o $out/librccl_stub.so tests/rccl_stub/rccl_stub.cpp -I/opt/rocm/include || exit 1
export BDOF_RCCL_LIB=$PWD/$out/librccl_stub.so BDOF_STUB_SLOT_MB=1100
timeout -k 10 540 python bench.py --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $out/bench_n4_full.json 2> $out/bench_n4_full.err; echo "rc $?"
cat $out/bench_n4_full.json; grep -E "rccl communicator|setup|Traceback|Error" $out/bench_n4_full.err | head -8
rm -f $out/librccl_stub.so
