#!/bin/bash
# round 3: `python bench.py --gpus 4` rehearsed on ONE GPU — the product's RcclComm and the library's collectives, librccl replaced by
# tests/rccl_stub (RCCL refuses several ranks on one device).  Results are not performance figures: four ranks share the card and the
# stand-in stages through the host.
out=gpurun_out/r3_rehearse4; mkdir -p $out
/opt/rocm/bin/hipcc -O2 -std=c++17 -shared -fPIC -o $out/librccl_stub.so tests/rccl_stub/rccl_stub.cpp -I/opt/rocm/include || exit 1
export BDOF_RCCL_LIB=$PWD/$out/librccl_stub.so BDOF_STUB_SLOT_MB=600
timeout -k 10 500 python bench.py --gpus 4 --size 256 --angles-per-gpu 8 --n-theta 64 --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $out/bench_n4.json 2> $out/bench_n4.err; echo "rc $?"
cat $out/bench_n4.json; grep -E "rccl communicator|tail|slab" $out/bench_n4.err | head -12
rm -f $out/librccl_stub.so
