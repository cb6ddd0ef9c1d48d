#!/bin/bash
# Two bench.py ranks on ONE GPU with gloo as the collective backend: rehearsal of the N-rank code path of bench.py
# (sharded schedule, tune_allreduce, slab pipeline, max-over-ranks timing).  usage: tools/rehearse_2ranks.sh [bench args]
set -e
cd "$(dirname "$0")/.."
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29577} WORLD_SIZE=2 LOCAL_RANK=0 BDOF_COMM_BACKEND=gloo
RANK=1 python bench.py --gpus 2 "$@" > /dev/null 2> gpurun_out/rehearse_rank1.err &
pid=$!
RANK=0 python bench.py --gpus 2 "$@" 2> gpurun_out/rehearse_rank0.err
wait $pid
