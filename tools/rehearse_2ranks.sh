#!/bin/bash
# Two bench.py ranks on ONE GPU with gloo as the collective backend: rehearsal of the N-rank code path of bench.py
# (self-launch, sharded schedule, tune_tail, slab pipeline, sharded Adam, max-over-ranks timing).
# usage: tools/rehearse_2ranks.sh [bench args]
set -e
cd "$(dirname "$0")/.."
unset RANK WORLD_SIZE LOCAL_RANK
BDOF_COMM_BACKEND=gloo exec python bench.py --gpus 2 "$@"
