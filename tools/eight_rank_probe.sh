#!/bin/bash
# 8 ranks on the ONE GPU of the test box with the host budget of an 8-GPU node (one pool thread per rank): per-rank time spread.
#   usage: tools/eight_rank_probe.sh [calls-per-step] [steps]
C=${1:-16}; S=${2:-3}
for rep in 1 2 3; do
  SRLA_BENCH_SHARED_GPU=1 python bench.py --gpus 8 --steps $S --warmup 1 --seconds 60 --calls-per-step $C --no-cpu-baseline --pack-threads 1 2>/dev/null | grep '^{' | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  per_rank %s' % (d['value'], d['per_rank']))"
done
