#!/bin/bash
# round 6, GPU call 19: more parity sweeps with the final build (evidence only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp19; mkdir -p $O
export TMPDIR=/tmp
run() { label=$1; shift; ( time timeout 1200 "$@" ) > $O/$label.out 2>&1; echo "$label: $(grep -E 'compared|mismatch|sequences|calls' $O/$label.out | tail -2 | tr '\n' ' ')" >> $O/summary.txt; }
run sweep_621 python tools/gpu_sweep.py 500 621 --mutate --paths
run sweep_622 python tools/gpu_sweep.py 700 622 --mutate --history
run sweep_623 python tools/gpu_sweep.py 200 623 --max-samples=20000000
run sweep_624_p3 env SRLA_MI355X_FIR_MFMA=0 python tools/gpu_sweep.py 200 624 --mutate
run capacity python tools/gpu_capacity_sweep.py
run batch python tools/gpu_batch_sweep.py 60 63
run reuse python tools/gpu_reuse_sweep.py 400 64
cat $O/summary.txt
