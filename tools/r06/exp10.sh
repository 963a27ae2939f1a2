#!/bin/bash
# round 6, GPU call 10: parity sweeps beyond the test suite's with the round's final code (library bytes against oracle bytes), suite durations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp10; mkdir -p $O
export TMPDIR=/tmp
run() { label=$1; shift; ( time timeout 900 "$@" ) > $O/$label.out 2>&1; echo "$label: $(grep -E 'compared|mismatch|sequences|calls' $O/$label.out | tail -2 | tr '\n' ' ')" >> $O/summary.txt; }
run sweep_603_mutate_paths python tools/gpu_sweep.py 250 603 --mutate --paths
run sweep_604_history python tools/gpu_sweep.py 400 604 --mutate --history
run sweep_605_long python tools/gpu_sweep.py 120 605 --max-samples=14000000
run sweep_606_inplace env SRLA_MI355X_PIN_INPLACE=1 SRLA_MI355X_PACK_THREADS=1 python tools/gpu_sweep.py 160 606 --mutate --max-samples=9000000
run sweep_607_subregions env SRLA_MI355X_FFT_WP=2 python tools/gpu_sweep.py 200 607 --mutate
run batch_sweep python tools/gpu_batch_sweep.py 40 61
run reuse_sweep python tools/gpu_reuse_sweep.py 300 62
timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=25 > $O/suite_durations.out 2>&1; tail -32 $O/suite_durations.out >> $O/summary.txt
cat $O/summary.txt
