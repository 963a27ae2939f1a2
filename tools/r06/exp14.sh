#!/bin/bash
# round 6, GPU call 14: whole suite with the final code (PIN_INPLACE=1 unconditional), then host run-ahead (exp13)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp14; mkdir -p $O
export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite.out 2> $O/suite.err; echo "suite rc=$?" > $O/summary.txt; tail -3 $O/suite.out >> $O/summary.txt
cat $O/summary.txt
bash tools/r06/exp13.sh
