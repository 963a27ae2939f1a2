#!/bin/bash
# round 6, GPU call 12: the bench lines with the final summary in place, the default line, one whole-suite run with the shorter sweep streams, smoke()
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp12; mkdir -p $O
export TMPDIR=/tmp
python tools/profiles.py collect --tag r06b --bench-only > $O/bench_only.log 2>&1
( time timeout 600 python bench.py > $O/default_line.json 2> $O/default_line.err ) 2> $O/default_line.time
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.out 2>&1; tail -1 $O/smoke.out > $O/summary.txt
timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=6 > $O/suite.out 2> $O/suite.err; echo "suite rc=$?" >> $O/summary.txt; tail -9 $O/suite.out >> $O/summary.txt
cat $O/summary.txt $O/default_line.time
