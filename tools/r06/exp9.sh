#!/bin/bash
# round 6, GPU call 9: the bench lines again with this round's summary in place, three more whole-suite runs (8 - 10 of 10), the assembly's LDS at -B 8192
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp9; mkdir -p $O
export TMPDIR=/tmp
python tools/profiles.py collect --tag r06 --bench-only > $O/bench_only.log 2>&1
( time timeout 600 python bench.py > $O/default_line.json 2> $O/default_line.err ) 2> $O/default_line.time
for run in 8 9 10; do
  timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite_$run.out 2> $O/suite_$run.err
  echo "suite $run rc=$?" >> $O/summary.txt; tail -1 $O/suite_$run.out >> $O/summary.txt
done
ls gpurun_out/abort_* >> $O/summary.txt 2>&1
cat gpurun_out/test_process_at_exit.txt >> $O/summary.txt
for rep in 1 2 3; do
 for words in 0 5632; do
  line=$(SRLA_MI355X_PACK_LDS_WORDS=$words timeout 300 python bench.py --config C4 --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs --no-extras 2>/dev/null | grep '^{' | tail -1)
  python - "$words" "$line" >> $O/summary.txt <<'P'
import json, sys
w, line = sys.argv[1:3]
try:
    d = json.loads(line); st = d["roofline"]["stages"]
    print("C4 PACK_LDS_WORDS=%-6s value %8.1f  stages(ms/job): %s" % (w, d["value"], " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items() if v.get("ms_per_job"))))
except Exception as e:
    print("C4", w, "FAILED", e, line[:200])
P
 done
done
cat $O/summary.txt; cat $O/default_line.time
