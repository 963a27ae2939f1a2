#!/bin/bash
# round-5 baseline on one box: bench line, kernel stats of the bench command, stand-alone kernel times (kernels serialised by a counter)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05_base}
mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_M.json 2> $O/bench_M.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/ktrace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $O/alone -o run -- python $R/tools/perf_probe.py 174.8 device 4 1 0 4096 > $O/alone.log 2>&1
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
best = collections.defaultdict(list)
for path in glob.glob(O + "/alone/*kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        best[(r["Kernel_Name"].split("(")[0][:60], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(O + "/alone_summary.txt", "w") as f:
    for (k, g), v in sorted(best.items()):
        f.write("%-60s grid %8d  n=%3d  min %8.1f us  median %8.1f us\n" % (k, g, len(v), min(v), sorted(v)[len(v)//2]))
PY
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
cat $O/alone_summary.txt
tail -c 1500 $O/bench_M.json
