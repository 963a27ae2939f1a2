# kernel timing experiment (DIAG build only: make EXTRA=-DSRLA_DIAG_STOP): duration of the largest srla_residual_cost dispatches
# cut short at successive points (SRLA_MI355X_K3_STOP=1..5; 21: the whole kernel with every variant loaded as one channel, 22: the
# same cut after the loads), kernels serialised by the counter collection.  STOPS="1 22 0 21" selects
#   gpurun -- bash tools/diag_residual_cost.sh [V] [P] [B]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/diag_rc
mkdir -p $O
rm -f $O/result.txt
for st in ${STOPS:-1 2 3 4 5 0}; do
  if [ $st = 0 ]; then unset SRLA_MI355X_K3_STOP; else export SRLA_MI355X_K3_STOP=$st; fi
  rm -rf /tmp/dd; timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d /tmp/dd -o run -- python $R/tools/perf_probe.py 174.8 device 4 ${1:-1} ${2:-0} ${3:-4096} > /tmp/dd.log 2>&1
  python - "$st" >> $O/result.txt <<'PY'
import csv, glob, sys, collections
best = collections.defaultdict(list)
for path in glob.glob("/tmp/dd/*kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        if "residual_cost" in r["Kernel_Name"]:
            best[(r["Kernel_Name"].split("(")[0][-28:], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(best.items()):
    if g >= 100000: print("stop=%s %-28s grid %8d  n=%d  min %.1f us  median %.1f us" % (sys.argv[1], k, g, len(v), min(v), sorted(v)[len(v)//2]))
PY
done
cat $O/result.txt
