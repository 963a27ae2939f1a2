# kernel timing experiment (DIAG build only: make EXTRA=-DSRLA_DIAG_STOP): duration of the largest srla_autocorr
# dispatches cut short at successive points (SRLA_MI355X_K3_STOP=11..16), kernels serialised by the counter collection
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/diag_ac
mkdir -p $O
rm -f $O/result.txt
for v in ${VARIANTS:-0 1}; do for st in 11 12 13 14 15 16 0; do
  export SRLA_MI355X_FUSED_FFT=$v
  if [ $st = 0 ]; then unset SRLA_MI355X_K3_STOP; else export SRLA_MI355X_K3_STOP=$st; fi
  rm -rf /tmp/dd; timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d /tmp/dd -o run -- python $R/tools/perf_probe.py 174.8 device 4 ${1:-1} ${2:-0} ${3:-4096} > /tmp/dd.log 2>&1
  python - "$v" "$st" >> $O/result.txt <<'PY'
import csv, glob, sys, collections
best = collections.defaultdict(list)
for path in glob.glob("/tmp/dd/*kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        if "autocorr" in r["Kernel_Name"]:
            best[(r["Kernel_Name"].split("(")[0][-28:], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(best.items()):
    if g >= 100000: print("fused=%s stop=%s %-28s grid %8d  n=%d  min %.1f us  median %.1f us" % (sys.argv[1], sys.argv[2], k, g, len(v), min(v), sorted(v)[len(v)//2]))
PY
done; done
cat $O/result.txt
