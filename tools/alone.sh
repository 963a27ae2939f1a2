#!/bin/bash
# stand-alone kernel times (kernels serialised by a counter pass) of one or more library builds: tools/alone.sh OUTNAME lib1.so [lib2.so ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
for so in "$@"; do
  tag=$(basename $so .so)
  SRLA_PRODUCT_SO=$R/$so timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $O/$tag -o run -- python $R/tools/perf_probe.py 174.8 device 4 1 0 4096 > $O/$tag.log 2>&1
  python - $O/$tag <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
best = collections.defaultdict(list)
for path in glob.glob(O + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        best[(r["Kernel_Name"].split("(")[0][:60], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(O + "_alone.txt", "w") as f:
    for (k, g), v in sorted(best.items()):
        if "autocorr" in k or "residual" in k or "pack_blocks" in k:
            f.write("%-60s grid %8d  n=%3d  min %8.1f us  median %8.1f us\n" % (k, g, len(v), min(v), sorted(v)[len(v)//2]))
PY
  find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
  echo "== $so"; cat $O/${tag}_alone.txt
done
