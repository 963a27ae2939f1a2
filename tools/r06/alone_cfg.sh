#!/bin/bash
# stand-alone kernel times (kernels serialised by a counter pass) + LDS / VALU counters of one library build at given flags:
#   tools/r06/alone_cfg.sh OUTNAME lib.so V P B [seconds]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; so=$2; V=$3; P=$4; B=$5; S=${6:-174.8}
mkdir -p $O
tag=$(basename $so .so)_V${V}_P${P}_B${B}
SQ="SQ_WAVES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
SRLA_PRODUCT_SO=$R/$so timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/$tag -o run -- python $R/tools/perf_probe.py $S device 3 $V $P $B > $O/$tag.log 2>&1
python - $O/$tag <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
dur, cnt = {}, collections.defaultdict(dict)
for path in glob.glob(O + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:52], int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for path in glob.glob(O + "/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
groups = collections.defaultdict(list)
for d, (k, g, us) in dur.items():
    groups[(k, g)].append((us, d))
with open(O + "_alone.txt", "w") as f:
    for (k, g), lst in sorted(groups.items()):
        if g < 20000 and "pack" not in k and "pitch" not in k: continue
        lst.sort(); us, d = lst[len(lst) // 2]; c = cnt[d]
        f.write("%-52s grid %8d n=%2d min %7.1f med %7.1f us  valu %.1fM  valu_busy %.2f  lds_active %.1fM conflict %.3f  wait %.2f\n" % (
            k, g, len(lst), lst[0][0], us, c.get("SQ_INSTS_VALU", 0) / 1e6, 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_BUSY_CYCLES", 1)) / 4,
            c.get("SQ_LDS_IDX_ACTIVE", 0) / 1e6, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1)), c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1))))
PY
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete; rm -rf $O/$tag
echo "== $tag"; cat $O/${tag}_alone.txt
