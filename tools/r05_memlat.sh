#!/bin/bash
# memory-side counters of the wide kernels (kernels serialised): tools/r05_memlat.sh TAG [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/$TAG
mkdir -p $O
CMD="python $R/tools/perf_probe.py 174.8 device 3 ${V:-1} ${P:-0} ${B:-4096}"
P1="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum"
P2="TCC_HIT_sum TCC_MISS_sum"
P3="TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum"
P4="SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1));
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o run -- $CMD > $O/p$i.log 2>&1
done
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
python - $O <<'PY'
import csv, sys, collections, glob
O = sys.argv[1]
for sub in ("p1", "p2", "p3", "p4"):
    try:
        dur = {}
        for r in csv.DictReader(open(glob.glob(O + "/%s/*kernel_trace.csv" % sub)[0])):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        cnt = collections.defaultdict(dict)
        for r in csv.DictReader(open(glob.glob(O + "/%s/*counter_collection.csv" % sub)[0])):
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        groups = collections.defaultdict(list)
        for d, (k, g, us) in dur.items(): groups[(k, g)].append((us, d))
        for (k, g), lst in sorted(groups.items()):
            if g < 500000: continue
            lst.sort(); us, d = lst[len(lst) // 2]
            print(sub, "%-44s %7.1f us " % (k, us), " ".join("%s=%.4g" % (a.replace("_sum", ""), b) for a, b in sorted(cnt[d].items())))
    except Exception as e:
        print(sub, "failed", e)
PY
