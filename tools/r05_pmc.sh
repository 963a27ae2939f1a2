#!/bin/bash
# per-kernel counters, kernels serialised: tools/r05_pmc.sh TAG [ENV=VAL ...]  -> gpurun_out/TAG/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/$TAG
mkdir -p $O
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
SQ3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"
CMD="python $R/tools/perf_probe.py ${SECONDS_:-174.8} device 3 ${V:-1} ${P:-0} ${B:-4096}"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/sq1 -o run -- $CMD > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ3 --output-format csv -d $O/sq3 -o run -- $CMD > $O/sq3.log 2>&1
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
python - $O <<'PY'
import csv, sys, collections, glob
O = sys.argv[1]
def load(sub):
    dur = {}
    for r in csv.DictReader(open(glob.glob(O + "/%s/*kernel_trace.csv" % sub)[0])):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cnt = collections.defaultdict(dict)
    for r in csv.DictReader(open(glob.glob(O + "/%s/*counter_collection.csv" % sub)[0])):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    return dur, cnt
out = open(O + "/summary.txt", "w")
for sub in ("sq1", "sq3"):
    dur, cnt = load(sub)
    groups = collections.defaultdict(list)
    for d, (k, g, us) in dur.items(): groups[(k, g)].append((us, d))
    for (k, g), lst in sorted(groups.items()):
        if g < 100000 and "pack" not in k: continue
        lst.sort(); us, d = lst[len(lst) // 2]
        c = cnt[d]
        if sub == "sq1":
            wc = c.get("SQ_WAVE_CYCLES", 0) or 1
            line = "%-44s grid %8d n=%2d median %7.1f us  VALUissue %.3f  wait_any %.3f  wait_inst %.3f  active_any %.3f lds_inst_active %.3f  insts_valu %.3gM busy_cycles %.3g" % (
                k, g, len(lst), us, 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / (c.get("SQ_BUSY_CYCLES", 1) or 1) / 4, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                c.get("SQ_ACTIVE_INST_LDS", 0) / wc, c.get("SQ_INSTS_VALU", 0) / 1e6, c.get("SQ_BUSY_CYCLES", 0))
        else:
            line = "%-44s grid %8d n=%2d median %7.1f us  lds_idx_active %.4gM bank_conflict %.4gM (%.3f) insts_lds %.4gM insts_salu %.4gM vmem_rd %.4gM waves %d" % (
                k, g, len(lst), us, c.get("SQ_LDS_IDX_ACTIVE", 0) / 1e6, c.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6, c.get("SQ_LDS_BANK_CONFLICT", 0) / (c.get("SQ_LDS_IDX_ACTIVE", 1) or 1),
                c.get("SQ_INSTS_LDS", 0) / 1e6, c.get("SQ_INSTS_SALU", 0) / 1e6, c.get("SQ_INSTS_VMEM_RD", 0) / 1e6, c.get("SQ_WAVES", 0))
        out.write(sub + " " + line + "\n")
out.close()
print(open(O + "/summary.txt").read())
PY
