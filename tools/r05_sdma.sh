#!/bin/bash
# Do the host-issued copies run on the SDMA engines or as blit kernels on CUs?  (review item 4b)  tools/r05_sdma.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sdma
mkdir -p $O
rm -f $O/result.txt
run() {
  tag=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    rm -rf $O/$tag
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o run -- python $R/bench.py --steps 3 --warmup 1 --calls-per-step 22 --no-cpu-baseline --no-extras > $O/$tag.log 2>&1
    python - "$tag" "$O" "$*" <<'PY'
import csv, json, sys
tag, O, env = sys.argv[1:4]
val = None
for l in open("%s/%s.log" % (O, tag)):
    if l.startswith("{"): val = json.loads(l)["value"]
rows = list(csv.DictReader(open("%s/%s/run_kernel_stats.csv" % (O, tag))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
cp = [r for r in rows if "copyBuffer" in r["Name"]]
print("%-28s value %8s  copyBuffer: %s calls, %.1f ms = %.1f %% of kernel time" % (env or "(default)", val, sum(int(r["Calls"]) for r in cp), sum(float(r["TotalDurationNs"]) for r in cp) / 1e6, 100.0 * sum(float(r["TotalDurationNs"]) for r in cp) / tot if tot else 0))
PY
  ) >> $O/result.txt 2>&1
}
run default
run sdma1 HSA_ENABLE_SDMA=1
run blitsize0 GPU_FORCE_BLIT_COPY_SIZE=0
run engine GPU_BLIT_ENGINE_TYPE=0
run wg16 DEBUG_CLR_LIMIT_BLIT_WG=16
run wg64 DEBUG_CLR_LIMIT_BLIT_WG=64
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete; find $O -name '*kernel_trace.csv' -delete
cat $O/result.txt
