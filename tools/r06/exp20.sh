#!/bin/bash
# round 6, GPU call 20: do the workgroups of srla_autocorr run in lock step?  start delays that differ between the workgroups of a CU (-DSRLA_AC_STAGGER=16 / 64 x 64 cycles x 0..7)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp20; mkdir -p $O
export TMPDIR=/tmp
for so in srla_amd/libsrla_mi355x.so srla_amd/libsrla_stg16.so srla_amd/libsrla_stg64.so; do
  bash tools/r06/alone_cfg.sh r06_exp20 $so 1 0 4096 600 >> $O/summary.txt 2>&1
  bash tools/r06/alone_cfg.sh r06_exp20 $so 2 3 4096 300 >> $O/summary.txt 2>&1
done
for rep in 1 2; do
 for c in M C5; do
  for so in libsrla_mi355x.so libsrla_stg16.so libsrla_stg64.so; do
    line=$(SRLA_PRODUCT_SO=$PWD/srla_amd/$so timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs --no-extras 2>/dev/null | grep '^{' | tail -1)
    python - "$c" "$so" "$line" >> $O/summary.txt <<'P'
import json, sys
c, so, line = sys.argv[1:4]
try:
    d = json.loads(line); st = d["roofline"]["stages"]
    print("%-3s %-22s value %8.1f  stages(ms/job): %s" % (c, so, d["value"], " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items() if v.get("ms_per_job"))))
except Exception as e:
    print(c, so, "FAILED", e, line[:200])
P
  done
 done
done
grep -E "^==|autocorr|^M |^C5 " $O/summary.txt
