#!/bin/bash
# round 6, GPU call 8: whole suite once with the committed code (runs 7 of 10), a parity sweep beyond the suite's, srla_residual_cost<4> for four wavefronts per SIMD (A/B at C4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp8; mkdir -p $O
export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite.out 2> $O/suite.err; echo "suite rc=$?" > $O/summary.txt; tail -2 $O/suite.out >> $O/summary.txt
ls gpurun_out/abort_* >> $O/summary.txt 2>&1
timeout 600 python tools/gpu_sweep.py 220 601 --mutate --paths > $O/sweep_601.out 2>&1; tail -2 $O/sweep_601.out >> $O/summary.txt
timeout 600 python tools/gpu_sweep.py 100 602 --max-samples=14000000 > $O/sweep_602.out 2>&1; tail -2 $O/sweep_602.out >> $O/summary.txt
for rep in 1 2 3; do
 for so in libsrla_mi355x.so libsrla_rc44.so; do
  line=$(SRLA_PRODUCT_SO=$PWD/srla_amd/$so timeout 300 python bench.py --config C4 --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs --no-extras 2>/dev/null | grep '^{' | tail -1)
  python - "$so" "$line" >> $O/summary.txt <<'P'
import json, sys
so, line = sys.argv[1:3]
try:
    d = json.loads(line); st = d["roofline"]["stages"]
    print("C4 %-22s value %8.1f  stages(ms/job): %s" % (so, d["value"], " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items() if v.get("ms_per_job"))))
except Exception as e:
    print("C4", so, "FAILED", e, line[:200])
P
 done
done
cat $O/summary.txt
