#!/bin/bash
# round 6, GPU call 2: parity of the new kernel forms, instruction rates, stand-alone times old / new, bench A/B on ONE box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp2; mkdir -p $O
export TMPDIR=/tmp
echo "--- parity (new kernels)" | tee $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or item_records or full_scale or options or block_calls or stream_bytes" > $O/parity.out 2>&1; tail -3 $O/parity.out | tee -a $O/summary.txt
echo "--- valu rates" >> $O/summary.txt
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/valu_rates tools/probes/valu_rates.hip > $O/valu_build.log 2>&1 && /tmp/valu_rates > $O/valu_rates.txt 2>&1
head -3 $O/valu_rates.txt >> $O/summary.txt
for so in srla_amd/libsrla_base.so srla_amd/libsrla_mi355x.so; do
  bash tools/r06/alone_cfg.sh r06_exp2 $so 1 0 4096 >> $O/summary.txt 2>&1
  bash tools/r06/alone_cfg.sh r06_exp2 $so 2 3 4096 >> $O/summary.txt 2>&1
  bash tools/r06/alone_cfg.sh r06_exp2 $so 2 3 8192 >> $O/summary.txt 2>&1
done
echo "--- bench A/B (alternating): base / new" >> $O/summary.txt
for rep in 1 2; do
 for c in M C3 C4 C5; do
  for so in srla_amd/libsrla_base.so srla_amd/libsrla_mi355x.so; do
    line=$(SRLA_PRODUCT_SO=$PWD/$so timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs 2>/dev/null | grep '^{' | tail -1)
    python - "$c" "$so" "$line" >> $O/summary.txt <<'P'
import json, sys
c, so, line = sys.argv[1:4]
try:
    d = json.loads(line)
    st = d["roofline"]["stages"]
    print("%-3s %-28s value %8.1f resident %8s  stages(ms/job): %s  60s %s 10s %s" % (c, so.split('/')[-1], d["value"], (d.get("device_resident") or {}).get("value"),
          " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items() if v.get("ms_per_job")),
          (d.get("stream_60s") or {}).get("value"), (d.get("stream_10s") or {}).get("value")), flush=True)
except Exception as e:
    print(c, so, "FAILED", e, line[:300], flush=True)
P
  done
 done
done
cat $O/summary.txt
