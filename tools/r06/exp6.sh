#!/bin/bash
# round 6, GPU call 6: both autocorrelation classes of every job in one launch (A/B), the default bench line with its config legs, the 8-rank host budget with both feeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fork.py tests/test_bench_launch.py -m gpu -x -q -k "options or golden or fork or pcm" > $O/parity.out 2>&1; tail -3 $O/parity.out > $O/summary.txt
run() {  # config, label, extra bench args, env...
  c=$1; label=$2; extra=$3; shift 3
  line=$(env "$@" timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs $extra 2>/dev/null | grep '^{' | tail -1)
  echo "$line" > $O/line_${c}_${label}.json
  python - "$c" "$label" "$line" >> $O/summary.txt <<'P'
import json, sys
c, label, line = sys.argv[1:4]
try:
    d = json.loads(line)
    st = d["roofline"]["stages"]
    print("%-3s %-26s value %8.1f resident %8s  stages(ms/job): %s  60s %s 10s %s | %s" % (c, label, d["value"], (d.get("device_resident") or {}).get("value"),
          " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items() if v.get("ms_per_job")),
          (d.get("stream_60s") or {}).get("value"), (d.get("stream_10s") or {}).get("value"), d.get("host_buffers", "")[:90]), flush=True)
except Exception as e:
    print(c, label, "FAILED", e, line[:300], flush=True)
P
}
for rep in 1 2; do
 for c in M C3 C4 C5; do
  run $c pair_small_jobs "" SRLA_MI355X_PAIR_MAX_ITEMS=6144
  run $c pair_every_job "" SRLA_MI355X_PAIR_MAX_ITEMS=100000000
 done
done
echo "--- the host budget of one rank of an 8-GPU node (one pool thread), both feeds, against the 1-GPU default (8 threads)" >> $O/summary.txt
for rep in 1 2; do
 for c in M C5; do
  run $c threads8_planes "--no-extras"
  run $c threads1_planes "--no-extras --pack-threads 1"
  run $c threads1_pcm "--no-extras --pack-threads 1 --feed pcm"
  run $c threads8_pcm "--no-extras --feed pcm"
 done
done
echo "--- the default line (python bench.py --steps 5 --warmup 2)" >> $O/summary.txt
( time timeout 600 python bench.py --steps 5 --warmup 2 > $O/default_line.json 2> $O/default_line.err ) 2>> $O/summary.txt
python - $O/default_line.json >> $O/summary.txt <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("M value", d["value"], "cpu", d["cpu_baseline"]["value"], "int_valu", d["roofline"].get("int_valu"))
    for k, v in d.get("configs", {}).items():
        print(k, v["value"], "ms/call", v["ms_per_step"], "cpu", (v.get("cpu_baseline") or {}).get("value"), "x", v.get("speedup_vs_cpu_1core"), "roof", v["roofline"]["frac"], (v["roofline"].get("fp64") or {}).get("frac"), "ratio", v["compression_ratio"], v["lossless_roundtrip_first_stream"])
except Exception as e:
    print("default line FAILED", e)
P
cat $O/summary.txt
