#!/bin/bash
# round 6, GPU call 17: bench.py after its last edit: the default line, the launch tests, smoke()
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp17; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python bench.py > $O/default_line.json 2> $O/default_line.err ) 2> $O/default_line.time
python - $O/default_line.json > $O/summary.txt <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default line: M", d["value"], "60s", d["stream_60s"]["value"], "10s", d["stream_10s"]["value"], "resident", d["device_resident"]["value"], "stale", d["roofline"]["profile_stale"], "lossless", d["lossless_roundtrip"])
for k, v in d["configs"].items(): print(" ", k, v["value"], "cpu", v["cpu_baseline"]["value"], "x", v["speedup_vs_cpu_1core"], v["lossless_roundtrip_first_stream"], v["compression_ratio"])
P
timeout 600 python -m pytest tests/test_bench_launch.py tests/test_bench_stages.py -m gpu -q > $O/launch_tests.out 2>&1; tail -2 $O/launch_tests.out >> $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/summary.txt
cat $O/summary.txt $O/default_line.time
