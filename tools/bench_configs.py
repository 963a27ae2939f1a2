#!/usr/bin/env python3
"""GPU throughput next to the single-core reference for every BASELINE.json configuration (and a few more).

    python tools/bench_configs.py            # prints a markdown table

GPU: python bench.py --no-cpu-baseline with the configuration's flags (300 s of audio per step, 3 steps).
CPU: bench.py's cpu_baseline (the compiled reference where oracle/_ref exists, else the oracle) on 20 s."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [
    ("-m 0 -B 2048 -V 1", ["--preset", "0", "--block", "2048", "--divisions", "1"]),
    ("-m 2 -B 4096 -V 1", ["--preset", "2", "--block", "4096", "--divisions", "1"]),
    ("-m 4 -B 4096 -V 1 (metric)", ["--preset", "4", "--block", "4096", "--divisions", "1"]),
    ("-m 4 -B 4096 -V 2", ["--preset", "4", "--block", "4096", "--divisions", "2"]),
    ("-m 4 -B 8192 -V 2 -P 3", ["--preset", "4", "--block", "8192", "--divisions", "2", "--ltp", "3"]),
    ("-m 4 -B 4096 -V 2 -P 3", ["--preset", "4", "--block", "4096", "--divisions", "2", "--ltp", "3"]),
    ("-m 4 -B 4096 -V 1, 24-bit", ["--preset", "4", "--block", "4096", "--divisions", "1", "--bps", "24"]),
]


def main():
    print("| configuration | MI355X Msamples/s | reference, 1 core | ratio | compression |")
    print("|---|---|---|---|---|")
    for name, flags in CONFIGS:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--seconds", "300",
               "--cpu-seconds", "20"] + flags
        out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()
        line = json.loads([l for l in out if l.startswith("{")][-1])
        cpu = line["cpu_baseline"]
        print("| `%s` | %.0f | %.2f (%s) | %.0fx | %.4f |" % (name, line["value"], cpu["value"], cpu["kind"],
                                                           line["value"] / cpu["value"], line["compression_ratio"]), flush=True)


if __name__ == "__main__":
    main()
