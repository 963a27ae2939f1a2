#!/usr/bin/env python3
"""BASELINE config 5 end to end on one GPU: synthesise a corpus of WAV files to a tmpfs directory, encode it with the native
front end (srla_amd/srla_corpus: WAV reader -> SRLAMI355X_EncodeBatch -> .srl writer), check a sample of the outputs against
the oracle byte for byte, report Msamples/s (wall clock of the whole tool: directory scan, file reads, encode, file writes).

    python tools/corpus_run.py [--files 72] [--seconds 300] [-m 4 -B 4096 -V 2 -P 3] [--check 3] [--ranks 1]

With --ranks N > 1 the N rank processes run one after the other on this box's GPU (what a node does side by side)."""
import argparse
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402

TOOL = os.path.join(ROOT, "srla_amd", "srla_corpus")


def write_wav16(path, pcm, rate):
    nch, n = pcm.shape
    body = np.ascontiguousarray(pcm.T).astype("<i2").tobytes()
    fmt = struct.pack("<HHIIHH", 1, nch, rate, rate * nch * 2, nch * 2, 16)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVEfmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(body)))
        f.write(body)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=72)
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("-m", type=int, default=4); ap.add_argument("-B", type=int, default=4096)
    ap.add_argument("-V", type=int, default=2); ap.add_argument("-P", type=int, default=3)
    ap.add_argument("--check", type=int, default=3, help="files compared with the oracle")
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--dir", default="/dev/shm/srla_corpus")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--batch-samples", type=int, default=0, help="passed on to srla_corpus (0: its default)")
    ap.add_argument("--tool-args", default="", help="more arguments for srla_corpus, e.g. '--readers 10 --writers 4'")
    ap.add_argument("--host-deinterleave", action="store_true", help="the tool de-interleaves on the host (SRLAMI355X_EncodeBatchEx) instead of the device")
    a = ap.parse_args()
    shutil.rmtree(a.dir, ignore_errors=True)
    ind, outd = os.path.join(a.dir, "in"), os.path.join(a.dir, "out")
    os.makedirs(ind)
    n = int(a.seconds * 48000)
    t0 = time.perf_counter()
    for i in range(a.files):
        write_wav16(os.path.join(ind, "track%03d.wav" % i), helpers.synth(helpers.MUSIC, 7000 + i, 48000, 2, n), 48000)
    print("synthesised %d files x %.0f s (%.2f GB of WAV) in %.1f s" % (a.files, a.seconds, a.files * n * 4 / 1e9, time.perf_counter() - t0), flush=True)
    flags = ["-e", "-m", str(a.m), "-B", str(a.B), "-V", str(a.V), "-P", str(a.P)] + (["--host-deinterleave"] if a.host_deinterleave else []) \
        + (["--batch-samples", str(a.batch_samples)] if a.batch_samples else []) + a.tool_args.split()
    result = None
    for rep in range(a.repeat):
        shutil.rmtree(outd, ignore_errors=True)
        total_dt, lines = 0.0, []
        for r in range(a.ranks):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(a.ranks), LOCAL_RANK="0")
            t1 = time.perf_counter()
            p = subprocess.run([TOOL] + flags + ["--manifest", os.path.join(a.dir, "manifest%d.json" % r), "--verbose", ind, outd], capture_output=True, text=True, env=env)
            dt = time.perf_counter() - t1
            assert p.returncode == 0, p.stderr[-2000:]
            total_dt = max(total_dt, dt)          # ranks run side by side on a node
            lines.append(p.stdout.strip().splitlines()[-1])
            if rep == 0 and os.environ.get("CORPUS_VERBOSE"):
                print("\n".join(l for l in p.stderr.splitlines() if l.startswith("[srla_corpus]")), flush=True)
        print("run %d: %s" % (rep, " | ".join(lines)), flush=True)
        result = total_dt
    files = []
    for r in range(a.ranks):
        files += json.load(open(os.path.join(a.dir, "manifest%d.json" % r)))["files"]
    assert len(files) == a.files and all(e["error"] == "" for e in files)
    cli = dict(preset=a.m, max_block=a.B, divisions=a.V, ltp_order=a.P)
    checked = 0
    for i in sorted({0, a.files // 2, a.files - 1})[:a.check]:
        pcm = helpers.synth(helpers.MUSIC, 7000 + i, 48000, 2, n)
        want = helpers.Oracle(2, **cli).encode_whole(pcm)
        got = np.fromfile(os.path.join(outd, "track%03d.srl" % i), dtype=np.uint8)
        e = [x for x in files if x["name"] == "track%03d.wav" % i][0]
        assert np.array_equal(got, want) and e["bytes"] == want.size, "track%03d differs from the oracle" % i
        checked += 1
    tin = sum(e["in_bytes"] for e in files); tout = sum(e["bytes"] for e in files)
    print(json.dumps({"config": "srla_corpus -e -m %d -B %d -V %d -P %d" % (a.m, a.B, a.V, a.P), "files": a.files, "seconds_per_file": a.seconds,
                      "ranks": a.ranks, "wall_s": round(result, 3), "Msamples_per_s": round(a.files * n / result / 1e6, 1),
                      "in_bytes": tin, "out_bytes": tout, "ratio": round(tout / tin, 6), "files_checked_against_oracle": checked}), flush=True)
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
