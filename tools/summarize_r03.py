#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles_r03.sh (gpurun_out/<tag>/<config>/...) into what is committed
under profiles/<round>/<config>/ and into profiles/pmc_summary.json (read by bench.py for roofline.traffic / valu_util).

    python tools/summarize_r03.py gpurun_out/r03b profiles/r03

Per config: kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --config X`), bench_line.json (the same command
without the profiler) and bench_line_under_rocprof.json, pmc_<pass>.csv (one row per kernel: dispatches, average duration,
average of every counter per dispatch -- tools/summarize_pmc.py).  Derived per kernel in pmc_summary.json:
  hbm_bytes_per_launch  = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH doubled per MI355X_MICROARCH.md: gfx950 tallies
                          64 B per 128-B request), largest dispatch of the kernel (a full job)
  valu_util             = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * duration * 2.4 GHz)   [quad-cycles -> cycles]
  lds_util              = SQ_LDS_IDX_ACTIVE / (256 CUs * duration * 2.4 GHz)
  fp64_inst_frac        = (ADD_F64 + MUL_F64 + FMA_F64) / SQ_INSTS_VALU
  wait_frac             = SQ_WAIT_ANY / SQ_WAVE_CYCLES   (wave cycles parked in s_waitcnt / barrier)
"""
import collections
import csv
import glob
import io
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import summarize_pmc  # noqa: E402

CLOCK_GHZ = 2.4
SIMDS, CUS = 1024, 256


def short(name):
    return name.split("(")[0].replace("void ", "")


def per_kernel(run_dir):
    """{kernel: {"n": dispatches, "dur_ns": avg, "max": {counter: largest per-dispatch sum}, "avg": {counter: avg}}}"""
    cc = glob.glob(os.path.join(run_dir, "*counter_collection.csv"))
    if not cc:
        return {}
    per_disp = collections.defaultdict(lambda: collections.defaultdict(float))
    kern = {}
    for r in csv.DictReader(open(cc[0])):
        per_disp[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        kern[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    dur = collections.defaultdict(list)
    dur_of = {}
    for path in glob.glob(os.path.join(run_dir, "*kernel_trace.csv")):
        for r in csv.DictReader(open(path)):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            dur[short(r["Kernel_Name"])].append(d)
            dur_of[r.get("Dispatch_Id", "")] = d
    out = {}
    for d, k in kern.items():
        e = out.setdefault(k, {"n": 0, "sum": collections.defaultdict(float), "max": collections.defaultdict(float)})
        e["n"] += 1
        for c, v in per_disp[d].items():
            e["sum"][c] += v
            e["max"][c] = max(e["max"][c], v)
    # the utilisation figures are those of the kernel's LONGEST dispatch (a full job), not an average over full and tail jobs
    longest = {}
    for d, k in kern.items():
        if d in dur_of and (k not in longest or dur_of[d] > dur_of[longest[k]]):
            longest[k] = d
    for k, e in out.items():
        if k in longest:
            e["full"] = dict(per_disp[longest[k]])
            e["full_dur_ns"] = float(dur_of[longest[k]])
    for k, e in out.items():
        e["avg"] = {c: v / e["n"] for c, v in e["sum"].items()}
        e["dur_ns"] = sum(dur[k]) / len(dur[k]) if dur.get(k) else 0.0
        e["dur_max_ns"] = max(dur[k]) if dur.get(k) else 0.0
    return out


def last_json_line(path):
    try:
        lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def main(src, dst):
    summary = {}
    for cfg_dir in sorted(glob.glob(os.path.join(src, "*"))):
        if not os.path.isdir(cfg_dir):
            continue
        cfg = os.path.basename(cfg_dir)
        out = os.path.join(dst, cfg)
        os.makedirs(out, exist_ok=True)
        st = os.path.join(cfg_dir, "stats", "run_kernel_stats.csv")
        if os.path.exists(st):
            shutil.copy(st, os.path.join(out, "kernel_stats.csv"))
        for name, log in (("bench_line.json", "bench.log"), ("bench_line_under_rocprof.json", "bench_under_rocprof.log")):
            line = last_json_line(os.path.join(cfg_dir, log))
            if line:
                json.dump(line, open(os.path.join(out, name), "w"), indent=1)
        passes = {}
        for p in ("fetch", "write", "sq1", "sq2", "sq3", "sq4"):
            d = os.path.join(cfg_dir, p)
            if not glob.glob(os.path.join(d, "*counter_collection.csv")):
                continue
            buf = io.StringIO()
            old = sys.stdout
            sys.stdout = buf
            try:
                summarize_pmc.main(d)
            finally:
                sys.stdout = old
            open(os.path.join(out, "pmc_%s.csv" % p), "w").write(buf.getvalue())
            passes[p] = per_kernel(d)
        kernels = set()
        for v in passes.values():
            kernels |= set(v)
        entry = {}
        for k in sorted(kernels):
            e = {}
            f, w = passes.get("fetch", {}).get(k), passes.get("write", {}).get(k)
            if f and w:
                e["fetch_kb_max"] = f["max"].get("FETCH_SIZE", 0.0)
                e["write_kb_max"] = w["max"].get("WRITE_SIZE", 0.0)
                e["hbm_bytes_per_launch"] = (2.0 * e["fetch_kb_max"] + e["write_kb_max"]) * 1024.0
            s1 = passes.get("sq1", {}).get(k)
            if s1 and s1.get("full_dur_ns", 0) > 0:
                cyc = s1["full_dur_ns"] * CLOCK_GHZ
                e["avg_duration_us"] = round(s1["dur_ns"] / 1e3, 2)
                e["full_job_duration_us"] = round(s1["full_dur_ns"] / 1e3, 2)
                e["valu_util"] = round(s1["full"].get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (SIMDS * cyc), 4)
                wc = s1["full"].get("SQ_WAVE_CYCLES", 0.0)
                e["wait_frac"] = round(s1["full"].get("SQ_WAIT_ANY", 0.0) / wc, 4) if wc else None
                e["waves_per_simd"] = round(wc * 4.0 / (SIMDS * cyc), 3) if wc else None
            s3 = passes.get("sq3", {}).get(k)
            if s3 and s3.get("full_dur_ns", 0) > 0:
                cyc = s3["full_dur_ns"] * CLOCK_GHZ
                e["lds_util"] = round(s3["full"].get("SQ_LDS_IDX_ACTIVE", 0.0) / (CUS * cyc), 4)
                e["lds_bank_conflict_frac"] = round(s3["full"].get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, s3["full"].get("SQ_LDS_IDX_ACTIVE", 0.0)), 4)
                e["lds_data_fifo_full_per_idx_active"] = round(s3["full"].get("SQ_LDS_DATA_FIFO_FULL", 0.0) / max(1.0, s3["full"].get("SQ_LDS_IDX_ACTIVE", 0.0)), 4)
            s4 = passes.get("sq4", {}).get(k)
            if s4:
                tot = s4["avg"].get("SQ_INSTS_VALU", 0.0)
                f64 = sum(s4["avg"].get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
                e["fp64_inst_frac"] = round(f64 / tot, 4) if tot else None
                e["int_inst_frac"] = round((s4["avg"].get("SQ_INSTS_VALU_INT32", 0.0) + s4["avg"].get("SQ_INSTS_VALU_INT64", 0.0)) / tot, 4) if tot else None
            if e:
                entry[k] = e
        # bench.py looks the dominant kernel up under its bare name
        rc = [k for k in entry if k.startswith("srla_residual_cost<")]
        if rc:
            big = max(rc, key=lambda k: entry[k].get("hbm_bytes_per_launch", 0.0))
            e = dict(entry[big])
            instants = 4194304          # a full job of the PMC command (174.8 s = two jobs of 4 Mi sample instants)
            e["hbm_bytes_per_instant"] = e.get("hbm_bytes_per_launch", 0.0) / instants
            e["source"] = "%s/%s/pmc_*.csv (rocprofv3 --pmc, separate passes, kernel %s)" % (os.path.relpath(dst, ROOT), cfg, big)
            entry["srla_residual_cost"] = e
        summary[cfg] = entry
    json.dump(summary, open(os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print("wrote", dst, "and pmc_summary.json for", sorted(summary))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
