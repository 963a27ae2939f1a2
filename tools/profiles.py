#!/usr/bin/env python3
"""Profile collection and summary for the BASELINE configurations (one tool for every round).

    gpurun -- python tools/profiles.py collect --tag r04 [--configs M,C2,C3,C4,C5]       # on the GPU box
    python tools/profiles.py summarize gpurun_out/r04 profiles/r04                       # here, afterwards

`collect` writes raw rocprofv3 output under gpurun_out/<tag>/<config>/ (scratch).  Per configuration:
  stats/   rocprofv3 --kernel-trace --stats -- python bench.py --config X --steps 3 --warmup 1 --no-cpu-baseline --no-extras
  bench.log   python bench.py --config X --steps 20 --warmup 3 --no-cpu-baseline   (no profiler)
  fetch/ write/ sq1/ .. sq4/   one rocprofv3 --kernel-trace --pmc <counters> pass each (never together with other trace
           domains) over the PMC command: bench.py --config X --steps 1 --warmup 0 --calls-per-step 1 --seconds 524.3
           --files 1 --no-cpu-baseline --no-extras  (six jobs of 4 Mi sample instants)

`summarize` turns that into what is committed: profiles/<round>/<config>/{kernel_stats.csv, bench_line.json,
bench_line_under_rocprof.json, pmc_<pass>.csv} and profiles/pmc_summary.json (read by bench.py for roofline.traffic / valu_util).
Derived per kernel in pmc_summary.json:
  hbm_bytes_total       = sum over ALL the kernel's dispatches in the PMC command of 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes;
                          FETCH doubled per MI355X_MICROARCH.md: gfx950 tallies 64 B per 128-B request); with the command's
                          sample instants (`_pmc_instants` per config) this is the kernel's HBM bytes per instant
  hbm_bytes_per_launch  = the same for the kernel's largest dispatch (a full job)
  valu_util             = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * duration * 2.4 GHz)   [quad-cycles -> cycles]
  waves_per_simd        = SQ_WAVE_CYCLES * 4 / (1024 SIMDs * duration * 2.4 GHz)
  lds_util              = SQ_LDS_IDX_ACTIVE / (256 CUs * duration * 2.4 GHz)
  fp64_inst_frac        = (ADD_F64 + MUL_F64 + FMA_F64) / SQ_INSTS_VALU
  wait_frac             = SQ_WAIT_ANY / SQ_WAVE_CYCLES   (wave cycles parked in s_waitcnt / barrier)
the utilisation figures being those of ONE full job's dispatch -- among the kernel's dispatches with its largest grid the one of
median duration (the first one runs cold) --, not an average over full and tail jobs."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha256():
    """SHA-256 over the device sources a profile describes (every .hip file and the device headers under srla_amd/csrc, in name
    order).  `collect` writes it next to the raw output, `summarize` copies it into profiles/<round>/<config>/ and into
    pmc_summary.json; bench.py compares it with the sources it runs on and says `profile_stale` when they differ."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "srla_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith(".hip") or name in ("device_common.h", "device_layout.h", "huffman_codes.inc"):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


CLOCK_GHZ = 2.4
SIMDS, CUS = 1024, 256
PMC_SECONDS = 524.3                       # six jobs of 4 Mi sample instants at 48 kHz: more than three, so that the bytes leave by host-issued
                                          # copies as in the headline run (a call of at most three jobs uses srla_stream_out throughout)
PASSES = {
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "sq1": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM",
    "sq2": "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM",
    "sq3": "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES",
    "sq4": "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES",
}


# ------------------------------------------------------------------------------------------------------------ collect
def sh(cmd, log, timeout):
    with open(log, "w") as f:
        try:
            return subprocess.call(cmd, shell=True, stdout=f, stderr=subprocess.STDOUT, timeout=timeout, cwd="/tmp",
                                   env=dict(os.environ, TMPDIR="/tmp"))
        except subprocess.TimeoutExpired:
            f.write("\n[profiles.py] timed out after %d s\n" % timeout)
            return -1


def collect(tag, configs, passes, bench_only=False):
    out = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    bench = "python %s/bench.py" % ROOT
    for cfg in configs:
        d = os.path.join(out, cfg)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "kernel_source_sha256.txt"), "w").write(kernel_source_sha256() + "\n")
        if not bench_only:
            sh("rocprofv3 --kernel-trace --stats --output-format csv -d %s/stats -o run -- %s --config %s --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
               % (d, bench, cfg), d + "/bench_under_rocprof.log", 900)
        # (--bench-only: the line again once the summary of THIS round is in profiles/, so that its stage table and dominant_kernel
        # are priced with this round's counters and kernel_stats.csv)
        sh("%s --config %s --steps 20 --warmup 3 --no-cpu-baseline" % (bench, cfg), d + "/bench.log", 900)
        if bench_only:
            continue
        pmc = "%s --config %s --steps 1 --warmup 0 --calls-per-step 1 --seconds %s --files 1 --no-cpu-baseline --no-extras" % (bench, cfg, PMC_SECONDS)
        for p in passes:
            sh("rocprofv3 --kernel-trace --pmc %s --output-format csv -d %s/%s -o run -- %s" % (PASSES[p], d, p, pmc), "%s/%s.log" % (d, p), 600)
        # keep what travels back small: the per-dispatch CSVs only
        for root, _, files in os.walk(d):
            for f in files:
                if f.endswith(".db") or "agent_info" in f:
                    os.remove(os.path.join(root, f))
        print("[profiles.py] %s done" % cfg, flush=True)


# ---------------------------------------------------------------------------------------------------------- summarize
def short(name):
    return name.split("(")[0].replace("void ", "")


def read_pass(run_dir):
    """-> ({dispatch: {counter: value}}, {dispatch: kernel}, {dispatch: duration ns}, {kernel: [durations]}, {dispatch: grid})"""
    cc = glob.glob(os.path.join(run_dir, "*counter_collection.csv"))
    per_disp = collections.defaultdict(lambda: collections.defaultdict(float))
    kern, dur_of, dur, grid_of = {}, {}, collections.defaultdict(list), {}
    if not cc:
        return per_disp, kern, dur_of, dur, grid_of
    for r in csv.DictReader(open(cc[0])):
        per_disp[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        kern[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        grid_of[r["Dispatch_Id"]] = int(r.get("Grid_Size", 0) or 0)
    for path in glob.glob(os.path.join(run_dir, "*kernel_trace.csv")):
        for r in csv.DictReader(open(path)):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            dur[short(r["Kernel_Name"])].append(d)
            dur_of[r.get("Dispatch_Id", "")] = d
    return per_disp, kern, dur_of, dur, grid_of


def per_kernel(run_dir):
    """{kernel: {"n", "sum": {counter: total}, "max": {counter: largest dispatch}, "avg", "full": counters of a full job's dispatch, ...}}

    "a full job's dispatch" = among the kernel's dispatches with its largest grid, the one of MEDIAN duration: the longest one is
    the first (cold code, cold TLB: 285 vs 187 us for srla_residual_cost<2> at M) and would misstate occupancy and VALU share."""
    per_disp, kern, dur_of, dur, grid_of = read_pass(run_dir)
    out = {}
    for d, k in kern.items():
        e = out.setdefault(k, {"n": 0, "sum": collections.defaultdict(float), "max": collections.defaultdict(float)})
        e["n"] += 1
        for c, v in per_disp[d].items():
            e["sum"][c] += v
            e["max"][c] = max(e["max"][c], v)
    full_jobs = collections.defaultdict(list)
    for d, k in kern.items():
        if d in dur_of:
            full_jobs[k].append(d)
    longest = {}
    for k, ds in full_jobs.items():
        top = max(grid_of.get(d, 0) for d in ds)
        ds = sorted((d for d in ds if grid_of.get(d, 0) == top), key=lambda d: dur_of[d])
        longest[k] = ds[len(ds) // 2]
    for k, e in out.items():
        if k in longest:
            e["full"] = dict(per_disp[longest[k]])
            e["full_dur_ns"] = float(dur_of[longest[k]])
        e["avg"] = {c: v / e["n"] for c, v in e["sum"].items()}
        e["dur_ns"] = sum(dur[k]) / len(dur[k]) if dur.get(k) else 0.0
    return out


def pass_table(run_dir, path):
    """one row per kernel: dispatch count, average duration and the average of every counter per dispatch"""
    pk = per_kernel(run_dir)
    counters = sorted({c for v in pk.values() for c in v["sum"]})
    with open(path, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "avg_duration_ns"] + ["%s_per_dispatch" % c for c in counters])
        for k in sorted(pk, key=lambda k: -pk[k]["dur_ns"] * pk[k]["n"]):
            w.writerow([k, pk[k]["n"], int(pk[k]["dur_ns"])] + [int(pk[k]["avg"].get(c, 0.0)) for c in counters])
    return pk


def last_json_line(path):
    try:
        lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def summarize(src, dst):
    spath = os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_summary.json")
    try:
        summary = json.load(open(spath))
    except Exception:
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
        stamp = None
        if os.path.exists(os.path.join(cfg_dir, "kernel_source_sha256.txt")):
            shutil.copy(os.path.join(cfg_dir, "kernel_source_sha256.txt"), os.path.join(out, "kernel_source_sha256.txt"))
            stamp = open(os.path.join(cfg_dir, "kernel_source_sha256.txt")).read().strip()
        for name, log in (("bench_line.json", "bench.log"), ("bench_line_under_rocprof.json", "bench_under_rocprof.log")):
            line = last_json_line(os.path.join(cfg_dir, log))
            if line:
                json.dump(line, open(os.path.join(out, name), "w"), indent=1)
        passes = {}
        for p in PASSES:
            d = os.path.join(cfg_dir, p)
            if glob.glob(os.path.join(d, "*counter_collection.csv")):
                passes[p] = pass_table(d, os.path.join(out, "pmc_%s.csv" % p))
        if not passes:
            continue
        kernels = set()
        for v in passes.values():
            kernels |= set(v)
        entry = {}
        for k in sorted(kernels):
            e = {}
            f, w = passes.get("fetch", {}).get(k), passes.get("write", {}).get(k)
            if f and w:
                e["dispatches"] = f["n"]
                e["fetch_kb_max"] = f["max"].get("FETCH_SIZE", 0.0)
                e["write_kb_max"] = w["max"].get("WRITE_SIZE", 0.0)
                e["hbm_bytes_per_launch"] = (2.0 * e["fetch_kb_max"] + e["write_kb_max"]) * 1024.0
                e["hbm_bytes_total"] = (2.0 * f["sum"].get("FETCH_SIZE", 0.0) + w["sum"].get("WRITE_SIZE", 0.0)) * 1024.0
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
                idx = max(1.0, s3["full"].get("SQ_LDS_IDX_ACTIVE", 0.0))
                e["lds_util"] = round(s3["full"].get("SQ_LDS_IDX_ACTIVE", 0.0) / (CUS * cyc), 4)
                e["lds_bank_conflict_frac"] = round(s3["full"].get("SQ_LDS_BANK_CONFLICT", 0.0) / idx, 4)
                e["lds_data_fifo_full_per_idx_active"] = round(s3["full"].get("SQ_LDS_DATA_FIFO_FULL", 0.0) / idx, 4)
            s4 = passes.get("sq4", {}).get(k)
            if s4:
                tot = s4["avg"].get("SQ_INSTS_VALU", 0.0)
                f64 = sum(s4["avg"].get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
                e["fp64_inst_frac"] = round(f64 / tot, 4) if tot else None
                e["int_inst_frac"] = round((s4["avg"].get("SQ_INSTS_VALU_INT32", 0.0) + s4["avg"].get("SQ_INSTS_VALU_INT64", 0.0)) / tot, 4) if tot else None
                # of ONE full job's dispatch: wave-level VALU instructions, and the fp64 flops they carry (64 lanes; an FMA counts two)
                if s4.get("full"):
                    fj = s4["full"]
                    e["valu_wave_insts_full_job"] = fj.get("SQ_INSTS_VALU", 0.0)
                    e["fp64_flops_full_job"] = 64.0 * (fj.get("SQ_INSTS_VALU_ADD_F64", 0.0) + fj.get("SQ_INSTS_VALU_MUL_F64", 0.0) + 2.0 * fj.get("SQ_INSTS_VALU_FMA_F64", 0.0))
            if e:
                entry[k] = e
        # the PMC command's sample instants (its bench line says what one step was)
        line = None
        for p in PASSES:
            line = line or last_json_line(os.path.join(cfg_dir, p + ".log"))
        entry["_pmc_instants"] = float(line["config"]["samples_per_channel_per_step"]) if line else None
        entry["_source"] = "%s/%s/pmc_*.csv (rocprofv3 --pmc, separate passes)" % (os.path.relpath(dst, ROOT), cfg)
        entry["_kernel_sha256"] = stamp
        summary[cfg] = entry
    json.dump(summary, open(spath, "w"), indent=1, sort_keys=True)
    print("wrote", dst, "and", os.path.relpath(spath, ROOT), "for", sorted(summary))


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "collect":
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument("cmd")
        ap.add_argument("--tag", default="r04")
        ap.add_argument("--configs", default="M,C2,C3,C4,C5")
        ap.add_argument("--passes", default=",".join(PASSES))
        ap.add_argument("--bench-only", action="store_true")
        a = ap.parse_args()
        collect(a.tag, a.configs.split(","), a.passes.split(","), a.bench_only)
    elif len(sys.argv) == 4 and sys.argv[1] == "summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
