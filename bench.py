#!/usr/bin/env python3
"""bench.py -- encode throughput of the MI355X SRLA path on BASELINE.json's metric configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config M|C1..C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (one process per GPU).

Workload (config.workload): by default the metric configuration `srla -e -m 4 -B 4096` (CLI defaults -V 1 -L 4 -P 0)
on synthetic 48 kHz / 16-bit stereo PCM (tools/synth, kind "music"), `--seconds` of audio per GPU per step.
`--config C1..C5` selects one of BASELINE.json's five configurations instead (C5: a corpus of 300 s files per GPU,
encoded by one SRLAMI355X_EncodeBatch call per step).

Timed region of `value`: SRLAEncoder_EncodeWhole -- planar int32 samples in ordinary (pageable) host
memory in, the complete .srl stream in ordinary host memory out; staging, H2D, every kernel and the way back are
inside.  K steps between barriers, max over ranks; value = sample instants encoded by all ranks / that time.  A step is
`--calls-per-step` back-to-back calls (default: about 0.5 s of device work, 111 x 600 s at the metric configuration: 20 steps
keep the GPU busy for 10 s without a pause; the CPU baseline leg runs BEFORE the timed region).
Every rank encodes its own streams (windows and files are independent units: no collective on the data path), so
scaling is weak.

Extra objects on the JSON line:
  roofline         every stage of a job (`stages`: srla_stage_in, srla_autocorr, srla_pitch_solve, srla_lpc_solve, srla_residual_cost,
                   srla_price_windows, srla_pack_blocks) priced against the same contract -- algorithmic bytes = 16 B per
                   stereo sample instant x instants per launch, over the stage's average duration per job measured
                   with HIP events attached to the dispatches inside the timed region; peak = 8000 GB/s HBM3E.  `kernel` (and
                   achieved / frac / traffic at the top level) = the stage with the LONGEST measured duration among ALL stages;
                   `dominant_kernel` = the single kernel with the largest total time in the committed rocprofv3 --kernel-trace
                   --stats summary of the same command (profiles/rNN/<config>/kernel_stats.csv), priced the same way;
                   `end_to_end` = the whole path (algorithmic bytes of everything a rank encoded / its wall time).  A stage's
                   `kernels`, traffic, valu_util, lds_util, lds_bank_conflict_frac, wait_frac, fp64_inst_frac come from the committed
                   rocprofv3 PMC summary (profiles/pmc_summary.json): the kernels that RAN in that command, never a hand-kept list.
                   `fp64` = the roof that actually bounds srla_autocorr (SURVEY 8d's honest note): the fp64 operations one job's
                   transforms execute (counted from the configuration: items per FFT-size class x the butterflies of fft.c:71-198 that
                   the pruned device transform runs, the Welch window, the spectrum pass) over the stage's measured time, against
                   the 39.3 TFLOP/s an MI355X issues WITHOUT fused multiply-adds (the reference is C90: no FMA); `int_valu` = the
                   wave-level VALU instructions of one srla_residual_cost launch (committed SQ_INSTS_VALU) x 64 lanes over the
                   stage's measured time, against the rate THIS kernel's instruction mix can issue (round 6: cycles per instruction
                   class measured on the device, tools/probes/valu_rates.hip, weighted with the kernel's ISA histogram,
                   tools/valu_roof.py -> profiles/r06/valu_roof.json: 46.1 T lane-ops/s; the data sheet's 78.6 T, which no instruction
                   of the mix reaches, is carried beside it as peak_datasheet_tera_lane_ops).  `profile_stale`: the committed
                   counters were collected from other device sources than the ones now in srla_amd/csrc (SHA-256 stamp).
  cpu_baseline     the compiled reference (oracle/_ref, "reference") or the oracle ("port") timed on ONE host core on
                   a bounded sample of the same workload (rank 0, N = 1 only); `all_cores`: for context, the same on
                   every usable core at once (one handle per thread).
  device_resident  the same encode with the samples already in HBM and a pinned output buffer
                   (SRLAMI355X_EncodeWholeDevice), median of a few calls outside the timed region -- never `value`.
  stream_60s, stream_10s   one 60 s (SURVEY 8d's own input length) / 10 s stream per SRLAEncoder_EncodeWhole call, pageable
                   host memory to pageable host memory, median of 20 calls outside the timed region -- never `value`.
  alternating_inputs   the headline's calls again, alternating between TWO different streams of the same length (nothing a cache of
                   the previous call could flatter); unequal_corpus: nine files of unequal lengths (120-420 s, fixed seed) in one
                   SRLAMI355X_EncodeBatch call under config 5's flags, each timed call behind one of nine OTHER lengths -- tail jobs of ever new
                   shapes (value_same_corpus_repeated: the same nine back to back).  Beside `value`, never it.
  configs          (the metric configuration's default run only) BASELINE.json's other configurations, C2 .. C5, as bounded legs
                   behind the metric's own: ~3 s of calls each under the same timed-region rule (value, ms_per_step = per call), the
                   stage times priced like the headline's (roofline: frac, fp64, int_valu, end_to_end), compression_ratio, a lossless
                   round trip of the first stream, and cpu_baseline = the compiled reference on ONE core at the same flags on a
                   bounded sample (measured before the timed region; speedup_vs_cpu_1core).  --no-config-legs drops them.
  config.feed      "planes" (the metric: planar int32 through the reference's API) or "pcm" (--feed pcm: the same streams as
                   interleaved 16-bit frames through SRLAMI355X_EncodeBatchPcm, de-interleaved on the device -- DESIGN.md 8).
  per_rank         (N > 1) every rank's own time inside the library per step (min / max), how many ranks locked their input /
                   output in place instead of staging, how many ranks' streams decoded back to their input, pool threads.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# name -> (cli flags of `srla -e`, input description)
CONFIGS = {
    "M":  dict(cli=dict(preset=4, max_block=4096, divisions=1, ltp_order=0), rate=48000, nch=2, kind="music", seconds=600.0),
    "C1": dict(cli=dict(preset=0, max_block=2048, divisions=1, ltp_order=0), rate=44100, nch=1, kind="sine", seconds=10.0),
    "C2": dict(cli=dict(preset=2, max_block=4096, divisions=1, ltp_order=0), rate=48000, nch=2, kind="music", seconds=600.0),
    "C3": dict(cli=dict(preset=4, max_block=4096, divisions=2, ltp_order=0), rate=48000, nch=2, kind="music", seconds=300.0),
    "C4": dict(cli=dict(preset=4, max_block=8192, divisions=2, ltp_order=3), rate=48000, nch=2, kind="music", seconds=300.0),
    "C5": dict(cli=dict(preset=4, max_block=4096, divisions=2, ltp_order=3), rate=48000, nch=2, kind="music", seconds=300.0, files=9),
}


class Stats(C.Structure):
    _fields_ = [("num_windows", C.c_uint64), ("num_candidates", C.c_uint64), ("num_items", C.c_uint64),
                ("num_blocks", C.c_uint64), ("num_raw_blocks", C.c_uint64), ("num_silent_blocks", C.c_uint64),
                ("num_tie_items", C.c_uint64), ("num_odd_items", C.c_uint64), ("analyze_launches", C.c_uint64),
                ("analyze_ms", C.c_double), ("price_ms", C.c_double), ("gather_ms", C.c_double),
                ("h2d_ms", C.c_double), ("history_ms", C.c_double), ("pack_ms", C.c_double), ("total_ms", C.c_double),
                ("analyzed_samples", C.c_uint64), ("autocorr_ms", C.c_double), ("solve_ms", C.c_double),
                ("residual_ms", C.c_double), ("timed_jobs", C.c_uint64),
                ("num_tie_resolved", C.c_uint64), ("num_tie_overrides", C.c_uint64), ("num_restarts", C.c_uint64),
                ("num_inplace_pins", C.c_uint64), ("num_inplace_out_pins", C.c_uint64),
                ("num_nonidentical_calls", C.c_uint64), ("nonidentical_reasons", C.c_uint64), ("num_svr_tie_items", C.c_uint64),
                ("num_history_windows", C.c_uint64), ("pitch_ms", C.c_double), ("num_hybrid_jobs", C.c_uint64)]


def usable_cpus():
    """CPUs this container may actually use: min(visible CPUs, cgroup v2 quota)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(pcm, cli, seconds, rate, bps=16):
    """Single-thread CPU encode of the first `seconds` of the workload (reference if it travelled here)."""
    import numpy as np
    import helpers
    from srla_amd import capi
    kind = "reference" if os.path.exists(helpers.REF_SO) else "port"
    ref = capi.EncoderLib(helpers.REF_SO) if kind == "reference" else None

    def encode(clip):
        if ref is not None:
            return ref.encode(clip, bits_per_sample=bps, sampling_rate=rate, **cli)
        return helpers.Oracle(clip.shape[0], bits_per_sample=bps, sampling_rate=rate, **cli).encode_whole(clip)
    if seconds <= 0:
        # a short probe (it also pages the library in) sizes the sample: two timed runs of about 6 s of CPU work each
        probe = np.ascontiguousarray(pcm[:, :min(pcm.shape[1], 10 * rate)])
        t0 = time.perf_counter()
        encode(probe)
        seconds = max(10.0, 6.0 * probe.shape[1] / (time.perf_counter() - t0) / rate)
    n = min(pcm.shape[1], int(seconds * rate))
    clip = np.ascontiguousarray(pcm[:, :n])
    run = lambda: encode(clip)
    if n <= 20 * rate:
        run()  # warm caches / page in
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        out = run()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # for context: the same clip on every usable core at once, one handle per thread (ctypes releases the GIL)
    all_cores = None
    ncores = usable_cpus()
    if ref is not None and ncores > 1:
        import threading
        short = np.ascontiguousarray(clip[:, :max(10 * rate, n // 4)])
        times = [0.0] * ncores

        def work(i):
            t0 = time.perf_counter()
            ref.encode(short, bits_per_sample=bps, sampling_rate=rate, **cli)
            times[i] = time.perf_counter() - t0
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(ncores)]
        [t.start() for t in th]
        [t.join() for t in th]
        wall = time.perf_counter() - t0
        all_cores = {"value": round(ncores * short.shape[1] / wall / 1e6, 3), "unit": "Msamples/s", "cores": ncores,
                     "sample": "%d threads, one handle and one copy of the first %.0f s each" % (ncores, short.shape[1] / rate)}
    model = "unknown"
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(n / best / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": kind, "cpu_model": model,
            "sample": "first %.0f s of the same workload (%d samples/ch, %d ch), best of 2, %s" %
                      (n / rate, n, clip.shape[0], "AVX2 build of the reference, EncodeWhole in memory" if kind == "reference"
                       else "oracle/srla_oracle.c, scalar C"),
            "bytes": int(out.size), "all_cores": all_cores}


# stage of a job -> (Stats field with its HIP-event time or None, "timed jobs only"?, every kernel the library can launch in the
# stage, as rocprofv3 names them).  Which of them a configuration really launches is read from the committed profile of the same
# command (profiles/pmc_summary.json[config]): a stage is priced with the kernels that ran, never with a list kept by hand.
# tests/test_bench_stages.py fails when a kernel of the library or of a committed profile belongs to no stage.
STAGES = (
    ("srla_stage_in",      None,          True,  ("srla_widen16", "srla_deinterleave", "srla_or_reduce", "srla_mask_to_shift", "srla_chain_commit",
                                                  "__amd_rocclr_fillBuffer")),
    ("srla_autocorr",      "autocorr_ms", True,  ("srla_autocorr<", "srla_autocorr_big", "srla_autocorr_pair")),
    ("srla_pitch_solve",   "pitch_ms",    True,  ("srla_pitch_solve",)),
    ("srla_lpc_solve",     "solve_ms",    True,  ("srla_lpc_errvars", "srla_order_select", "srla_lpc_taps", "srla_lpc_recursion",
                                                  "srla_lpc_quantize", "srla_svr_refine")),
    ("srla_residual_cost", "residual_ms", True,  ("srla_residual_cost<", "srla_residual_cost_big")),
    ("srla_price_windows", "price_ms",    True,  ("srla_price_windows",)),
    # block offsets + assembly + the way out: srla_stream_out where the device stores into the caller's buffer, the runtime's
    # copy kernel where the host issues the copies (it also carries the uploads of pageable input: they are the same dispatches
    # to the profiler)
    ("srla_pack_blocks",   "gather_ms",   True,  ("srla_block_offsets", "srla_pack_blocks", "srla_stream_out", "__amd_rocclr_copyBuffer")),
)


def kernel_stage(name):
    """stage of a kernel as rocprofv3 names it (None: unknown to this table)"""
    bare = name.split("(")[0].replace("void ", "")
    if bare == "srla_residual_cost":              # (alias entry of older summaries)
        return "srla_residual_cost"
    for stage, _, _, prefixes in STAGES:
        if any(bare.startswith(p) for p in prefixes):
            return stage
    return None


def committed_profile(config):
    """(pmc summary entry of the config, path of the committed kernel_stats.csv of the same command or None): newest round first"""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json"))).get(config, {})
    except Exception:
        pmc = {}
    stats = None
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r") and d[1:].isdigit()), reverse=True):
        cand = os.path.join(ROOT, "profiles", rnd, config, "kernel_stats.csv")
        if os.path.exists(cand):
            stats = cand
            break
    return pmc, stats


def dominant_kernel(stats_csv, algo_bytes_per_job, peak):
    """the single kernel with the largest total time in the committed rocprofv3 --kernel-trace --stats summary of this command,
    priced like a stage: algorithmic bytes of one job / its average duration (what the review recomputes by hand)"""
    import csv
    try:
        rows = list(csv.DictReader(open(stats_csv)))
    except Exception:
        return None
    rows = [r for r in rows if kernel_stage(r["Name"]) is not None]
    if not rows:
        return None
    r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    avg_ms = float(r["AverageNs"]) / 1e6
    gbs = algo_bytes_per_job / (avg_ms * 1e-3) / 1e9
    return {"name": r["Name"].split("(")[0].replace("void ", ""), "stage": kernel_stage(r["Name"]), "calls": int(r["Calls"]),
            "avg_launch_ms": round(avg_ms, 4), "share_of_kernel_time": round(float(r["Percentage"]) / 100.0, 4),
            "achieved": round(gbs, 2), "frac": round(gbs / peak, 6),
            "source": os.path.relpath(stats_csv, ROOT) + " (average over full and tail jobs of that run)"}


def next_pow2(v):
    p = 1
    while p < v:
        p *= 2
    return p


def autocorr_item_flops(n, num_lags):
    """fp64 additions + multiplications the device executes for ONE autocorrelation item of n samples (no FMA: the reference is
    C90): Welch window (lpc.c:256-266: two exact index differences, the divisor times a times b, the sample's scaling, the product:
    6 per sample), forward complex transform of nfft/2 points (fft.c:71-136: a radix-4 butterfly is 8 complex additions and 3
    complex multiplications = 34, the closing radix-2 stage 4 per pair), the real-transform symmetry pass + |X|^2 + the inverse
    symmetry pass (fft.c:147-198, lpc.c:357-365: 46 per bin pair), the inverse transform pruned to the outputs that reach the
    first ceil(lags / 2) complex results (srla_amd/csrc/autocorr.hip: fft_regions), one multiplication per lag."""
    nfft = max(2, next_pow2(n))
    m = nfft // 2
    need = (min(num_lags, nfft) + 1) // 2
    flops = 6.0 * n + 46.0 * (m // 2) + 8.0 + min(num_lags, nfft)
    nn, s = m, 1
    while nn > 2:
        flops += 34.0 * (m // 4)                                        # forward stage
        active = (m // 4) if s <= need else (m // 4 // s) * need       # butterflies with q < need
        outs = sum(1 for k in (1, 2, 3) if k * s < need)
        flops += active * (10.0 + 8.0 * outs)                           # pruned inverse stage
        nn //= 4
        s *= 4
    if nn == 2:
        flops += 4.0 * (m // 2) + 2.0 * min(m // 2, need)
    return flops


def autocorr_flops_per_job(cli, instants_per_launch):
    """fp64 operations of one job's srla_autocorr launches at this configuration: windows per job x the candidates of every block
    length of the block-division search (srla_encoder.c:336-389) x 4 variants (L, R, M, S) x the operations of one item.  With the
    long-term predictor every item runs the LTP pass (263 lags) and -- an upper bound: items without a pitch skip it -- the LPC pass."""
    maxb, minb = cli["max_block"], cli["max_block"] >> cli["divisions"]
    window = 4 * maxb
    nodes = window // minb + 1
    order = (0, 8, 16, 32, 64, 128, 255)[cli["preset"]]
    per_window, items = 0.0, 0
    for k in range(1, maxb // minb + 1):
        cands = nodes - k
        n = k * minb
        f = 0.0
        if order > 0:
            f += autocorr_item_flops(n, order + 1)
        if cli["ltp_order"] > 0:
            f += autocorr_item_flops(n, 263)
        per_window += 4.0 * cands * f
        items += 4 * cands
    return per_window * (instants_per_launch / window), items * (instants_per_launch / window)


def roofline_object(st, config, launches, instants_per_launch, algo_bytes, end_to_end_gbs, peak=8000.0, cli=None):
    """The roofline object of the JSON line.  Every stage of a job is priced against the same contract -- SURVEY 8d's 16 B per
    stereo sample instant x the instants one job (= one launch of every analysis kernel) covers, over the stage's average
    duration per job measured with HIP events attached to the dispatches inside the timed region (srla_residual_cost on every
    job, the other stages on one job in four: each start event costs stream time).  `kernel` is the stage with the LONGEST
    measured duration per job among ALL stages; `dominant_kernel` the single kernel with the largest total time in the committed
    rocprofv3 summary of the same command.  Counters (HBM traffic, VALU / LDS utilisation) come from the committed rocprofv3 PMC
    passes (profiles/pmc_summary.json): the traffic of a stage is the HBM bytes of ALL dispatches of the stage's kernels that ran
    in the PMC command, per sample instant of that command, times this run's instants per launch."""
    timed = max(1, st.timed_jobs)
    pmc, stats_csv = committed_profile(config)
    pmc_instants = pmc.get("_pmc_instants") or 2.0 * 4194304.0
    kernels_seen = {k: v for k, v in pmc.items() if not k.startswith("_") and isinstance(v, dict)}
    stages = {}
    for name, field, timed_only, _ in STAGES:
        ents = [(k, v) for k, v in kernels_seen.items() if kernel_stage(k) == name and k != "srla_residual_cost"]
        ms = (getattr(st, field) / (timed if timed_only else launches)) if field else 0.0
        if ms <= 0 and not ents:
            continue
        e = {"ms_per_job": round(ms, 4) if ms > 0 else None, "achieved": None, "frac": None, "traffic": None}
        if ms > 0:
            gbs = algo_bytes / (ms * 1e-3) / 1e9
            e.update(achieved=round(gbs, 2), frac=round(gbs / peak, 6))
        if ents:
            e["kernels"] = sorted(k for k, _ in ents)
        if ents and all("hbm_bytes_total" in v for _, v in ents):
            e["traffic"] = int(sum(v["hbm_bytes_total"] for _, v in ents) / pmc_instants * instants_per_launch)
        elif ents and all("hbm_bytes_per_launch" in v for _, v in ents):
            # (summaries of round 3 and before: the largest dispatch of every kernel = its full-job launch of 4 Mi instants)
            per = [v["hbm_bytes_per_launch"] for _, v in ents]
            e["traffic"] = int((max(per) if name == "srla_residual_cost" else sum(per)) * instants_per_launch / 4194304.0)
        if e["traffic"] is not None:
            e["traffic_over_algorithmic"] = round(e["traffic"] / algo_bytes, 2)
        if ents:
            big = max(ents, key=lambda kv: kv[1].get("full_job_duration_us", kv[1].get("avg_duration_us", 0.0)))[1]
            for key in ("valu_util", "lds_util", "lds_bank_conflict_frac", "wait_frac", "waves_per_simd", "fp64_inst_frac"):
                if big.get(key) is not None:
                    e[key] = big[key]
        stages[name] = e
    measured = [k for k in stages if stages[k]["ms_per_job"]]
    longest = max(measured, key=lambda k: stages[k]["ms_per_job"]) if measured else None
    d = stages.get(longest, {"ms_per_job": 0.0, "achieved": 0.0, "frac": 0.0, "traffic": None})
    all_traffic = [e["traffic"] for e in stages.values() if e.get("traffic") is not None]
    roof = {"bound": "hbm", "achieved": d["achieved"], "peak": peak, "unit": "GB/s", "frac": d["frac"], "traffic": d["traffic"],
            "kernel": longest, "avg_launch_ms": d["ms_per_job"],
            "dominant_kernel": dominant_kernel(stats_csv, algo_bytes, peak) if stats_csv else None,
            "launches": int(st.analyze_launches), "algorithmic_bytes_per_launch": int(algo_bytes),
            "items_per_launch": int(st.num_items / launches),
            "stages": stages,
            "traffic_all_stages": int(sum(all_traffic)) if all_traffic else None,
            # the whole path against the same contract: algorithmic bytes of everything one rank encoded / its wall time
            "end_to_end": {"achieved": round(end_to_end_gbs, 2), "frac": round(end_to_end_gbs / peak, 6)},
            "pmc_source": pmc.get("_source", "profiles/pmc_summary.json[%s] (rocprofv3 --pmc, separate passes)" % config) if pmc else None}
    # the roofs that actually bound the two wide kernels (none of this path's kernels is HBM bound: SURVEY 8d's honest note)
    ac = stages.get("srla_autocorr")
    if cli is not None and ac and ac.get("ms_per_job"):
        flops, items = autocorr_flops_per_job(cli, instants_per_launch)
        tf = flops / (ac["ms_per_job"] * 1e-3) / 1e12
        roof["fp64"] = {"kernel": "srla_autocorr", "flops_per_job": int(flops), "items_per_job": int(items),
                        "achieved_tflops": round(tf, 3), "peak_no_fma_tflops": 39.3, "frac": round(tf / 39.3, 4),
                        "peak_fma_tflops": 78.6,
                        "note": "fp64 additions + multiplications executed (pruned inverse transform), no FMA as the reference is C90; "
                                "stage time measured in flight with HIP events"}
        counted = [v.get("fp64_flops_full_job") for k, v in kernels_seen.items() if kernel_stage(k) == "srla_autocorr" and v.get("fp64_flops_full_job")]
        if counted:
            roof["fp64"]["flops_per_full_job_from_counters"] = int(sum(counted))
    rc = stages.get("srla_residual_cost")
    rc_insts = [v.get("valu_wave_insts_full_job") for k, v in kernels_seen.items() if kernel_stage(k) == "srla_residual_cost" and v.get("valu_wave_insts_full_job")]
    if rc and rc.get("ms_per_job") and rc_insts:
        lane_ops = 64.0 * sum(rc_insts) * instants_per_launch / 4194304.0
        tops = lane_ops / (rc["ms_per_job"] * 1e-3) / 1e12
        # the roof: what THIS kernel's instruction mix can issue -- measured cycles per wave-instruction of every instruction class
        # (tools/probes/valu_rates.hip on an MI355X) weighted with the kernel's own ISA histogram (tools/valu_roof.py); the data
        # sheet's 32 lanes per clock and SIMD (78.6 T lane-ops/s) is reached by no instruction of the mix
        peak, peak_src = 78.6, "data sheet: 1024 SIMDs x 32 lanes x 2.4 GHz (no measured rates committed)"
        try:
            vr = json.load(open(os.path.join(ROOT, "profiles", "r06", "valu_roof.json")))
            ent = vr.get("srla_residual_cost<2, true>")
            if ent:
                peak = float(ent["peak_tera_lane_ops"])
                peak_src = ("profiles/r06/valu_roof.json: %.2f cycles per VALU instruction of srla_residual_cost<2, true>'s mix (static ISA histogram x "
                            "profiles/r06/valu_rates.txt, measured issue rates) -> 64 lanes / that x 1024 SIMDs x 2.4 GHz" % ent["cycles_per_valu_instruction"])
        except Exception:
            pass
        roof["int_valu"] = {"kernel": "srla_residual_cost", "valu_wave_instructions_per_job": int(sum(rc_insts) * instants_per_launch / 4194304.0),
                            "achieved_tera_lane_ops": round(tops, 3), "peak_tera_lane_ops": round(peak, 2), "frac": round(tops / peak, 4),
                            "peak_source": peak_src, "peak_datasheet_tera_lane_ops": 78.6, "frac_of_datasheet": round(tops / 78.6, 4),
                            "note": "committed SQ_INSTS_VALU of a full job's launch x 64 lanes / this run's stage time"}
    # were the committed counters collected from the device sources this library was built from?
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import profiles as _profiles
        now = _profiles.kernel_source_sha256()
        stamp = pmc.get("_kernel_sha256")
        roof["profile_stale"] = (stamp != now)
        roof["profile_kernel_sha256"] = stamp
    except Exception:
        roof["profile_stale"] = None
    return roof


# the other BASELINE configurations on the default line (round 6): a bounded leg each behind the metric's own legs, so that what the
# driver records for the metric configuration comes with figures for C2 .. C5 measured by the same run on the same box, each next
# to the compiled reference on one core at the same flags.  (seconds of audio the reference encodes for its figure: about 4 s of
# CPU work each on the GPU boxes' hosts)
LEG_CONFIGS = ("C2", "C3", "C4", "C5")
LEG_CPU_SECONDS = {"C2": 100.0, "C3": 40.0, "C4": 28.0, "C5": 24.0}
LEG_NOMINAL = {"C2": 7500.0, "C3": 2800.0, "C4": 1800.0, "C5": 2100.0}


def leg_cpu_baseline(pcm, name, rate, bps):
    """the compiled reference (the oracle where it did not travel) on ONE core at the configuration's flags: one timed
    SRLAEncoder_EncodeWhole of the first LEG_CPU_SECONDS of the stream, in memory"""
    import numpy as np
    import helpers
    from srla_amd import capi
    cli = dict(CONFIGS[name]["cli"])
    kind = "reference" if os.path.exists(helpers.REF_SO) else "port"
    n = min(pcm.shape[1], int(LEG_CPU_SECONDS[name] * rate))
    clip = np.ascontiguousarray(pcm[:, :n])
    t0 = time.perf_counter()
    if kind == "reference":
        out = capi.EncoderLib(helpers.REF_SO).encode(clip, bits_per_sample=bps, sampling_rate=rate, **cli)
    else:
        out = helpers.Oracle(clip.shape[0], bits_per_sample=bps, sampling_rate=rate, **cli).encode_whole(clip)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": "first %.0f s of the leg's first stream (%d samples/ch, %d ch), one run" % (n / rate, n, clip.shape[0]), "bytes": int(out.size)}


def leg_windows(files, src_lengths, n):
    """(source stream, first sample) of the `files` windows of n samples a leg cuts out of the run's source streams: the sources
    take turns, a source's windows are spread evenly over what it holds beyond one window (even offsets: whole stereo frames of
    16-bit pairs stay aligned the same way), one window per source starts at 0"""
    nsrc = len(src_lengths)
    per_src = -(-files // nsrc)
    out = []
    for f in range(files):
        k = f % nsrc
        span = src_lengths[k] - n
        assert span >= 0, "a source stream is shorter than the leg's stream"
        off = 0 if per_src == 1 else ((f // nsrc) * (span // (per_src - 1))) & ~1
        out.append((k, off))
    return out


def config_leg(lib, L, name, streams_src, rate, nch, bps, pack_threads, cpu_line, budget_s=3.0):
    """One bounded leg of another BASELINE configuration: the same timed-region rule as the headline (calls of the unchanged API
    back to back between device synchronisations, pageable host memory -> pageable host memory), its stages' HIP-event times
    priced as the headline's are."""
    import numpy as np
    import torch
    import helpers
    from srla_amd import capi
    conf = CONFIGS[name]
    cli = dict(conf["cli"])
    files = conf.get("files", 1)
    n = int(conf["seconds"] * rate) & ~1
    # the leg's streams are cut out of the run's two synthetic streams (seeds 1000 / 4242) at different offsets: nine files of
    # 300 s are nine different 300 s windows -- synthesising nine more would cost the run ten seconds
    pcms = [np.ascontiguousarray(streams_src[k][:, off:off + n]) for k, off in leg_windows(files, [p_.shape[1] for p_ in streams_src], n)]
    cfg, par = capi.cli_setup(nch, bps, rate, **cli)
    enc = lib.create(cfg)
    assert enc and lib.set_parameter(enc, par) == capi.OK
    L.SRLAMI355X_SetPackThreads(enc, pack_threads)
    cap = 2 * pcms[0].size * (bps // 8) + 4096
    outs = [np.zeros(cap, dtype=np.uint8) for _ in range(files)]
    sizes = (C.c_uint32 * files)()
    if files == 1:
        planes = capi.planar_ptrs(pcms[0])

        def call():
            rc = L.SRLAEncoder_EncodeWhole(enc, planes, n, outs[0].ctypes.data_as(C.c_void_p), cap, C.cast(sizes, C.POINTER(C.c_uint32)), None)
            if rc != capi.OK:
                raise SystemExit("SRLAEncoder_EncodeWhole (%s leg) -> %d" % (name, rc))
    else:
        batch = capi.BatchCall(lib, pcms, outs)

        def call():
            rc = batch.run(enc, sizes)
            if rc != capi.OK:
                raise SystemExit("SRLAMI355X_EncodeBatch (%s leg) -> %d" % (name, rc))
    calls = max(3, int(round(budget_s * LEG_NOMINAL[name] * 1e6 / (float(n) * files))))
    for _ in range(3):
        call()
    st = Stats()
    L.SRLAMI355X_GetStats(enc, C.byref(st), 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        call()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.SRLAMI355X_GetStats(enc, C.byref(st), 0)
    value = float(n) * files * calls / dt / 1e6
    launches = max(1, st.analyze_launches)
    instants_per_launch = float(n) * files * calls / launches
    algo_bytes = 8.0 * nch * instants_per_launch
    roof = roofline_object(st, name, launches, instants_per_launch, algo_bytes, 8.0 * nch * float(n) * files * calls / dt / 1e9, cli=cli if nch == 2 else None)
    stream0 = outs[0][:sizes[0]].copy()
    leg = {"value": round(value, 3), "unit": "Msamples/s", "ms_per_step": round(1e3 * dt / calls, 3), "calls": calls,
           "workload": "%s: srla -e -m %d -B %d -V %d -L 4 -P %d; %d x %.0f s per %s call, %d calls back to back, pageable -> pageable" % (
               name, cli["preset"], cli["max_block"], cli["divisions"], cli["ltp_order"], files, n / rate,
               "SRLAEncoder_EncodeWhole" if files == 1 else "SRLAMI355X_EncodeBatch", calls),
           "compression_ratio": round(sum(int(sizes[f]) for f in range(files)) / float(sum(p_.size for p_ in pcms) * (bps // 8)), 6),
           "lossless_roundtrip_first_stream": bool((helpers.oracle_decode(stream0) == pcms[0]).all()),
           "roofline": {"bound": "hbm", "kernel": roof["kernel"], "avg_launch_ms": roof["avg_launch_ms"], "achieved": roof["achieved"], "peak": roof["peak"],
                        "unit": "GB/s", "frac": roof["frac"], "traffic": roof["traffic"],
                        "stage_ms_per_job": {k: v["ms_per_job"] for k, v in roof["stages"].items() if v.get("ms_per_job")},
                        "fp64": {k: roof["fp64"][k] for k in ("achieved_tflops", "peak_no_fma_tflops", "frac")} if roof.get("fp64") else None,
                        "int_valu": {k: roof["int_valu"][k] for k in roof["int_valu"] if k != "note"} if roof.get("int_valu") else None,
                        "end_to_end": roof["end_to_end"], "profile_stale": roof.get("profile_stale")},
           "cpu_baseline": cpu_line}
    if cpu_line is not None:
        leg["speedup_vs_cpu_1core"] = round(value / cpu_line["value"], 2)
    lib.destroy(enc)
    return leg


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one GPU each); default: WORLD_SIZE of the launcher, else 1")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="M", choices=sorted(CONFIGS), help="M: the metric configuration; C1..C5: BASELINE.json's configs")
    ap.add_argument("--calls-per-step", type=int, default=0,
                    help="encode calls per step, back to back (default: as many as make a step about 0.5 s of device work -- 111 x 600 s "
                         "at the metric configuration --, so that 20 steps keep the GPU busy for 10 s and an outside sampler sees it)")
    ap.add_argument("--seconds", type=float, default=None, help="audio per stream (default: the configuration's)")
    ap.add_argument("--files", type=int, default=None, help="streams per GPU per step (> 1: one SRLAMI355X_EncodeBatch call per step)")
    ap.add_argument("--cpu-seconds", type=float, default=0.0,
                    help="audio for the CPU baseline sample (default: sized by a probe for two timed runs of about 6 s of CPU work each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preset", type=int, default=None)
    ap.add_argument("--block", type=int, default=None)
    ap.add_argument("--divisions", type=int, default=None)
    ap.add_argument("--ltp", type=int, default=None)
    ap.add_argument("--bps", type=int, default=16, choices=[8, 16, 24], help="bits per sample of the synthetic input (the metric is quoted at 16)")
    ap.add_argument("--no-numa-pin", action="store_true", help="do not restrict the process to the GPU-local NUMA node")
    ap.add_argument("--pack-threads", type=int, default=0, help="host pool threads (staging copies; default: min(8, usable CPUs / ranks))")
    ap.add_argument("--pinned-io", action="store_true", help="headline with pinned input planes and a pinned output buffer (reported in config.workload)")
    ap.add_argument("--feed", default="planes", choices=["planes", "pcm"],
                    help="planes (default, the metric's timed region): planar int32 through SRLAEncoder_EncodeWhole / SRLAMI355X_EncodeBatch; "
                         "pcm: the same streams as interleaved 16-bit PCM frames -- what a WAV data chunk holds before libs/wav widens it -- "
                         "through SRLAMI355X_EncodeBatchPcm, de-interleaved on the device: no per-sample host work (reported in config.workload)")
    ap.add_argument("--no-extras", action="store_true", help="only the timed steps: no device_resident / stream_60s / stream_10s sections (profiler runs)")
    ap.add_argument("--no-config-legs", action="store_true", help="without the bounded legs of C2 .. C5 behind the metric's own (`configs` on the line)")
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous / reduce only, no GPU work (CPU test of the N-rank flow)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (what torchrun would set up)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SRLA_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies (no GPU of its own, a failed encode) must not leave the others waiting in a barrier for ever
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.1)
        for p in list(alive):
            if p.poll() is None:
                continue
            alive.remove(p)
            if p.returncode != 0 and rc == 0:
                rc = p.returncode
                for q in alive:
                    q.terminate()
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        raise SystemExit(spawn_ranks(args, argv))
    world = int(env_world or "1")
    if args.gpus is not None and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    conf = CONFIGS[args.config]
    cli = dict(conf["cli"])
    for k, v in (("preset", args.preset), ("max_block", args.block), ("divisions", args.divisions), ("ltp_order", args.ltp)):
        if v is not None:
            cli[k] = v
    rate, nch, bps = conf["rate"], conf["nch"], args.bps
    seconds = args.seconds if args.seconds is not None else conf["seconds"]
    files = args.files if args.files is not None else conf.get("files", 1)
    n = int(seconds * rate)
    n -= n % 2
    # a step of about 0.5 s: calls of the same unchanged API, back to back (nominal pace of the configurations on one MI355X)
    nominal = {"M": 6400.0, "C1": 300.0, "C2": 7000.0, "C3": 2300.0, "C4": 1300.0, "C5": 1800.0}[args.config]
    calls = args.calls_per_step or max(1, int(round(0.5 * nominal * 1e6 / max(1.0, float(n) * files))))
    metric = "encode Msamples/s (-m %d -B %d -V %d -P %d, %s %g kHz %d-bit)" % (
        cli["preset"], cli["max_block"], cli["divisions"], cli["ltp_order"], "stereo" if nch == 2 else "%d ch" % nch, rate / 1000.0, bps)

    import numpy as np
    import torch
    dist = None
    shared_gpu = os.environ.get("SRLA_BENCH_SHARED_GPU") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl = RCCL; used only for the timing barrier / max-reduce.  SRLA_BENCH_SHARED_GPU=1 (test hook): ranks may share
        # a device (RCCL refuses two ranks on one device, so the barrier / max-reduce then go through gloo)
        dist.init_process_group("gloo" if (shared_gpu or args.dry_run) else "nccl")

    def finish(line):
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.destroy_process_group()

    base_line = {"metric": metric, "value": None, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                 "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32+f64",
                 "data": "synthetic"}
    if args.dry_run:
        # the N-rank flow without a GPU: every rank "works" for a while, rank 0 reports the max
        dist and dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        dist and dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        base_line.update(ms_per_step=round(1e3 * float(t.item()), 3), dry_run=True, config={"workload": "dry run (no GPU work)"})
        return finish(base_line)

    import helpers
    from srla_amd import capi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")
    if local_rank >= torch.cuda.device_count() and not shared_gpu:
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible)" % (local_rank, torch.cuda.device_count()))
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)

    # one process per GPU, kept on the CPUs of the GPU's own NUMA node (doorbells, pinned buffers, pool threads)
    numa_cpus = None
    if not args.no_numa_pin:
        try:
            pr = torch.cuda.get_device_properties(local_dev)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            cpus = set()
            for part in open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
                numa_cpus = len(cpus)
        except Exception:
            numa_cpus = None
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    L = lib.lib
    L.SRLAMI355X_SetDevice.argtypes = [C.c_int]
    assert L.SRLAMI355X_SetDevice(local_dev) == 0
    L.SRLAMI355X_EncodeWholeDevice.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                               C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
    L.SRLAMI355X_GetStats.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_int]
    L.SRLAMI355X_SetPackThreads.argtypes = [C.c_void_p, C.c_uint32]

    kind = helpers.SINE if conf["kind"] == "sine" else helpers.MUSIC
    pcms = [helpers.synth(kind, 1000 + 97 * rank + f, rate, nch, n, bps) for f in range(files)]
    if args.pinned_io:
        pcms = [torch.from_numpy(p).pin_memory().numpy() for p in pcms]
    cfg, par = capi.cli_setup(nch, bps, rate, **cli)
    enc = lib.create(cfg)
    assert enc and lib.set_parameter(enc, par) == capi.OK
    # the rank's share of the usable CPUs, less two for the HIP runtime's own threads (the calling thread is one of the pool):
    # 16 CPUs -> 8 / 6 / 2 / 1 threads at 1 / 2 / 4 / 8 ranks; below 6 the library stops staging and locks the input in place
    pack_threads = args.pack_threads or max(1, min(8, usable_cpus() // max(1, world) - (2 if world > 1 else 0)))
    L.SRLAMI355X_SetPackThreads(enc, pack_threads)
    cap = 2 * pcms[0].size * (bps // 8) + 4096
    outs = [(torch.empty(cap, dtype=torch.uint8).pin_memory().numpy() if args.pinned_io else np.zeros(cap, dtype=np.uint8))
            for _ in range(files)]
    out_sizes = (C.c_uint32 * files)()
    planes = [capi.planar_ptrs(p) for p in pcms]

    if args.feed == "pcm":
        if bps != 16:
            raise SystemExit("bench.py --feed pcm: 16-bit input only")
        # interleaved little-endian frames [n][nch] int16: the WAV data chunk of the same stream (libs/wav/src/wav.c:848-852 widens
        # it to the planes the reference's API takes)
        frames = [np.ascontiguousarray(p.T.astype(np.int16)) for p in pcms]
        if args.pinned_io:
            frames = [torch.from_numpy(f).pin_memory().numpy() for f in frames]
        pbatch = capi.PcmBatchCall(lib, frames, [n] * files, 2, outs)

        def call():
            rc = pbatch.run(enc, out_sizes)
            if rc != capi.OK:
                raise SystemExit("SRLAMI355X_EncodeBatchPcm -> %d" % rc)
    elif files == 1:
        def call():
            rc = L.SRLAEncoder_EncodeWhole(enc, planes[0], n, outs[0].ctypes.data_as(C.c_void_p), cap, C.cast(out_sizes, C.POINTER(C.c_uint32)), None)
            if rc != capi.OK:
                raise SystemExit("SRLAEncoder_EncodeWhole -> %d" % rc)
    else:
        batch = capi.BatchCall(lib, pcms, outs)

        def call():
            rc = batch.run(enc, out_sizes)
            if rc != capi.OK:
                raise SystemExit("SRLAMI355X_EncodeBatch -> %d" % rc)

    def step():
        for _ in range(calls):
            call()

    # the CPU baseline leg FIRST (rank 0 at N = 1 only): the timed region below is then the last long thing this process does,
    # and an outside GPU-busy sampler does not spend the run looking at a host-only phase
    cpu_line = None
    if not args.no_cpu_baseline and world == 1 and rank == 0:
        cpu_line = cpu_baseline(pcms[0], cli, args.cpu_seconds, rate, bps)
    # ... and the reference at the other configurations' flags (the legs of `configs` below), host-only work as well
    want_legs = world == 1 and files == 1 and args.config == "M" and not args.no_extras and not args.no_config_legs and bps == 16 and args.feed == "planes"
    leg_cpu = {}
    if want_legs and not args.no_cpu_baseline:
        for name in LEG_CONFIGS:
            leg_cpu[name] = leg_cpu_baseline(pcms[0], name, rate, bps)

    for _ in range(args.warmup):
        step()
    st = Stats()
    L.SRLAMI355X_GetStats(enc, C.byref(st), 1)  # reset counters

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own time inside the library per step and how its buffers reached the device (for the N > 1 line)
        L.SRLAMI355X_GetStats(enc, C.byref(st), 0)
        # (every rank checks its own streams: they decode back to its input -- oracle decoder = checker only)
        mine_ok = all(bool((helpers.oracle_decode(outs[f][:out_sizes[f]].copy()) == pcms[f]).all()) for f in range(files))
        mine = torch.tensor([st.total_ms / max(1, args.steps), float(st.num_inplace_pins > 0), float(st.num_inplace_out_pins > 0), float(mine_ok)],
                            dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_ms = [[float(x) for x in r.cpu()] for r in allr]
    L.SRLAMI355X_GetStats(enc, C.byref(st), 0)
    streams = [outs[f][:out_sizes[f]].copy() for f in range(files)]

    line = None
    if rank == 0:
        # sanity inside the bench: the streams decode back to the input (oracle decoder = checker only)
        lossless = all(bool((helpers.oracle_decode(streams[f]) == pcms[f]).all()) for f in range(files))
        total_instants = float(n) * files * calls * args.steps * world
        value = total_instants / elapsed / 1e6
        launches = max(1, st.analyze_launches)          # one launch of srla_residual_cost per job
        timed = max(1, st.timed_jobs)
        instants_per_launch = float(n) * files * calls * args.steps / launches
        algo_bytes = 8.0 * nch * instants_per_launch            # 8 B per channel-sample
        roof = roofline_object(st, args.config, launches, instants_per_launch, algo_bytes,
                               8.0 * nch * total_instants / world / elapsed / 1e9, cli=cli if nch == 2 else None)
        total_out = sum(int(s.size) for s in streams)
        line = dict(base_line)
        line.update({
            "value": round(value, 3), "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "config": {"workload": "%s: srla -e -m %d -B %d -V %d -L 4 -P %d; %d x %.0f s synthetic %d-ch %g kHz/%d-bit (%s) per call, "
                                   "%d call(s) back to back per GPU per step; "
                                   "%s: input in %s host memory -> complete .srl stream(s) in %s host memory" %
                                   (args.config, cli["preset"], cli["max_block"], cli["divisions"], cli["ltp_order"], files, n / rate, nch,
                                    rate / 1000.0, bps, conf["kind"], calls,
                                    "SRLAMI355X_EncodeBatchPcm (--feed pcm: interleaved 16-bit frames instead of planar int32)" if args.feed == "pcm" else
                                    ("SRLAEncoder_EncodeWhole" if files == 1 else "SRLAMI355X_EncodeBatch"),
                                    "pinned" if args.pinned_io else "pageable", "pinned" if args.pinned_io else "pageable"),
                       "feed": args.feed,
                       "samples_per_channel_per_step": n * files * calls, "samples_per_channel_per_call": n * files,
                       "calls_per_step": calls, "streams_per_step": files * calls,
                       "parallelism": "windows / files sharded per GPU, no collective"},
            "compression_ratio": round(total_out / float(sum(p.size for p in pcms) * (bps // 8)), 6),
            "lossless_roundtrip": lossless,
            "channel_samples_per_s_M": round(value * nch, 3),
            "roofline": roof,
            "phase_ms_per_step": {"autocorr": round(st.autocorr_ms / timed * launches / args.steps, 3),
                                  "solve": round(st.solve_ms / timed * launches / args.steps, 3),
                                  "residual_cost": round(st.residual_ms / timed * launches / args.steps, 3), "price": round(st.price_ms / timed * launches / args.steps, 3),
                                  "pack_blocks": round(st.gather_ms / timed * launches / args.steps, 3),
                                  "enqueue_host": round(st.h2d_ms / args.steps, 3), "collect_host": round(st.pack_ms / args.steps, 3),
                                  "total_host": round(st.total_ms / args.steps, 3)},
            "host_cores": os.cpu_count(), "host_cpu_quota": usable_cpus(), "host_pool_threads": pack_threads, "numa_local_cpus": numa_cpus,
            "per_rank": None if rank_ms is None else {
                "encode_ms_per_step_min": round(min(r[0] for r in rank_ms), 3), "encode_ms_per_step_max": round(max(r[0] for r in rank_ms), 3),
                "ranks_input_locked_in_place": int(sum(r[1] for r in rank_ms)), "ranks_output_locked_in_place": int(sum(r[2] for r in rank_ms)),
                "ranks_lossless_roundtrip": int(sum(r[3] for r in rank_ms)),
                "host_pool_threads_per_rank": pack_threads},
            "tie_items": int(st.num_tie_items), "tie_resolved": int(st.num_tie_resolved), "tie_overrides": int(st.num_tie_overrides),
            # how the pageable buffers reached the device: staged through pinned buffers by the pool threads, or -- when the
            # ranks' share of the CPU quota is too small for that -- page-locked in place for the call and read by DMA
            "host_buffers": ("pinned by the caller" if args.pinned_io else "input: %s; output: %s" % (
                ("page-locked in place per call (hipHostRegister), read by DMA" + (
                    "; half of the channels packed to int16 on the way by %d host thread(s)" % pack_threads if st.num_hybrid_jobs else "")) if st.num_inplace_pins else
                "staged through pinned buffers by %d host threads" % pack_threads,
                "page-locked in place per call, written by the device" if st.num_inplace_out_pins else "copied out of pinned staging buffers")),
        })
        if world == 1 and files == 1 and not args.no_extras and args.feed == "planes":
            # the same encode with the samples resident in HBM and a pinned output buffer (what a caller that already holds
            # the samples on the device gets): reported beside `value`, never as `value`
            d_pcm = torch.from_numpy(pcms[0]).cuda()
            out_t = torch.empty(cap, dtype=torch.uint8).pin_memory()
            dsz = C.c_uint32(0)
            reps = max(3, min(args.steps, 7))
            times = []
            for k in range(reps + 1):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rc = L.SRLAMI355X_EncodeWholeDevice(enc, C.c_void_p(d_pcm.data_ptr()), n, n, C.c_void_p(out_t.data_ptr()), cap, C.byref(dsz), None)
                if rc != capi.OK:
                    raise SystemExit("SRLAMI355X_EncodeWholeDevice -> %d" % rc)
                if k:
                    times.append(time.perf_counter() - t1)
            dt = sorted(times)[len(times) // 2]
            line["device_resident"] = {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_call": round(1e3 * dt, 3),
                                       "same_bytes": bool(np.array_equal(out_t.numpy()[:dsz.value], streams[0])),
                                       "note": "SRLAMI355X_EncodeWholeDevice: samples resident in HBM, pinned output buffer; median of %d calls" % reps}
        if world == 1 and files == 1 and not args.no_extras and args.feed == "planes":
            # SURVEY 8d's own input length (60 s; and 10 s) through the same unchanged SRLAEncoder_EncodeWhole, pageable host
            # memory to pageable host memory: short streams cannot hide the pipeline's fill and drain.  Beside `value`, never it.
            for secs in (60, 10):
                m = min(n, secs * rate)
                if m >= n:
                    continue
                clip = np.ascontiguousarray(pcms[0][:, :m])
                cp = capi.planar_ptrs(clip)
                sz = C.c_uint32(0)
                times = []
                for k in range(23):
                    t1 = time.perf_counter()
                    rc = L.SRLAEncoder_EncodeWhole(enc, cp, m, outs[0].ctypes.data_as(C.c_void_p), cap, C.byref(sz), None)
                    if rc != capi.OK:
                        raise SystemExit("SRLAEncoder_EncodeWhole (%d s) -> %d" % (secs, rc))
                    if k >= 3:
                        times.append(time.perf_counter() - t1)
                dt = sorted(times)[len(times) // 2]
                line["stream_%ds" % secs] = {"value": round(m / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_call": round(1e3 * dt, 3),
                                             "lossless_roundtrip": bool((helpers.oracle_decode(outs[0][:sz.value].copy()) == clip).all()),
                                             "note": "one %d s stream per SRLAEncoder_EncodeWhole call, pageable -> pageable; median of %d calls" % (secs, len(times))}
        if world == 1 and files == 1 and not args.no_extras and args.feed == "planes":
            # (a) the headline's calls alternating between two DIFFERENT streams of the same length: whatever the library keeps from
            # the call before (descriptor tables by job shape, host_plan.cpp) is keyed by shape, never by content -- this leg shows it
            other = helpers.synth(kind, 4242, rate, nch, n, bps)
            oplanes = capi.planar_ptrs(other)
            oout = np.zeros(cap, dtype=np.uint8)
            osz = C.c_uint32(0)
            reps = max(4, min(calls, 24))
            reps -= reps % 2
            # (one untimed call on the second stream: its pages and its output buffer's are touched for the first time there)
            rc = L.SRLAEncoder_EncodeWhole(enc, oplanes, n, oout.ctypes.data_as(C.c_void_p), cap, C.byref(osz), None)
            if rc != capi.OK:
                raise SystemExit("SRLAEncoder_EncodeWhole (alternating, warm-up) -> %d" % rc)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(reps):
                if k & 1:
                    rc = L.SRLAEncoder_EncodeWhole(enc, oplanes, n, oout.ctypes.data_as(C.c_void_p), cap, C.byref(osz), None)
                else:
                    rc = L.SRLAEncoder_EncodeWhole(enc, planes[0], n, outs[0].ctypes.data_as(C.c_void_p), cap, C.cast(out_sizes, C.POINTER(C.c_uint32)), None)
                if rc != capi.OK:
                    raise SystemExit("SRLAEncoder_EncodeWhole (alternating) -> %d" % rc)
            dt = time.perf_counter() - t1
            line["alternating_inputs"] = {"value": round(n * reps / dt / 1e6, 3), "unit": "Msamples/s", "calls": reps,
                                          "same_bytes_first_stream": bool(np.array_equal(outs[0][:out_sizes[0]], streams[0])),
                                          "lossless_roundtrip_second_stream": bool((helpers.oracle_decode(oout[:osz.value].copy()) == other).all()),
                                          "note": "%d calls back to back, alternating between two different %d s streams (seeds 1000 / 4242), pageable -> pageable" % (reps, n // rate)}
            del oout
            # (b) a corpus of UNEQUAL files in one SRLAMI355X_EncodeBatch call under config 5's flags: every call plans tail jobs of
            # shapes the equal-length C5 line never sees
            import random
            rnd = random.Random(20260930)
            lens = [int(rnd.uniform(120.0, 420.0) * rate) & ~1 for _ in range(9)]
            ucli = dict(CONFIGS["C5"]["cli"])
            upcms = [helpers.synth(kind, 7000 + f, rate, nch, m_, bps) for f, m_ in enumerate(lens)]
            uouts = [np.zeros(2 * p.size * (bps // 8) + 4096, dtype=np.uint8) for p in upcms]
            ucfg, upar = capi.cli_setup(nch, bps, rate, **ucli)
            uenc = lib.create(ucfg)
            assert uenc and lib.set_parameter(uenc, upar) == capi.OK
            L.SRLAMI355X_SetPackThreads(uenc, pack_threads)
            ubatch = capi.BatchCall(lib, upcms, uouts)
            usz = (C.c_uint32 * len(lens))()
            # a second corpus of other lengths (the same files cut shorter), encoded -- untimed -- before every timed call: a front end's
            # next batch never has the lengths of the last one, so the timed call finds none of its remainder jobs' tables cached
            lens2 = [int(m_ * rnd.uniform(0.55, 0.95)) & ~1 for m_ in lens]
            upcms2 = [np.ascontiguousarray(p_[:, :m2]) for p_, m2 in zip(upcms, lens2)]
            uouts2 = [np.zeros(2 * p_.size * (bps // 8) + 4096, dtype=np.uint8) for p_ in upcms2]
            ubatch2 = capi.BatchCall(lib, upcms2, uouts2)
            usz2 = (C.c_uint32 * len(lens))()
            times, fresh = [], []
            for k in range(4):
                t1 = time.perf_counter()
                rc = ubatch.run(uenc, usz)
                if rc != capi.OK:
                    raise SystemExit("SRLAMI355X_EncodeBatch (unequal corpus) -> %d" % rc)
                if k:
                    times.append(time.perf_counter() - t1)
            for k in range(3):
                rc = ubatch2.run(uenc, usz2)
                if rc != capi.OK:
                    raise SystemExit("SRLAMI355X_EncodeBatch (unequal corpus, second set) -> %d" % rc)
                t1 = time.perf_counter()
                rc = ubatch.run(uenc, usz)
                if rc != capi.OK:
                    raise SystemExit("SRLAMI355X_EncodeBatch (unequal corpus) -> %d" % rc)
                fresh.append(time.perf_counter() - t1)
            dt = sorted(fresh)[len(fresh) // 2]
            dt_rep = sorted(times)[len(times) // 2]
            line["unequal_corpus"] = {"value": round(sum(lens) / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_call": round(1e3 * dt, 3),
                                      "value_same_corpus_repeated": round(sum(lens) / dt_rep / 1e6, 3),
                                      "files": len(lens), "seconds": [round(m_ / rate, 1) for m_ in lens],
                                      "flags": "-m %d -B %d -V %d -P %d" % (ucli["preset"], ucli["max_block"], ucli["divisions"], ucli["ltp_order"]),
                                      "lossless_roundtrip_first_file": bool((helpers.oracle_decode(uouts[0][:usz[0]].copy()) == upcms[0]).all()),
                                      "lossless_roundtrip_second_set_last_file": bool((helpers.oracle_decode(uouts2[-1][:usz2[len(lens) - 1]].copy()) == upcms2[-1]).all()),
                                      "note": "nine files of unequal lengths (120-420 s, fixed seed) in one SRLAMI355X_EncodeBatch call, pageable -> pageable; "
                                              "value: median of 3 calls, each behind a call of nine OTHER lengths (no remainder job finds its tables); "
                                              "value_same_corpus_repeated: median of 3 calls of the same nine files back to back"}
            del upcms2, uouts2
            lib.destroy(uenc)
            del upcms, uouts
            # (c) the other BASELINE configurations, a bounded leg each (LEG_CONFIGS): value, stage times, roofs, the reference beside it
            if want_legs:
                # (the metric's own handle goes first: a second handle's streams share the device's few hardware queues with the first
                # one's, and which stream lands beside which costs the later handle 5 - 10 % -- what the first legs of round 6 showed)
                lib.destroy(enc)
                enc = None
                line["configs"] = {name: config_leg(lib, L, name, [pcms[0], other], rate, nch, bps, pack_threads, leg_cpu.get(name)) for name in LEG_CONFIGS}
            del other
        if cpu_line is not None:                           # rank 0 at N = 1 only; measured before the timed region
            line["cpu_baseline"] = cpu_line
            line["speedup_vs_cpu_1core"] = round(value / cpu_line["value"], 2)
    if enc is not None:
        lib.destroy(enc)
    finish(line)


if __name__ == "__main__":
    main()
