#!/usr/bin/env python3
"""bench.py -- encode throughput of the MI355X SRLA path on BASELINE.json's metric configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): `srla -e -m 4 -B 4096` with the CLI defaults -V 1 -L 4 -P 0 on synthetic
48 kHz / 16-bit stereo PCM (tools/synth, kind "music"), `--seconds` of audio per GPU per step.

A step is one complete encode of that batch: the planar int32 samples are already resident in HBM when
the timed region starts (SRLAMI355X_EncodeWholeDevice); the step covers the offset-shift reduction, the
item-analysis / pricing / gather kernels, the D2H of residuals and parameters and the multi-threaded host
bit pack, and ends with the complete .srl stream in host memory.  Every rank encodes its own batch
(frames shard embarrassingly: no collective on the data path), so scaling is weak; value = total sample
instants (per channel) encoded by all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel (srla_analyze_items): algorithmic bytes = 16 B per stereo sample instant
                 x instants per launch, over the launch duration measured with HIP events on
                the launch stream inside the timed region; peak = 8000 GB/s HBM3E.
  cpu_baseline  the compiled reference (oracle/_ref, "reference") or the oracle ("port") timed on ONE host
                core on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


class Stats(C.Structure):
    _fields_ = [("num_windows", C.c_uint64), ("num_candidates", C.c_uint64), ("num_items", C.c_uint64),
                ("num_blocks", C.c_uint64), ("num_raw_blocks", C.c_uint64), ("num_silent_blocks", C.c_uint64),
                ("num_tie_items", C.c_uint64), ("num_odd_items", C.c_uint64), ("analyze_launches", C.c_uint64),
                ("analyze_ms", C.c_double), ("price_ms", C.c_double), ("gather_ms", C.c_double),
                ("h2d_ms", C.c_double), ("d2h_ms", C.c_double), ("pack_ms", C.c_double), ("total_ms", C.c_double),
                ("analyzed_samples", C.c_uint64), ("autocorr_ms", C.c_double), ("solve_ms", C.c_double),
                ("residual_ms", C.c_double), ("timed_jobs", C.c_uint64)]


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
    import helpers
    from srla_amd import capi
    n = min(pcm.shape[1], int(seconds * rate))
    clip = np.ascontiguousarray(pcm[:, :n])
    kind = "port"
    if os.path.exists(helpers.REF_SO):
        ref = capi.EncoderLib(helpers.REF_SO)
        run = lambda: ref.encode(clip, bits_per_sample=bps, sampling_rate=rate, **cli)
        kind = "reference"
    else:
        def run():
            return helpers.Oracle(clip.shape[0], bits_per_sample=bps, sampling_rate=rate, **cli).encode_whole(clip)
    run()  # warm caches / page in
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        out = run()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": round(n / best / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": "first %.0f s of the same workload (%d samples/ch, stereo), best of 2, %s" %
                      (n / rate, n, "AVX2 build of the reference, EncodeWhole in memory" if kind == "reference"
                       else "oracle/srla_oracle.c, scalar C"),
            "bytes": int(out.size)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=600.0, help="audio per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=40.0, help="audio for the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preset", type=int, default=4)
    ap.add_argument("--block", type=int, default=4096)
    ap.add_argument("--divisions", type=int, default=1)
    ap.add_argument("--ltp", type=int, default=0)
    ap.add_argument("--bps", type=int, default=16, choices=[8, 16, 24], help="bits per sample of the synthetic input (the metric is quoted at 16)")
    ap.add_argument("--no-numa-pin", action="store_true", help="do not restrict the process to the GPU-local NUMA node")
    ap.add_argument("--pack-threads", type=int, default=0, help="host threads copying staged blocks when the output is pageable (default: min(8, usable CPUs / (2 * ranks)))")
    ap.add_argument("--pageable-output", action="store_true", help="give the encoder an ordinary (pageable) output buffer: the device then "
                    "writes the blocks into the library's pinned staging buffers and host threads copy them out")
    args = ap.parse_args()

    import torch
    import helpers
    from srla_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")
    # SRLA_BENCH_SHARED_GPU=1 (test hook): ranks may share a device, so that the N > 1 flow can be exercised on a box with
    # one GPU; RCCL refuses two ranks on one device, so the barrier / max-reduce then go through gloo
    shared_gpu = os.environ.get("SRLA_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if shared_gpu else "nccl")   # nccl = RCCL; used only for the timing barrier / max-reduce

    # one process per GPU, kept on the CPUs of the GPU's own NUMA node (doorbells, pinned buffers, pack threads)
    numa_cpus = None
    if not args.no_numa_pin:
        try:
            pr = torch.cuda.get_device_properties(local_rank)
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
    lib.lib.SRLAMI355X_SetDevice.argtypes = [C.c_int]
    assert lib.lib.SRLAMI355X_SetDevice(local_rank) == 0
    lib.lib.SRLAMI355X_EncodeWholeDevice.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                     C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
    lib.lib.SRLAMI355X_GetStats.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_int]

    rate, nch, bps = 48000, 2, args.bps
    n = int(args.seconds * rate)
    n -= n % 2  # even length (odd tails are history dependent in the reference)
    cli = dict(preset=args.preset, max_block=args.block, divisions=args.divisions, ltp_order=args.ltp)
    pcm = helpers.synth(helpers.MUSIC, 1000 + rank, rate, nch, n, bps)
    d_pcm = torch.from_numpy(pcm).cuda()
    torch.cuda.synchronize()

    cfg, par = capi.cli_setup(nch, bps, rate, **cli)
    enc = lib.create(cfg)
    assert enc and lib.set_parameter(enc, par) == capi.OK
    pack_threads = args.pack_threads or max(1, min(8, usable_cpus() // (2 * max(1, world))))
    lib.lib.SRLAMI355X_SetPackThreads.argtypes = [C.c_void_p, C.c_uint32]
    lib.lib.SRLAMI355X_SetPackThreads(enc, pack_threads)
    cap = 2 * pcm.size * 2 + 4096
    if args.pageable_output:
        out = np.zeros(cap, dtype=np.uint8)
    else:
        # pinned host memory: the pack kernel stores every block at its final offset of this buffer
        out_t = torch.empty(cap, dtype=torch.uint8).pin_memory()
        out = out_t.numpy()
    out_size = C.c_uint32(0)

    def step():
        rc = lib.lib.SRLAMI355X_EncodeWholeDevice(enc, C.c_void_p(d_pcm.data_ptr()), n, n,
                                                  out.ctypes.data_as(C.c_void_p), cap, C.byref(out_size), None)
        if rc != capi.OK:
            raise SystemExit("SRLAMI355X_EncodeWholeDevice -> %d" % rc)

    for _ in range(args.warmup):
        step()
    st = Stats()
    lib.lib.SRLAMI355X_GetStats(enc, C.byref(st), 1)  # reset counters

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
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lib.lib.SRLAMI355X_GetStats(enc, C.byref(st), 0)
    stream = out[:out_size.value].copy()

    if rank == 0:
        # sanity inside the bench: the stream decodes back to the input (oracle decoder = checker only)
        diag = bool(os.environ.get("SRLA_MI355X_K3_STOP"))          # kernel timing experiments: no stream is produced
        lossless = None if diag else bool((helpers.oracle_decode(stream) == pcm).all())
        total_instants = float(n) * args.steps * world
        value = total_instants / elapsed / 1e6
        launches = max(1, st.analyze_launches)          # one launch of each analysis kernel per job
        # HIP events recorded by the library on the stream each kernel runs on, inside the timed region
        # the roofline line is about ONE kernel: srla_residual_cost, the longest single launch of a job
        # srla_residual_cost is timed on every job, the other stages on one job in four (each start event costs
        # stream time); srla_autocorr is two launches per job (one per FFT-size class)
        timed = max(1, st.timed_jobs)
        per_job = {"srla_autocorr": st.autocorr_ms / timed, "srla_lpc_recursion+order_select+quantize": st.solve_ms / timed,
                   "srla_residual_cost": st.residual_ms / launches}
        dominant = "srla_residual_cost"
        avg_launch_ms = per_job[dominant]
        instants_per_launch = float(n) * args.steps / launches
        algo_bytes = 16.0 * instants_per_launch            # 8 B per channel-sample, stereo
        achieved = algo_bytes / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = int(json.load(open(tpath))["bytes_per_instant"] * instants_per_launch)
            except Exception:
                traffic = None
        line = {
            "metric": "encode Msamples/s (-m %d -B %d -V %d -P %d, stereo 48 kHz %d-bit)" % (args.preset, args.block, args.divisions, args.ltp, bps),
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32+f64", "data": "synthetic",
            "config": {"workload": "srla -e -m %d -B %d -V %d -L 4 -P %d; %.0f s synthetic stereo 48 kHz/%d-bit (music-like) per GPU per step, "
                                   "samples resident in HBM, complete .srl stream produced in %s host memory" %
                                   (args.preset, args.block, args.divisions, args.ltp, n / rate, bps, "pageable" if args.pageable_output else "pinned"),
                       "samples_per_channel_per_step": n, "parallelism": "windows sharded per GPU, no collective"},
            "compression_ratio": round(out_size.value / float(pcm.size * (bps // 8)), 6),
            "lossless_roundtrip": lossless,
            "channel_samples_per_s_M": round(value * nch, 3),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": traffic,
                         "kernel": dominant, "avg_launch_ms": round(avg_launch_ms, 4),
                         "per_job_stage_ms": {k: round(v, 4) for k, v in per_job.items()},
                         "launches": int(st.analyze_launches), "algorithmic_bytes_per_launch": int(algo_bytes),
                         "items_per_launch": int(st.num_items / launches)},
            "phase_ms_per_step": {"autocorr": round(st.autocorr_ms / timed * launches / args.steps, 3),
                                  "solve": round(st.solve_ms / timed * launches / args.steps, 3),
                                  "residual_cost": round(st.residual_ms / args.steps, 3), "price": round(st.price_ms / timed * launches / args.steps, 3),
                                  "pack_blocks": round(st.gather_ms / timed * launches / args.steps, 3),
                                  "enqueue_host": round(st.h2d_ms / args.steps, 3), "collect_host": round(st.pack_ms / args.steps, 3), "total_host": round(st.total_ms / args.steps, 3)},
            "host_cores": os.cpu_count(), "host_cpu_quota": usable_cpus(), "host_pack_threads": pack_threads, "numa_local_cpus": numa_cpus,
            "tie_items": int(st.num_tie_items),
        }
        # PCIe-inclusive rate of the reference's own entry point (pageable host planes in, same output buffer), best of 3
        # calls outside the timed region; reported beside `value`, never as `value`
        host_t = []
        for _ in range(0 if (diag or world > 1) else 3):     # single-GPU runs only
            t1 = time.perf_counter()
            rc = lib.lib.SRLAEncoder_EncodeWhole(enc, capi.planar_ptrs(pcm), n, out.ctypes.data_as(C.c_void_p), cap, C.byref(out_size), None)
            host_t.append(time.perf_counter() - t1)
            if rc != capi.OK:
                raise SystemExit("SRLAEncoder_EncodeWhole -> %d" % rc)
        if host_t:
            line["host_input"] = {"value": round(n / min(host_t) / 1e6, 3), "unit": "Msamples/s", "ms_per_call": round(1e3 * min(host_t), 3),
                              "same_bytes": bool(np.array_equal(out[:out_size.value], stream)),
                              "note": "SRLAEncoder_EncodeWhole, pageable int32 planes in host memory: staging copies (which pack to int16 and gather the offset-shift OR) + H2D + widening + the same device pipeline"}
        if not args.no_cpu_baseline and world == 1:        # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(pcm, cli, args.cpu_seconds, rate, bps)
            line["speedup_vs_cpu_1core"] = round(value / line["cpu_baseline"]["value"], 2)
        print(json.dumps(line), flush=True)
    lib.destroy(enc)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
