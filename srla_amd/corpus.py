"""Multi-GPU sharding of a corpus of streams (BASELINE config 5: many files across 8 GPUs).

One process per GPU.  Streams (files) and, inside a stream, look-ahead windows are independent
units, so the data path needs NO collective: every rank encodes the files it owns
and only a tiny manifest (name, bytes, SHA-256) is exchanged at the end -- with torch.distributed
when a process group exists (RCCL on the GPU box, gloo in the CPU tests).

The assignment is deterministic and identical on every rank (longest-processing-time greedy on the
sample counts, ties by index), so no coordination is needed to agree on it.
"""
import hashlib

import numpy as np


def assign(sample_counts, world):
    """-> owner rank per file index.  Greedy LPT: biggest file first onto the least loaded rank."""
    order = sorted(range(len(sample_counts)), key=lambda i: (-int(sample_counts[i]), i))
    load = [0] * world
    owner = [0] * len(sample_counts)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += int(sample_counts[i])
    return owner


def encode_corpus(encode_fn, files, rank=0, world=1, group=None, keep_streams=True):
    """files: list of (name, loader) where loader() -> int32 [ch][n] array, or (name, array).
    encode_fn(pcm) -> uint8 array (.srl bytes).  Every rank returns the full manifest
    [{name, owner, samples, bytes, sha256, error}] in file order; `streams` holds this rank's own outputs (keep_streams=False:
    nothing is retained -- a large corpus must not pile up in host memory).  A file that fails (unreadable WAV, encoder error)
    is recorded in the manifest with its error and the rank goes on: every rank must reach the final exchange."""
    counts = []
    for name, src in files:
        counts.append(int(src.shape[1]) if hasattr(src, "shape") else int(getattr(src, "num_samples")))
    owner = assign(counts, world)
    mine = []
    streams = {}
    for i, (name, src) in enumerate(files):
        if owner[i] != rank:
            continue
        try:
            pcm = src if hasattr(src, "shape") else src()
            data = np.ascontiguousarray(encode_fn(pcm), dtype=np.uint8)
        except Exception as e:                      # noqa: BLE001 -- whatever went wrong with this file stays with this file
            mine.append(dict(index=i, name=name, owner=rank, samples=counts[i], bytes=0, sha256="", error="%s: %s" % (type(e).__name__, e)))
            continue
        if keep_streams:
            streams[name] = data
        mine.append(dict(index=i, name=name, owner=rank, samples=counts[i], bytes=int(data.size),
                         sha256=hashlib.sha256(data.tobytes()).hexdigest(), error=""))
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, mine, group=group)
        merged = [e for part in gathered for e in part]
    else:
        merged = mine
    merged.sort(key=lambda e: e["index"])
    return merged, streams


def main_cli(lib, in_dir, out_dir, cli):
    """`python -m srla_amd.cli -e --corpus IN --out OUT` (optionally under torchrun, one rank per GPU): every .wav
    below IN is encoded to OUT/<relative name>.srl by the rank that owns it; rank 0 prints the manifest summary."""
    import ctypes as C
    import os
    import sys
    import time

    from . import wavio

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    lib.lib.SRLAMI355X_SetDevice.argtypes = [C.c_int]
    if lib.lib.SRLAMI355X_SetDevice(local_rank) != 0:
        raise SystemExit("srla_amd.cli: no MI355X for local rank %d (there is no CPU fallback)" % local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")       # only a manifest travels; the data path has no collective
    paths = []
    for base, _dirs, names in sorted(os.walk(in_dir)):
        for nm in sorted(names):
            if nm.lower().endswith(".wav"):
                paths.append(os.path.join(base, nm))
    # sample counts from the file sizes (44-byte canonical header): only the balance depends on it
    class Src:
        def __init__(self, path):
            self.path = path
            self.num_samples = max(1, os.path.getsize(path))
        def __call__(self):
            return self.path

    # One encoder per stream format, kept for the whole corpus (its device buffers and cached job tables are reused from
    # file to file); the next file is read and parsed while the current one is encoded, finished streams are written by
    # a thread of their own -- the encode call itself releases the GIL (ctypes).
    from concurrent.futures import ThreadPoolExecutor
    from . import capi
    encoders = {}
    reader = ThreadPoolExecutor(max_workers=1)
    writer = ThreadPoolExecutor(max_workers=1)
    pending_reads = {}
    writes = []
    my_paths = None

    def prefetch(path):
        if path is not None and path not in pending_reads:
            pending_reads[path] = reader.submit(wavio.read_wav, path)

    def write_out(path, data):
        rel = os.path.relpath(path, in_dir)
        out_path = os.path.join(out_dir, os.path.splitext(rel)[0] + ".srl")
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
        with open(out_path, "wb") as f:
            f.write(memoryview(data))

    def encode_path(path):
        nonlocal my_paths
        if my_paths is None:
            # the files this rank owns, in the order encode_corpus will ask for them
            counts = [max(1, os.path.getsize(p)) for p in paths]
            own = assign(counts, world)
            my_paths = [p for p, o in zip(paths, own) if o == rank]
        prefetch(path)
        pcm, rate, bps = pending_reads.pop(path).result()
        k = my_paths.index(path)
        prefetch(my_paths[k + 1] if k + 1 < len(my_paths) else None)
        key = (pcm.shape[0], bps, rate)
        if key not in encoders:
            cfg, par = capi.cli_setup(pcm.shape[0], bps, rate, **cli)
            enc = lib.create(cfg)
            if lib.set_parameter(enc, par) != capi.OK:
                raise RuntimeError("SetEncodeParameter failed for %s" % path)
            encoders[key] = enc
        rc, data = lib.encode_whole(encoders[key], pcm, cap=2 * os.path.getsize(path))
        if rc != capi.OK:
            raise RuntimeError("EncodeWhole -> %d for %s" % (rc, path))
        writes.append(writer.submit(write_out, path, data))
        return data

    t0 = time.perf_counter()
    try:
        manifest, _ = encode_corpus(encode_path, [(p, Src(p)) for p in paths], rank=rank, world=world, group=group, keep_streams=False)
        for w in writes:
            w.result()
    finally:
        reader.shutdown(wait=True)
        writer.shutdown(wait=True)
        for enc in encoders.values():
            lib.destroy(enc)
    dt = time.perf_counter() - t0
    if rank == 0:
        for e in manifest:
            if e.get("error"):
                print("failed: %s: %s" % (e["name"], e["error"]), file=sys.stderr)
        total_out = sum(e["bytes"] for e in manifest)
        total_in = sum(os.path.getsize(e["name"]) for e in manifest)
        print("finished: %d files, %d -> %d (%6.2f %%) in %.2f s on %d GPU(s)"
              % (len(manifest), total_in, total_out, 100.0 * total_out / max(1, total_in), dt, world))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 1 if any(e.get("error") for e in manifest) else 0
