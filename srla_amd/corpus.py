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


def encode_corpus(encode_fn, files, rank=0, world=1, group=None):
    """files: list of (name, loader) where loader() -> int32 [ch][n] array, or (name, array).
    encode_fn(pcm) -> uint8 array (.srl bytes).  Every rank returns the full manifest
    [{name, owner, samples, bytes, sha256}] in file order; `streams` holds this rank's own outputs."""
    counts = []
    for name, src in files:
        counts.append(int(src.shape[1]) if hasattr(src, "shape") else int(getattr(src, "num_samples")))
    owner = assign(counts, world)
    mine = []
    streams = {}
    for i, (name, src) in enumerate(files):
        if owner[i] != rank:
            continue
        pcm = src if hasattr(src, "shape") else src()
        data = np.ascontiguousarray(encode_fn(pcm), dtype=np.uint8)
        streams[name] = data
        mine.append(dict(index=i, name=name, owner=rank, samples=counts[i], bytes=int(data.size),
                         sha256=hashlib.sha256(data.tobytes()).hexdigest()))
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, mine, group=group)
        merged = [e for part in gathered for e in part]
    else:
        merged = mine
    merged.sort(key=lambda e: e["index"])
    return merged, streams
