"""Multi-GPU sharding logic (srla_amd/corpus.py) exercised with two gloo processes on the CPU.
The encoder function here is the oracle -- it stands in for a GPU so that the ORCHESTRATION
(assignment, ordered manifest, no data-path collective) is what is under test."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from srla_amd import corpus

CLI = dict(preset=2, max_block=2048, divisions=1)
FILES = [("f%d" % i, (helpers.MUSIC if i % 2 else helpers.VARIED, 200 + i, 6000 + 1500 * i)) for i in range(7)]


def _make(spec):
    kind, seed, n = spec
    return helpers.synth(kind, seed, 48000, 2, n)


def _encode(pcm):
    return helpers.Oracle(2, **CLI).encode_whole(pcm)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    files = [(name, _make(spec)) for name, spec in FILES]
    manifest, streams = corpus.encode_corpus(_encode, files, rank, world)
    q.put((rank, manifest, sorted(streams)))
    dist.barrier()
    dist.destroy_process_group()


def test_assignment_is_balanced_and_deterministic():
    counts = [n for _, (_, _, n) in FILES]
    a = corpus.assign(counts, 2)
    assert a == corpus.assign(counts, 2)
    loads = [sum(c for c, o in zip(counts, a) if o == r) for r in range(2)]
    assert abs(loads[0] - loads[1]) <= max(counts)
    assert corpus.assign(counts, 1) == [0] * len(counts)
    assert sorted(set(corpus.assign([5] * 8, 8))) == list(range(8))


def test_two_ranks_produce_the_single_process_result():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, _ = corpus.encode_corpus(_encode, [(name, _make(spec)) for name, spec in FILES], 0, 1)
    want = [(e["name"], e["bytes"], e["sha256"]) for e in single]
    owned = set()
    for rank, manifest, mine in results:
        assert [(e["name"], e["bytes"], e["sha256"]) for e in manifest] == want   # every rank sees the full, ordered manifest
        assert all(e["owner"] == rank for e in manifest if e["name"] in mine)
        assert not (owned & set(mine))
        owned |= set(mine)
    assert owned == {name for name, _ in FILES}


def test_a_failing_file_is_recorded_and_the_rest_goes_on():
    files = [(name, _make(spec)) for name, spec in FILES[:3]]

    def enc(pcm):
        if pcm.shape[1] == files[1][1].shape[1]:
            raise ValueError("broken file")
        return _encode(pcm)

    manifest, streams = corpus.encode_corpus(enc, files, 0, 1, keep_streams=False)
    assert [bool(e["error"]) for e in manifest] == [False, True, False] and streams == {}
    assert manifest[0]["sha256"] == hashlib.sha256(_encode(files[0][1]).tobytes()).hexdigest()
