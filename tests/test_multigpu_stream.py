"""One stream sharded over ranks (srla_amd/multigpu.py, SURVEY 8e "contiguous window ranges per GPU").  CPU: two gloo ranks
with the oracle standing in for the GPU, so that the ORCHESTRATION is under test (ranges, the one-integer OR all-reduce,
ordered gather); GPU: the real SRLAMI355X_EncodeWindows ranges concatenated == SRLAEncoder_EncodeWhole."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from srla_amd import capi, multigpu

CLI = dict(preset=2, max_block=2048, divisions=1, lookahead_factor=2)
WINDOW = 4096


def test_ranges_are_contiguous_whole_windows():
    for n in (1, 4095, 4096, 4097, 8192, 8193, 100000, 1_000_001):
        for world in (1, 2, 3, 8):
            r = multigpu.shard_ranges(n, WINDOW, world)
            assert len(r) == world and sum(c for _, c in r) == n
            pos = 0
            for first, count in r:
                if count:
                    assert first == pos and first % WINDOW == 0
                    pos += count
            assert pos == n
            last = [x for x in r if x[1]][-1]
            assert last[0] + last[1] == n and (last[1] >= min(n, 2 * WINDOW - (WINDOW - 1)) or last[0] == 0)
            for first, count in r:
                if count and first + count != n:
                    assert count % WINDOW == 0


def _oracle_range(pcm, shift, is_end):
    assert shift == 0
    return helpers.Oracle(pcm.shape[0], **CLI).encode_whole(pcm)[capi.HEADER_SIZE:]


def _header_from_whole(pcm):
    whole = helpers.Oracle(pcm.shape[0], **CLI).encode_whole(pcm)
    return whole


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pcm = helpers.synth(helpers.MUSIC, 77, 48000, 2, 6 * WINDOW + 1234)
    want = helpers.Oracle(2, **CLI).encode_whole(pcm)
    got = multigpu.encode_stream_sharded(_oracle_range, pcm, lambda s: want[:capi.HEADER_SIZE], WINDOW, rank, world)
    q.put((rank, None if got is None else bool(np.array_equal(got, want))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_whole_stream():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: None}


@pytest.mark.gpu
@pytest.mark.parametrize("cli", [dict(preset=4, max_block=4096, divisions=1), dict(preset=4, max_block=4096, divisions=2, ltp_order=3),
                                 dict(preset=2, max_block=2048, divisions=0)])
@pytest.mark.parametrize("n,shift", [(10 * 16384 + 4000, 0), (7 * 16384 + 4001, 3), (16384, 0), (5000, 2)])
def test_window_ranges_concatenate_to_the_whole_stream(product, cli, n, shift):
    pcm = helpers.synth(helpers.MUSIC, 88, 48000, 2, n)
    pcm = np.ascontiguousarray((pcm >> shift) << shift)
    want = product.encode(pcm, **cli)
    assert np.array_equal(want, helpers.Oracle(2, **cli).encode_whole(pcm))
    for world in (1, 3):
        we = multigpu.WindowEncoder(product, 2, 16, 48000, **cli)
        we.num_samples = n
        try:
            parts = []
            ranges = multigpu.shard_ranges(n, we.window_len, world)
            # every rank's share of the OR from the library (SRLAMI355X_OrMask), combined as the all-reduce would
            mask = 0
            for first, count in ranges:
                if count:
                    mask |= we.or_mask(np.ascontiguousarray(pcm[:, first:first + count]))
            assert mask == int(np.bitwise_or.reduce(pcm.view(np.uint32), axis=None))
            s = multigpu.offset_lshift_of(mask)
            for first, count in ranges:
                if count:
                    parts.append(we.encode_range(np.ascontiguousarray(pcm[:, first:first + count]), s, first + count == n))
            got = np.concatenate([we.header(s)] + parts)
            assert np.array_equal(got, want), (cli, n, shift, world)
        finally:
            we.close()


@pytest.mark.gpu
def test_sharded_encode_takes_the_or_from_the_library_and_reuses_its_buffer(product):
    """encode_stream_sharded with a WindowEncoder (world 1 here; the 2-rank flow above): the range's OR comes from
    SRLAMI355X_OrMask, the output buffer is allocated once per encoder"""
    cli = dict(preset=4, max_block=4096, divisions=1)
    pcm = helpers.synth(helpers.MUSIC, 89, 48000, 2, 9 * 16384 + 77)
    pcm = np.ascontiguousarray((pcm >> 2) << 2)
    we = multigpu.WindowEncoder(product, 2, 16, 48000, **cli)
    try:
        calls = []
        orig = we.or_mask
        we.or_mask = lambda a: calls.append(1) or orig(a)
        got = multigpu.encode_stream_sharded(we.encode_range, pcm, we.header, we.window_len)
        assert calls == [1]
        assert np.array_equal(got, product.encode(pcm, **cli))
        buf = we.buf
        got2 = multigpu.encode_stream_sharded(we.encode_range, pcm, we.header, we.window_len)
        assert we.buf is buf and np.array_equal(got2, got)
    finally:
        we.close()
