"""Calls behind one another on ONE handle (DESIGN.md 5 "the buffer from call to call"): the reference keeps one LPC calculator per
encoder, so a later call's history-dependent blocks can inherit what an earlier call left in its FFT buffer -- under `-B 4095 -V 0`
five of six streams differ from a fresh handle's, and so do the odd last blocks of a stream handed over with ComputeBlockSize +
EncodeBlock.  Goldens: the compiled reference, one fresh process per sequence (tools/gen_golden_reuse.py ->
tests/golden/reuse_sequences.json).  CPU: the oracle (which keeps the buffer per handle as the reference does).  GPU: the library."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import helpers
import reuse

GOLD = json.load(open(os.path.join(helpers.GOLDEN, "reuse_sequences.json")))
NAMES = sorted(reuse.SEQUENCES)
HANDLE_HISTORY = 4


def _same(got, gold):
    d = reuse.digest(got)
    return d is None or (d["size"] == gold["size"] and d.get("sha256") == gold.get("sha256"))


def test_goldens_cover_the_sequences_and_some_calls_show_the_carry():
    assert sorted(GOLD) == sorted(NAMES + list(reuse.COUNTED))
    for name in NAMES:
        steps = reuse.SEQUENCES[name][1]
        assert [c["api"] for c in GOLD[name]] == [s["api"] for s in steps]
    assert sum(1 for name in NAMES for c in GOLD[name] if c.get("differs_from_fresh_handle")) >= 10


@pytest.mark.parametrize("name", NAMES + list(reuse.COUNTED))
def test_oracle_reproduces_the_reference_call_after_call(name):
    cli, steps = reuse.SEQUENCES[name] if name in reuse.SEQUENCES else reuse.COUNTED[name]
    for st, gold in zip(steps, GOLD[name]):
        assert "input" not in st or helpers.sha256(reuse.make_input(st["input"])) == gold["input_sha256"]
    outs = reuse.run_on_oracle(cli, steps)
    bad = [i for i, (o, g) in enumerate(zip(outs, GOLD[name])) if not _same(o, g)]
    assert not bad, (name, bad)


def _stats(product, enc):
    import bench
    st = bench.Stats()
    product.lib.SRLAMI355X_GetStats.argtypes = [C.c_void_p, C.POINTER(bench.Stats), C.c_int]
    product.lib.SRLAMI355X_GetStats(enc, C.byref(st), 0)
    return st


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_library_reproduces_the_reference_call_after_call(product, name, capfd):
    cli, steps = reuse.SEQUENCES[name]
    outs, enc = reuse.run_on_library(product, cli, steps)
    try:
        st = _stats(product, enc)
    finally:
        product.destroy(enc)
    bad = [i for i, (o, g) in enumerate(zip(outs, GOLD[name])) if not _same(o, g)]
    assert not bad, (name, bad)
    assert st.num_nonidentical_calls == 0 and st.nonidentical_reasons == 0
    assert "WARNING" not in capfd.readouterr().err


@pytest.mark.gpu
def test_a_word_the_library_does_not_know_is_counted(product, capfd):
    """of a regular call the library keeps the last audible window and the one before; when those rewrite only the buffer's first words
    (a silent window and 100 samples), a clip that reaches beyond them is counted and named, and still decodes to its input"""
    cli, steps = reuse.UNKNOWN_WORDS
    outs, enc = reuse.run_on_library(product, cli, steps)
    try:
        st = _stats(product, enc)
    finally:
        product.destroy(enc)
    assert st.num_nonidentical_calls >= 1 and st.nonidentical_reasons == HANDLE_HISTORY
    assert "NOT guaranteed bit-identical" in capfd.readouterr().err
    assert _same(outs[0], GOLD["unknown_words"][0]) and _same(outs[1], GOLD["unknown_words"][1])
    for stp, o in zip(steps, outs):
        assert np.array_equal(helpers.oracle_decode(o), reuse.make_input(stp["input"]))


@pytest.mark.gpu
def test_a_handle_that_ran_only_regular_calls_is_not_a_fresh_one(product, capfd):
    """the same on a handle whose FIRST call is the stream of several windows (round 4's library took the buffer for the fresh
    handle's zeros there and counted nothing): counted, named, the regular call's bytes the reference's, both streams lossless"""
    cli, steps = reuse.UNKNOWN_WORDS_FRESH
    outs, enc = reuse.run_on_library(product, cli, steps)
    try:
        st = _stats(product, enc)
    finally:
        product.destroy(enc)
    assert st.num_nonidentical_calls >= 1 and st.nonidentical_reasons == HANDLE_HISTORY
    assert "NOT guaranteed bit-identical" in capfd.readouterr().err
    assert _same(outs[0], GOLD["unknown_words_fresh_handle"][0])
    for stp, o in zip(steps, outs):
        assert np.array_equal(helpers.oracle_decode(o), reuse.make_input(stp["input"]))


@pytest.mark.gpu
def test_a_batch_neither_reads_nor_changes_the_handles_buffer(product):
    """EncodeBatch models one fresh handle per stream (the `srla` tool per file): between two calls of a sequence it changes nothing"""
    cli, steps = reuse.SEQUENCES["B4095_V0_odd_streams"]
    from srla_amd import capi
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg)
    try:
        assert product.set_parameter(enc, par) == capi.OK
        for i, (stp, gold) in enumerate(zip(steps, GOLD["B4095_V0_odd_streams"])):
            rc, got = product.encode_whole(enc, reuse.make_input(stp["input"]))
            assert rc == capi.OK and _same(got, gold), i
            if i == 1:
                pcm = reuse.make_input(steps[3]["input"])
                rc, outs, _ = capi.encode_batch(product, enc, [pcm, pcm])
                fresh = helpers.Oracle(2, **cli).encode_whole(pcm)
                assert rc == capi.OK and all(np.array_equal(o, fresh) for o in outs)
    finally:
        product.destroy(enc)


@pytest.mark.gpu
def test_streams_from_device_memory_leave_the_buffer_too(product):
    """SRLAMI355X_EncodeWholeDevice counts as EncodeWhole: its kept windows come out of device memory"""
    import torch
    from srla_amd import capi
    name = "B4096_V0_long_then_odd_blocks"
    cli, steps = reuse.SEQUENCES[name]
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg)
    fn = product.lib.SRLAMI355X_EncodeWholeDevice
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
    try:
        assert product.set_parameter(enc, par) == capi.OK
        for i, (stp, gold) in enumerate(zip(steps, GOLD[name])):
            pcm = reuse.make_input(stp["input"])
            if stp["api"] == "whole":
                d = torch.from_numpy(pcm).cuda()
                torch.cuda.synchronize()
                buf = np.zeros(pcm.size * 4 + 4096, np.uint8)
                out = C.c_uint32(0)
                rc = fn(enc, C.c_void_p(d.data_ptr()), pcm.shape[1], pcm.shape[1], buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(out), None)
                got = buf[:out.value]
                del d
            elif stp["api"] == "size":
                rc, got = product.compute_block_size(enc, pcm)
            else:
                rc, got = product.encode_block(enc, pcm)
            assert rc == capi.OK and _same(got, gold), i
        assert _stats(product, enc).num_nonidentical_calls == 0
    finally:
        product.destroy(enc)
