"""Byte-for-byte comparison of the oracle with the compiled reference on randomly chosen
configurations.  Only where the reference can be built (this container); skipped elsewhere --
tests/test_oracle_golden.py covers those machines."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.skipif(not helpers.have_reference(), reason="compiled reference not available")

CLIS = [dict(preset=4, max_block=4096, divisions=1), dict(preset=0, max_block=2048, divisions=1),
        dict(preset=2, max_block=4096, divisions=0), dict(preset=4, max_block=4096, divisions=2, ltp_order=3),
        dict(preset=6, max_block=1024, divisions=1, lookahead_factor=2, ltp_order=1),
        dict(preset=1, max_block=8192, divisions=2, ltp_order=3), dict(preset=5, max_block=2048, divisions=3, lookahead_factor=2)]


@pytest.mark.parametrize("kind", [helpers.VARIED, helpers.MUSIC, helpers.NOISE])
@pytest.mark.parametrize("nch", [1, 2, 3])
def test_streams_identical(kind, nch):
    ref = helpers.reference_encoder()
    for n in (20002, 8500, 4100, 700):
        pcm = helpers.synth(kind, 100 + nch, 48000, nch, n)
        for cli in CLIS:
            want = ref.encode(pcm, **cli)
            got = helpers.Oracle(nch, **cli).encode_whole(pcm)
            if not np.array_equal(got, want) and cli.get("ltp_order", 0):
                # the pitch search reads two never-written words of the reference's work area
                # (lpc.c:1508-1510 with j + 1 = 263, 264): in this long-lived process they hold heap
                # garbage; `srla -e` (a fresh process, zero pages) is the behaviour to match
                want = helpers.reference_encode_fresh(pcm, **cli)
            assert np.array_equal(got, want), (kind, nch, n, cli)
            assert np.array_equal(helpers.oracle_decode(want), pcm)


@pytest.mark.parametrize("kind,nch,n", [(helpers.MUSIC, 1, 4099), (helpers.MUSIC, 2, 20001), (helpers.VARIED, 2, 8501),
                                        (helpers.NOISE, 3, 4099), (helpers.VARIED, 1, 12345)])
def test_odd_lengths_match_a_fresh_reference_process(kind, nch, n):
    """Odd block lengths make the reference history dependent (H4); `srla -e` == a fresh process."""
    pcm = helpers.synth(kind, 100 + nch, 48000, nch, n)
    for cli in (CLIS[0], CLIS[3], CLIS[5]):
        want = helpers.reference_encode_fresh(pcm, **cli)
        got = helpers.Oracle(nch, **cli).encode_whole(pcm)
        assert np.array_equal(got, want), (kind, nch, n, cli)


@pytest.mark.parametrize("bps", [8, 24])
def test_bit_depths(bps):
    ref = helpers.reference_encoder()
    dec = helpers.reference_decoder()
    for kind in (helpers.VARIED, helpers.MUSIC):
        pcm = helpers.synth(kind, 3, 44100, 2, 30000, bps)
        for cli in CLIS[:4]:
            want = ref.encode(pcm, bits_per_sample=bps, sampling_rate=44100, **cli)
            got = helpers.Oracle(2, bits_per_sample=bps, sampling_rate=44100, **cli).encode_whole(pcm)
            if not np.array_equal(got, want):
                # heap-history dependence of the long-lived reference (see test_streams_identical): `srla -e` decides
                want = helpers.reference_encode_fresh(pcm, bits_per_sample=bps, sampling_rate=44100, **cli)
            assert np.array_equal(got, want)
            back, _ = dec.decode(got)
            assert np.array_equal(back, pcm)


def test_block_level_api_matches():
    ref = helpers.reference_encoder()
    from srla_amd import capi
    pcm = helpers.synth(helpers.MUSIC, 9, 48000, 2, 4096)
    cfg, par = capi.cli_setup(2, 16, 48000, preset=4, max_block=4096, divisions=1)
    enc = ref.create(cfg)
    assert ref.set_parameter(enc, par) == capi.OK
    rc, size = ref.compute_block_size(enc, pcm)
    rc2, data = ref.encode_block(enc, pcm)
    ref.destroy(enc)
    o = helpers.Oracle(2, preset=4, max_block=4096, divisions=1)
    assert rc == rc2 == capi.OK
    assert o.compute_block_size(pcm) == size == data.size
    assert np.array_equal(o.encode_block(pcm), data)


def test_compute_block_size_of_more_than_two_channels_prices_the_first_two():
    """srla_encoder.c:1287-1301 adds up the code lengths of channels 0 and 1 only, and :1519-1532 returns that sum: with 3+ channels
    ComputeBlockSize is NOT the size EncodeBlock writes -- it is the price the block division search works with"""
    ref = helpers.reference_encoder()
    from srla_amd import capi
    for nch, kind in ((3, helpers.MUSIC), (5, helpers.VARIED), (8, helpers.NOISE)):
        cli = dict(preset=4, max_block=2048, divisions=1)
        pcm = helpers.synth(kind, 3080 + nch, 48000, nch, 2048)
        cfg, par = capi.cli_setup(nch, 16, 48000, **cli)
        enc = ref.create(cfg)
        assert ref.set_parameter(enc, par) == capi.OK
        rc, size = ref.compute_block_size(enc, pcm)
        rc2, data = ref.encode_block(enc, pcm)
        ref.destroy(enc)
        o = helpers.Oracle(nch, **cli)
        assert rc == rc2 == capi.OK
        assert o.compute_block_size(pcm) == size and size < data.size
        assert np.array_equal(o.encode_block(pcm), data)


SVR = [dict(preset=2, max_block=4096, divisions=1, svr_iterations=1), dict(preset=4, max_block=4096, divisions=1, svr_iterations=5),
       dict(preset=4, max_block=4096, divisions=2, ltp_order=3, svr_iterations=2), dict(preset=6, max_block=2048, divisions=0, svr_iterations=3),
       dict(preset=3, max_block=8192, divisions=1, svr_iterations=10)]


@pytest.mark.parametrize("cli", SVR, ids=["m2_i1", "m4_i5", "m4_V2_P3_i2", "m6_i3", "m3_B8192_i10"])
def test_svr_refinement_matches(cli):
    """--svr-filter-learning-iteration (lpc.c:1036-1136): off by default, the oracle restates it all the same.  Odd lengths too:
    the refinement leaves its residual in the calculator's persistent buffer, which is what an odd block then inherits."""
    for kind, nch, n, bps in ((helpers.MUSIC, 2, 40000, 16), (helpers.VARIED, 2, 30001, 16), (helpers.MUSIC, 1, 9000, 24), (helpers.NOISE, 3, 8192, 16)):
        pcm = helpers.synth(kind, 9, 48000, nch, n, bps)
        want = helpers.reference_encode_fresh(pcm, bits_per_sample=bps, **cli)
        got = helpers.Oracle(nch, bits_per_sample=bps, **cli).encode_whole(pcm)
        assert np.array_equal(got, want), (cli, kind, nch, n, bps)
