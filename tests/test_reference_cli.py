"""The reference's OWN command line tool -- its option parser, WAV reader and DECODER, compiled from /root/reference by
`make -C oracle ref_cli` into oracle/_ref/srla_on_mi355x -- linked against this repository's encoder library instead
of libs/srla_encoder.  `srla -e` then runs on the MI355X and `srla -d` is the reference's
decoder: the north star's acceptance test ("srla -d decodes the output bit-identically") with no code of ours on the
decoding side.  The binary exists only where the reference was present at build time; the tests skip without it."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from srla_amd import wavio
from test_cli_wav import _write_wav

BIN = os.path.join(helpers.ROOT, "oracle", "_ref", "srla_on_mi355x")
needs_bin = pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/srla_on_mi355x not built (needs /root/reference)")


def _run(args):
    return subprocess.run([BIN] + args, capture_output=True, text=True, cwd=helpers.ROOT)


@needs_bin
def test_the_reference_cli_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pcm = helpers.synth(helpers.MUSIC, 5, 48000, 2, 70000)
    src = str(tmp_path / "a.wav")
    _write_wav(src, pcm, 48000, 16)
    r = _run(["-e", src, str(tmp_path / "a.srl")])
    assert r.returncode != 0
    assert "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


@needs_bin
@pytest.mark.gpu
@pytest.mark.parametrize("opts,cli,n,bps", [
    (["-m", "4", "-B", "4096"], dict(preset=4, max_block=4096, divisions=1), 48000 * 4, 16),
    (["-m", "4", "-B", "4096", "-V", "2", "-P", "3"], dict(preset=4, max_block=4096, divisions=2, ltp_order=3), 48000 * 3 + 1001, 16),
    (["-m", "2", "-B", "4096", "-V", "0"], dict(preset=2, max_block=4096, divisions=0), 100001, 24),
    (["-m", "0", "-B", "2048"], dict(preset=0, max_block=2048, divisions=1), 70000 + 77, 16),
])
def test_reference_cli_encodes_on_the_gpu_and_its_decoder_restores_the_input(tmp_path, opts, cli, n, bps):
    pcm = helpers.synth(helpers.VARIED, 17, 48000, 2, n, bps)
    src, srl, back = str(tmp_path / "in.wav"), str(tmp_path / "out.srl"), str(tmp_path / "back.wav")
    _write_wav(src, pcm, 48000, bps)
    r = _run(["-e"] + opts + [src, srl])
    assert r.returncode == 0, r.stderr
    got = np.fromfile(srl, dtype=np.uint8)
    want = helpers.Oracle(2, bits_per_sample=bps, **cli).encode_whole(pcm)
    assert np.array_equal(got, want)
    r = _run(["-d", srl, back])
    assert r.returncode == 0, r.stderr
    dec, rate, dec_bps = wavio.read_wav(back)
    assert rate == 48000 and dec_bps == bps and np.array_equal(dec, pcm)
