"""`srla_amd/srla_corpus`, the native corpus front end (tools/srla_codec/srla_codec.c:75-158 for many files,
libs/wav/src/wav.c reader): WAV files of mixed formats in, one .srl per file out, each byte-equal to the oracle's stream;
sharded deterministically over ranks without communication."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import helpers
from test_cli_wav import _write_wav

TOOL = os.path.join(helpers.ROOT, "srla_amd", "srla_corpus")


def _make_corpus(root):
    files = {}
    specs = [("a/one.wav", 2, 16, 48000, 90000, helpers.MUSIC), ("a/two.wav", 2, 16, 48000, 33001, helpers.VARIED),
             ("b/three.wav", 1, 16, 44100, 50000, helpers.SINE), ("b/deep/four.wav", 2, 24, 48000, 40000, helpers.MUSIC),
             ("five.WAV", 2, 16, 48000, 4097, helpers.NOISE), ("six.wav", 2, 8, 22050, 30000, helpers.VARIED),
             ("seven.wav", 2, 16, 48000, 250000, helpers.MUSIC)]
    for i, (rel, nch, bps, rate, n, kind) in enumerate(specs):
        pcm = helpers.synth(kind, 300 + i, rate, nch, n, bps)
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        _write_wav(path, pcm, rate, bps, extensible=(i == 3), extra_chunk=(i == 1))
        files[rel] = (pcm, bps, rate)
    return files


def test_tool_is_built_and_fails_loudly_without_a_gpu(tmp_path):
    assert os.path.exists(TOOL), "srla_amd/srla_corpus is not built (run __graft_entry__.build())"
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    (tmp_path / "in").mkdir()
    p = subprocess.run([TOOL, "-e", str(tmp_path / "in"), str(tmp_path / "out")], capture_output=True, text=True)
    assert p.returncode != 0 and "no CPU fallback" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("world,extra", [(1, []), (2, []), (1, ["--host-deinterleave"])], ids=["one_rank", "two_ranks", "host_deinterleave"])
def test_corpus_encodes_like_the_oracle(tmp_path, world, extra):
    """default: the files' data chunks go to the device as read (SRLAMI355X_EncodeBatchPcm); --host-deinterleave: planar int32
    made on the host (SRLAMI355X_EncodeBatchEx)"""
    files = _make_corpus(str(tmp_path / "in"))
    cli = dict(preset=4, max_block=4096, divisions=2, ltp_order=3)
    seen = {}
    for rank in range(world):
        man = str(tmp_path / ("manifest%d.json" % rank))
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        p = subprocess.run([TOOL, "-e", "-m", "4", "-B", "4096", "-V", "2", "-P", "3", "--manifest", man, "--sha256", "--batch-samples", "100000"] + extra +
                           [str(tmp_path / "in"), str(tmp_path / "out")], capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stderr
        assert "finished:" in p.stdout
        for e in json.load(open(man))["files"]:
            assert e["name"] not in seen and e["error"] == ""
            seen[e["name"]] = e
    assert sorted(seen) == sorted(files)                      # every file exactly once over the ranks
    for rel, (pcm, bps, rate) in files.items():
        want = helpers.Oracle(pcm.shape[0], bits_per_sample=bps, sampling_rate=rate, **cli).encode_whole(pcm)
        got = np.fromfile(os.path.join(str(tmp_path / "out"), os.path.splitext(rel)[0] + ".srl"), dtype=np.uint8)
        assert np.array_equal(got, want), rel
        assert seen[rel]["bytes"] == want.size and seen[rel]["sha256"] == hashlib.sha256(want.tobytes()).hexdigest()


@pytest.mark.gpu
def test_bad_files_are_reported_not_fatal(tmp_path):
    root = tmp_path / "in"
    root.mkdir()
    pcm = helpers.synth(helpers.MUSIC, 1, 48000, 2, 20000)
    _write_wav(str(root / "good.wav"), pcm, 48000, 16)
    (root / "bad.wav").write_bytes(b"RIFF\x00\x00\x00\x00WAVEjunk" + bytes(64))
    p = subprocess.run([TOOL, "-e", str(root), str(tmp_path / "out")], capture_output=True, text=True)
    assert p.returncode == 1 and "bad.wav" in p.stderr
    want = helpers.Oracle(2, preset=4, max_block=4096, divisions=1).encode_whole(pcm)
    assert np.array_equal(np.fromfile(str(tmp_path / "out" / "good.srl"), dtype=np.uint8), want)


@pytest.mark.gpu
def test_incompressible_input_with_small_blocks_fits_the_output_buffer(tmp_path):
    """full-scale 8-bit mono white noise in blocks of 256 samples (the smallest the reference's tool can create an encoder for: it asks
    for 255 parameters, srla_codec.c:95): every block is RAW, an 11-byte header per 256 bytes of samples -- 4.3 % more than the file,
    which the tool's output buffers once (file + 1/32 + 64 KiB) did not hold.  They are sized from the true bound now, PCM + a header
    per minimum block; the reference's tool allocates twice the file (srla_codec.c:125-129)."""
    root = tmp_path / "in"
    root.mkdir()
    pcm = np.random.RandomState(7).randint(-128, 128, size=(1, 6_000_000)).astype(np.int32)
    _write_wav(str(root / "noise.wav"), pcm, 22050, 8)
    p = subprocess.run([TOOL, "-e", "-m", "2", "-B", "256", "-V", "0", str(root), str(tmp_path / "out")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(str(tmp_path / "out" / "noise.srl"), dtype=np.uint8)
    assert got.size > pcm.size + pcm.size // 32 + 65536              # beyond what the old bound allowed
    want = helpers.Oracle(1, bits_per_sample=8, sampling_rate=22050, preset=2, max_block=256, divisions=0).encode_whole(pcm)
    assert np.array_equal(got, want)
