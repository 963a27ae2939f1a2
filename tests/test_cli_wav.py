"""The command line front end: WAV reading restated from libs/wav/src/wav.c, and `-e` end to end."""
import ctypes as C
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import helpers
from srla_amd import wavio


def _write_wav(path, pcm, rate, bps, extensible=False, extra_chunk=False):
    nch, n = pcm.shape
    inter = pcm.T.reshape(-1)
    if bps == 8:
        body = (inter + 128).astype(np.uint8).tobytes()
    elif bps == 16:
        body = inter.astype("<i2").tobytes()
    elif bps == 24:
        u = inter.astype(np.int64) & 0xFFFFFF
        body = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
    else:
        body = inter.astype("<i4").tobytes()
    bytes_ps = bps // 8
    if extensible:
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, nch, rate, rate * nch * bytes_ps, nch * bytes_ps, bps, 22, bps, 3) \
            + bytes([1, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xAA, 0, 0x38, 0x9B, 0x71])
    else:
        fmt = struct.pack("<HHIIHH", 1, nch, rate, rate * nch * bytes_ps, nch * bytes_ps, bps)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunk:
        chunks += b"LIST" + struct.pack("<I", 10) + b"INFOabcdef"
    chunks += b"data" + struct.pack("<I", len(body)) + body
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


CASES = [(1, 8, False, False), (2, 16, False, False), (2, 16, False, True), (2, 24, False, False), (3, 16, True, False),
         (2, 24, True, True)]


@pytest.mark.parametrize("nch,bps,ext,extra", CASES)
def test_read_wav_returns_the_samples(tmp_path, nch, bps, ext, extra):
    pcm = helpers.synth(helpers.VARIED, 7, 44100, nch, 5000, bps)
    path = str(tmp_path / "a.wav")
    _write_wav(path, pcm, 44100, bps, ext, extra)
    got, rate, got_bps = wavio.read_wav(path)
    assert rate == 44100 and got_bps == bps and got.dtype == np.int32 and got.flags["C_CONTIGUOUS"]
    assert np.array_equal(got, pcm)


class _WAVFormat(C.Structure):
    _fields_ = [("file_format", C.c_int), ("num_channels", C.c_uint32), ("sampling_rate", C.c_uint32),
                ("bits_per_sample", C.c_uint32), ("num_samples", C.c_uint32), ("u", C.c_uint8 * 24)]


class _WAVFile(C.Structure):
    _fields_ = [("format", _WAVFormat), ("data", C.POINTER(C.POINTER(C.c_int32)))]


@pytest.mark.parametrize("nch,bps,ext,extra", CASES)
def test_read_wav_agrees_with_the_reference_parser(tmp_path, nch, bps, ext, extra):
    """Where the reference could be compiled (oracle/_ref), its own WAV_CreateFromFile decides."""
    if not os.path.exists(helpers.REF_SO):
        pytest.skip("reference not built here")
    ref = C.CDLL(helpers.REF_SO)
    if not hasattr(ref, "WAV_CreateFromFile"):
        pytest.skip("reference library built without libs/wav")
    ref.WAV_CreateFromFile.restype = C.POINTER(_WAVFile)
    ref.WAV_CreateFromFile.argtypes = [C.c_char_p]
    ref.WAV_Destroy.argtypes = [C.POINTER(_WAVFile)]
    # the reference's parser needs more than one 32 KB read-ahead buffer of file (its relative seek assumes a full
    # buffer, libs/wav/src/wav.c:937-941), so the files are made larger than that
    pcm = helpers.synth(helpers.MUSIC, 8, 48000, nch, 70000, bps)
    path = str(tmp_path / "b.wav")
    _write_wav(path, pcm, 48000, bps, ext, extra)
    w = ref.WAV_CreateFromFile(path.encode())
    assert bool(w)
    f = w.contents.format
    want = np.stack([np.ctypeslib.as_array(w.contents.data[ch], shape=(f.num_samples,)).copy() for ch in range(f.num_channels)])
    meta = (f.num_channels, f.sampling_rate, f.bits_per_sample, f.num_samples)
    ref.WAV_Destroy(w)
    got, rate, got_bps = wavio.read_wav(path)
    assert meta == (got.shape[0], rate, got_bps, got.shape[1])
    assert np.array_equal(got, want)


def test_malformed_files_are_refused(tmp_path):
    p = str(tmp_path / "x.wav")
    open(p, "wb").write(b"RIFF\x00\x00\x00\x00WAVEdata\x00\x00\x00\x00")
    with pytest.raises(wavio.WavError):
        wavio.read_wav(p)
    pcm = helpers.synth(helpers.SINE, 1, 8000, 1, 100)
    _write_wav(p, pcm, 8000, 16)
    blob = bytearray(open(p, "rb").read())
    blob[20] = 3                                            # IEEE float tag
    open(p, "wb").write(bytes(blob))
    with pytest.raises(wavio.WavError):
        wavio.read_wav(p)


@pytest.mark.gpu
def test_cli_encodes_like_the_oracle(tmp_path):
    pcm = helpers.synth(helpers.MUSIC, 21, 48000, 2, 100000)
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.srl")
    _write_wav(src, pcm, 48000, 16, extra_chunk=True)
    r = subprocess.run([sys.executable, "-m", "srla_amd.cli", "-e", "-m", "4", "-B", "4096", "-V", "2", "-P", "3", src, dst],
                       cwd=helpers.ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("finished: ")
    want = helpers.Oracle(2, preset=4, max_block=4096, divisions=2, ltp_order=3).encode_whole(pcm)
    assert np.array_equal(np.fromfile(dst, dtype=np.uint8), want)


@pytest.mark.gpu
def test_cli_corpus_mode(tmp_path):
    ind, outd = tmp_path / "in", tmp_path / "out"
    (ind / "sub").mkdir(parents=True)
    want = {}
    for i, (nch, n) in enumerate([(2, 30000), (1, 50000), (2, 8192)]):
        pcm = helpers.synth(helpers.VARIED, 30 + i, 44100, nch, n)
        name = ("sub/f%d" % i) if i == 1 else ("f%d" % i)
        _write_wav(str(ind / (name + ".wav")), pcm, 44100, 16)
        want[name] = helpers.Oracle(nch, sampling_rate=44100).encode_whole(pcm)
    r = subprocess.run([sys.executable, "-m", "srla_amd.cli", "-e", "--corpus", str(ind), "--out", str(outd)],
                       cwd=helpers.ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "3 files" in r.stdout
    for name, data in want.items():
        assert np.array_equal(np.fromfile(str(outd / (name + ".srl")), dtype=np.uint8), data)
