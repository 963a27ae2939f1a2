"""WAV reader for the command line front end: what `WAV_CreateFromFile` of the reference accepts, restated with
numpy (libs/wav/src/wav.c:136-282 format parsing, :479-556 PCM conversion).

Accepted: RIFF/WAVE whose first chunk is `fmt ` of 16 bytes (format tag 1, PCMWAVEFORMAT) or 40 bytes (tag 0xFFFE
with a 22-byte extension, WAVEFORMATEXTENSIBLE); any chunks between `fmt ` and `data` are skipped; 8-bit samples
are offset binary (value - 128), 16 / 24 / 32-bit little endian two's complement.  Samples come back planar,
int32 [channels][samples], exactly as the reference hands them to SRLAEncoder_EncodeWhole."""
import struct

import numpy as np


class WavError(ValueError):
    pass


def read_wav(path):
    """-> (pcm int32 [nch][n] C-contiguous, sampling_rate, bits_per_sample)"""
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) < 12 or blob[0:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise WavError("%s: not a RIFF/WAVE file" % path)
    pos = 12
    if blob[pos:pos + 4] != b"fmt ":                      # wav.c:242: the format chunk must come first
        raise WavError("%s: 'fmt ' chunk expected right after 'WAVE'" % path)
    (fmt_size,) = struct.unpack_from("<I", blob, pos + 4)
    if fmt_size not in (16, 40):                           # wav.c:153-158
        raise WavError("%s: unsupported fmt chunk size %d" % (path, fmt_size))
    tag, nch, rate, _byte_rate, _align, bps = struct.unpack_from("<HHIIHH", blob, pos + 8)
    if (fmt_size == 16 and tag != 1) or (fmt_size == 40 and tag != 0xFFFE):
        raise WavError("%s: unsupported format tag 0x%04x" % (path, tag))
    if fmt_size == 40:
        (ext,) = struct.unpack_from("<H", blob, pos + 24)
        if ext != 22:                                      # wav.c:191-194
            raise WavError("%s: bad WAVEFORMATEXTENSIBLE extension size %d" % (path, ext))
    pos += 8 + fmt_size
    while True:                                            # wav.c:251-270: skip everything up to 'data'
        if pos + 8 > len(blob):
            raise WavError("%s: no 'data' chunk" % path)
        cid = blob[pos:pos + 4]
        (size,) = struct.unpack_from("<I", blob, pos + 4)
        pos += 8
        if cid == b"data":
            break
        pos += size                                        # the reference seeks by the raw size (no pad byte)
    if bps not in (8, 16, 24, 32) or nch == 0:
        raise WavError("%s: unsupported %d-bit / %d-channel data" % (path, bps, nch))
    bytes_ps = bps // 8
    frame = bytes_ps * nch
    if size % frame:
        raise WavError("%s: data size %d is not a whole number of sample frames" % (path, size))
    n = size // frame
    raw = np.frombuffer(blob, dtype=np.uint8, count=n * frame, offset=pos)
    if raw.size < n * frame:
        raise WavError("%s: truncated data chunk" % path)
    if bps == 8:
        inter = raw.astype(np.int32) - 128                                       # wav.c:841-845
    elif bps == 16:
        inter = raw.view("<i2").astype(np.int32)
    elif bps == 24:
        b = raw.reshape(-1, 3).astype(np.int32)
        inter = ((b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) << 8) >> 8          # sign extension, wav.c:855-859
    else:
        inter = raw.view("<i4").astype(np.int32)
    pcm = np.ascontiguousarray(inter.reshape(n, nch).T)
    return pcm, int(rate), int(bps)
