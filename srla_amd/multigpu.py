"""One long stream over several GPUs ("contiguous window ranges per GPU").

Look-ahead windows carry no state from one to the next (srla_encoder.c:1756-1783 re-seeds everything per block), so a
stream is sharded by giving every rank -- one process per GPU -- a contiguous range of whole windows:

    1. every rank ORs the samples of its range; ONE tiny all-reduce (bitwise OR of a single integer: the offset left shift
       is the only whole-stream quantity, srla_utility.c:177) gives the stream's shift;
    2. every rank encodes its range with SRLAMI355X_EncodeWindows (blocks only, no header);
    3. rank 0 gathers the byte strings and concatenates them in rank order behind the 30-byte header.

No collective touches the samples; the bytes are, by construction and by test, those of SRLAEncoder_EncodeWhole on one GPU.
"""
import ctypes as C

import numpy as np

from . import capi


def shard_ranges(num_samples, window_len, world):
    """-> [(first sample, count)] per rank: contiguous ranges of whole windows, balanced; the last rank's range ends the
    stream and holds its last two windows at least (ranks without work get (start, 0))."""
    nwin = -(-num_samples // window_len)
    last_min = min(nwin, 2)
    free = nwin - last_min
    base, extra = divmod(free, world)
    counts = [base + (1 if r < extra else 0) for r in range(world)]
    counts[world - 1] += last_min
    out, start = [], 0
    for r in range(world):
        first = start * window_len
        n = min(num_samples, (start + counts[r]) * window_len) - first if counts[r] else 0
        out.append((first, n))
        start += counts[r]
    return out


def offset_lshift_of(mask):
    mask &= 0xFFFFFFFF
    if mask == 0:
        return 0
    sh = 0
    while not (mask >> sh) & 1:
        sh += 1
    return sh


_host_group = None


def host_group(group=None):
    """The process group for the two host-side exchanges (a single integer and the finished bytes).  They are CPU objects:
    on a job whose default backend is nccl (= RCCL, device tensors only, no bitwise reductions) a gloo group over the same
    ranks is created once -- collectively, by every rank's first call -- and reused."""
    global _host_group
    import torch.distributed as dist
    if group is not None:
        return group
    if dist.get_backend() != "nccl":
        return None
    if _host_group is None:
        _host_group = dist.new_group(backend="gloo")
    return _host_group


def encode_stream_sharded(encode_range, pcm, header_fn, window_len, rank=0, world=1, group=None):
    """encode_range(range_pcm, offset_lshift, is_stream_end) -> uint8 array (blocks of the range);
    header_fn(offset_lshift) or header_fn(offset_lshift, num_samples) -> 30 header bytes.  Returns the complete stream on
    rank 0, None elsewhere."""
    n = pcm.shape[1]
    first, count = shard_ranges(n, window_len, world)[rank]
    mine = np.ascontiguousarray(pcm[:, first:first + count])
    # the range's OR: from the library where the encoder object offers it (SRLAMI355X_OrMask: the handle's host threads, no
    # temporaries), else one pass over an unsigned view of the range
    or_mask = getattr(getattr(encode_range, "__self__", None), "or_mask", None)
    if not count:
        mask = 0
    elif or_mask is not None:
        mask = or_mask(mine)
    else:
        # (the low 32 bits of every sample, whatever integer type the caller's planes have: a view would OR the upper halves of
        # negative int64 samples in)
        mask = int(np.bitwise_or.reduce(mine.astype(np.int32, copy=False).view(np.uint32), axis=None))
    if world > 1:
        import torch.distributed as dist
        group = host_group(group)
        masks = [None] * world
        dist.all_gather_object(masks, mask, group=group)       # one integer per rank, OR-ed on the host
        mask = 0
        for m in masks:
            mask |= int(m)
    shift = offset_lshift_of(mask)
    blocks = encode_range(mine, shift, first + count == n) if count else np.zeros(0, np.uint8)
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world if rank == 0 else None
        dist.gather_object(np.ascontiguousarray(blocks, dtype=np.uint8).tobytes(), parts, dst=0, group=group)
        if rank != 0:
            return None
        body = b"".join(parts)
    else:
        body = np.ascontiguousarray(blocks, dtype=np.uint8).tobytes()
    import inspect
    takes_length = len(inspect.signature(header_fn).parameters) >= 2
    hdr = header_fn(shift, n) if takes_length else header_fn(shift)
    return np.frombuffer(bytes(hdr) + body, dtype=np.uint8).copy()


class WindowEncoder:
    """SRLAMI355X_EncodeWindows + SRLAEncoder_EncodeHeader of one handle, as the callables encode_stream_sharded wants."""

    def __init__(self, lib, nch, bps, rate, **cli):
        self.lib = lib
        self.cfg, self.par = capi.cli_setup(nch, bps, rate, **cli)
        self.enc = lib.create(self.cfg)
        if not self.enc or lib.set_parameter(self.enc, self.par) != capi.OK:
            raise RuntimeError("cannot set up the encoder")
        fn = lib.lib.SRLAMI355X_EncodeWindows
        fn.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        fn.restype = C.c_int
        self.fn = fn
        same = self.par.min_num_samples_per_block == self.par.max_num_samples_per_block
        self.window_len = self.par.max_num_samples_per_block if same else self.par.num_lookahead_samples
        self.num_samples = 0
        orfn = lib.lib.SRLAMI355X_OrMask
        orfn.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.c_uint32, C.POINTER(C.c_uint32)]
        orfn.restype = C.c_int
        self.orfn = orfn
        self.buf = np.zeros(0, dtype=np.uint8)          # one output buffer per encoder, grown on demand (np.empty: never zeroed again)

    def or_mask(self, pcm):
        m = C.c_uint32(0)
        rc = self.orfn(self.enc, capi.planar_ptrs(pcm), pcm.shape[1], C.byref(m))
        if rc != capi.OK:
            raise RuntimeError("SRLAMI355X_OrMask -> %d" % rc)
        return int(m.value)

    def encode_range(self, pcm, shift, is_end):
        # bound of a range's blocks: the raw PCM + an 11-byte header per minimum block (RAW is the fallback, srla_encoder.c:1323-1327)
        bps = self.par.bits_per_sample
        cap = pcm.size * (bps // 8) + 16 * (pcm.shape[1] // max(1, self.par.min_num_samples_per_block) + 2) + 4096
        if self.buf.size < cap:
            self.buf = np.empty(cap, dtype=np.uint8)
        out = C.c_uint32(0)
        rc = self.fn(self.enc, capi.planar_ptrs(pcm), pcm.shape[1], shift, 1 if is_end else 0, self.buf.ctypes.data_as(C.c_void_p), self.buf.size, C.byref(out))
        if rc != capi.OK:
            raise RuntimeError("SRLAMI355X_EncodeWindows -> %d" % rc)
        return self.buf[:out.value].copy()

    def header(self, shift, num_samples=None):
        """The 30 header bytes of a stream of `num_samples` samples per channel (default: self.num_samples)."""
        if num_samples is not None:
            self.num_samples = int(num_samples)
        if self.num_samples <= 0:
            raise ValueError("WindowEncoder.header: the stream's length (num_samples) is needed")
        hdr = capi.SRLAHeader(10, 18, self.par.num_channels, self.num_samples, self.par.sampling_rate, self.par.bits_per_sample,
                              shift, self.par.max_num_samples_per_block, self.par.preset)
        buf = np.zeros(capi.HEADER_SIZE, np.uint8)
        if self.lib.lib.SRLAEncoder_EncodeHeader(C.byref(hdr), buf.ctypes.data_as(C.c_void_p), capi.HEADER_SIZE) != capi.OK:
            raise RuntimeError("SRLAEncoder_EncodeHeader failed")
        return buf

    def close(self):
        if self.enc:
            self.lib.destroy(self.enc)
            self.enc = None
