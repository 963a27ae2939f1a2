"""ctypes binding of the SRLA encoder C API (reference: include/srla_encoder.h:8-79,
include/srla.h:29-51).

The binding is ABI-level, so the same class drives either this repo's MI355X library
(``srla_amd/libsrla_mi355x.so``) or the compiled reference (``oracle/_ref/libsrla_ref.so``):
that is what "drop-in" means for this path.  Struct layouts and function signatures are the
reference's own.
"""
import ctypes as C
import numpy as np

# SRLAApiResult, include/srla.h:29-38
OK, INVALID_ARGUMENT, INVALID_FORMAT, INSUFFICIENT_BUFFER, INSUFFICIENT_DATA, \
    PARAMETER_NOT_SET, DETECT_DATA_CORRUPTION, NG = range(8)

HEADER_SIZE = 30


class SRLAHeader(C.Structure):
    _fields_ = [("format_version", C.c_uint32), ("codec_version", C.c_uint32),
                ("num_channels", C.c_uint16), ("num_samples", C.c_uint32),
                ("sampling_rate", C.c_uint32), ("bits_per_sample", C.c_uint16),
                ("offset_lshift", C.c_uint8), ("max_num_samples_per_block", C.c_uint32),
                ("preset", C.c_uint8)]


class SRLAEncodeParameter(C.Structure):
    _fields_ = [("num_channels", C.c_uint16), ("bits_per_sample", C.c_uint16),
                ("sampling_rate", C.c_uint32), ("min_num_samples_per_block", C.c_uint32),
                ("max_num_samples_per_block", C.c_uint32), ("num_lookahead_samples", C.c_uint32),
                ("ltp_order", C.c_uint32), ("num_svr_filter_learning_iteration", C.c_uint32),
                ("preset", C.c_uint8)]


class SRLAEncoderConfig(C.Structure):
    _fields_ = [("max_num_channels", C.c_uint32), ("min_num_samples_per_block", C.c_uint32),
                ("max_num_samples_per_block", C.c_uint32), ("max_num_lookahead_samples", C.c_uint32),
                ("max_num_parameters", C.c_uint32)]


class SRLADecoderConfig(C.Structure):
    _fields_ = [("max_num_channels", C.c_uint32), ("max_num_parameters", C.c_uint32),
                ("check_checksum", C.c_uint8)]


CALLBACK = C.CFUNCTYPE(None, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32)
_I32PP = C.POINTER(C.POINTER(C.c_int32))


def planar_ptrs(arr):
    """int32 [ch][n] C-contiguous numpy array -> const int32_t *const * (keeps `arr` alive)."""
    assert arr.dtype == np.int32 and arr.ndim == 2 and arr.flags.c_contiguous
    ptrs = (C.POINTER(C.c_int32) * arr.shape[0])(
        *[arr[c].ctypes.data_as(C.POINTER(C.c_int32)) for c in range(arr.shape[0])])
    return ptrs


def cli_setup(num_channels, bits_per_sample, sampling_rate, preset=4, max_block=4096,
              divisions=1, lookahead_factor=4, ltp_order=0, svr_iterations=0, min_block=None, lookahead=None):
    """(config, parameter) exactly as `srla -e -m -B -V -L -P` builds them
    (tools/srla_codec/srla_codec.c:91-116).  min_block / lookahead: what the API accepts beyond the tool's
    max >> divisions and factor * max (srla_encoder.c:727-741)."""
    minb = (max_block >> divisions) if min_block is None else min_block
    look = lookahead_factor * max_block if lookahead is None else lookahead
    cfg = SRLAEncoderConfig(8, minb, max_block, look, min(255, max_block))
    par = SRLAEncodeParameter(num_channels, bits_per_sample, sampling_rate, minb, max_block, look, ltp_order, svr_iterations, preset)
    return cfg, par


def _one_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64.so and ask for it under a name that
    does not match an already loaded system copy, so "this library first, torch later" ends with two runtimes, and whichever
    comes second finds no GPU.  The other order is fine (the encoder library then binds to the copy torch loaded, by
    SONAME) -- so if torch is installed, it is loaded first.  C / C++ users of the library are not concerned."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


class EncoderLib:
    """The nine SRLAEncoder_* entry points of include/srla_encoder.h:41-79."""

    def __init__(self, path):
        self.path = path
        _one_hip_runtime()
        lib = self.lib = C.CDLL(path)
        lib.SRLAEncoder_EncodeHeader.argtypes = [C.POINTER(SRLAHeader), C.c_void_p, C.c_uint32]
        lib.SRLAEncoder_EncodeHeader.restype = C.c_int
        lib.SRLAEncoder_CalculateWorkSize.argtypes = [C.POINTER(SRLAEncoderConfig)]
        lib.SRLAEncoder_CalculateWorkSize.restype = C.c_int32
        lib.SRLAEncoder_Create.argtypes = [C.POINTER(SRLAEncoderConfig), C.c_void_p, C.c_int32]
        lib.SRLAEncoder_Create.restype = C.c_void_p
        lib.SRLAEncoder_Destroy.argtypes = [C.c_void_p]
        lib.SRLAEncoder_Destroy.restype = None
        lib.SRLAEncoder_SetEncodeParameter.argtypes = [C.c_void_p, C.POINTER(SRLAEncodeParameter)]
        lib.SRLAEncoder_SetEncodeParameter.restype = C.c_int
        lib.SRLAEncoder_ComputeBlockSize.argtypes = [C.c_void_p, _I32PP, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.SRLAEncoder_ComputeBlockSize.restype = C.c_int
        for name in ("SRLAEncoder_EncodeBlock", "SRLAEncoder_EncodeOptimalPartitionedBlock"):
            fn = getattr(lib, name)
            fn.argtypes = [C.c_void_p, _I32PP, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
            fn.restype = C.c_int
        lib.SRLAEncoder_EncodeWhole.argtypes = [C.c_void_p, _I32PP, C.c_uint32, C.c_void_p, C.c_uint32,
                                                C.POINTER(C.c_uint32), C.c_void_p]
        lib.SRLAEncoder_EncodeWhole.restype = C.c_int

    # -- thin wrappers ---------------------------------------------------------------------
    def create(self, config):
        return self.lib.SRLAEncoder_Create(C.byref(config) if config is not None else None, None, 0)

    def destroy(self, enc):
        self.lib.SRLAEncoder_Destroy(enc)

    def set_parameter(self, enc, parameter):
        return self.lib.SRLAEncoder_SetEncodeParameter(enc, C.byref(parameter) if parameter is not None else None)

    def compute_block_size(self, enc, pcm, num_samples=None):
        n = pcm.shape[1] if num_samples is None else num_samples
        out = C.c_uint32(0)
        rc = self.lib.SRLAEncoder_ComputeBlockSize(enc, planar_ptrs(pcm), n, C.byref(out))
        return rc, out.value

    def _encode(self, fn, enc, pcm, cap, num_samples=None):
        n = pcm.shape[1] if num_samples is None else num_samples
        cap = int(cap if cap is not None else 2 * pcm.size * 4 + 1024)
        buf = np.zeros(cap, dtype=np.uint8)
        out = C.c_uint32(0)
        rc = fn(enc, planar_ptrs(pcm), n, buf.ctypes.data_as(C.c_void_p), cap, C.byref(out))
        return rc, buf[:out.value].copy()

    def encode_block(self, enc, pcm, cap=None, num_samples=None):
        return self._encode(self.lib.SRLAEncoder_EncodeBlock, enc, pcm, cap, num_samples)

    def encode_partitioned(self, enc, pcm, cap=None, num_samples=None):
        return self._encode(self.lib.SRLAEncoder_EncodeOptimalPartitionedBlock, enc, pcm, cap, num_samples)

    def encode_whole(self, enc, pcm, cap=None, callback=None):
        cap = int(cap if cap is not None else 2 * pcm.size * 4 + 1024)
        buf = np.zeros(cap, dtype=np.uint8)
        out = C.c_uint32(0)
        cb = CALLBACK(callback) if callback is not None else None
        rc = self.lib.SRLAEncoder_EncodeWhole(enc, planar_ptrs(pcm), pcm.shape[1],
                                              buf.ctypes.data_as(C.c_void_p), cap, C.byref(out),
                                              C.cast(cb, C.c_void_p) if cb is not None else None)
        return rc, buf[:out.value].copy()

    def encode(self, pcm, bits_per_sample=16, sampling_rate=48000, **cli):
        """One-shot `srla -e`: Create -> SetEncodeParameter -> EncodeWhole -> Destroy."""
        cfg, par = cli_setup(pcm.shape[0], bits_per_sample, sampling_rate, **cli)
        enc = self.create(cfg)
        if not enc:
            raise RuntimeError("SRLAEncoder_Create failed")
        try:
            rc = self.set_parameter(enc, par)
            if rc != OK:
                raise RuntimeError("SRLAEncoder_SetEncodeParameter -> %d" % rc)
            rc, data = self.encode_whole(enc, pcm)
            if rc != OK:
                raise RuntimeError("SRLAEncoder_EncodeWhole -> %d" % rc)
            return data
        finally:
            self.destroy(enc)


class DecoderLib:
    """SRLADecoder_* (include/srla_decoder.h:22-50); only the compiled reference exports it."""

    def __init__(self, path):
        lib = self.lib = C.CDLL(path)
        lib.SRLADecoder_DecodeHeader.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(SRLAHeader)]
        lib.SRLADecoder_DecodeHeader.restype = C.c_int
        lib.SRLADecoder_Create.argtypes = [C.POINTER(SRLADecoderConfig), C.c_void_p, C.c_int32]
        lib.SRLADecoder_Create.restype = C.c_void_p
        lib.SRLADecoder_Destroy.argtypes = [C.c_void_p]
        lib.SRLADecoder_DecodeWhole.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, _I32PP, C.c_uint32, C.c_uint32]
        lib.SRLADecoder_DecodeWhole.restype = C.c_int

    def decode(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        hdr = SRLAHeader()
        rc = self.lib.SRLADecoder_DecodeHeader(data.ctypes.data_as(C.c_void_p), data.size, C.byref(hdr))
        if rc != OK:
            raise RuntimeError("DecodeHeader -> %d" % rc)
        cfg = SRLADecoderConfig(8, 255, 1)
        dec = self.lib.SRLADecoder_Create(C.byref(cfg), None, 0)
        out = np.zeros((hdr.num_channels, hdr.num_samples), dtype=np.int32)
        rc = self.lib.SRLADecoder_DecodeWhole(dec, data.ctypes.data_as(C.c_void_p), data.size,
                                              planar_ptrs(out), hdr.num_channels, hdr.num_samples)
        self.lib.SRLADecoder_Destroy(dec)
        if rc != OK:
            raise RuntimeError("DecodeWhole -> %d" % rc)
        return out, hdr


class BatchCall:
    """SRLAMI355X_EncodeBatch (include/srla_mi355x.h): many streams of one format in one call.  Holds the pointer tables
    of a fixed set of input arrays / output buffers so that repeated calls cost nothing on the Python side."""

    def __init__(self, lib, pcms, outs):
        self.lib = lib
        fn = lib.lib.SRLAMI355X_EncodeBatch
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        fn.restype = C.c_int
        self.fn = fn
        n = len(pcms)
        self.n = n
        self._keep = [planar_ptrs(p) for p in pcms]
        self.inputs = (C.c_void_p * n)(*[C.cast(p, C.c_void_p) for p in self._keep])
        self.num_samples = (C.c_uint32 * n)(*[p.shape[1] for p in pcms])
        self.data = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        self.data_size = (C.c_uint32 * n)(*[o.size for o in outs])
        self.results = (C.c_int * n)()
        self._outs = outs
        self._pcms = pcms

    def run(self, enc, out_sizes):
        return self.fn(enc, self.n, self.inputs, self.num_samples, self.data, self.data_size, out_sizes, self.results)


class PcmBatchCall:
    """SRLAMI355X_EncodeBatchPcm with a fixed set of streams (interleaved little-endian PCM frames, what a WAV data chunk holds) and
    output buffers: the pointer tables are built once, so that repeated calls cost nothing on the Python side."""

    def __init__(self, lib, frames, num_samples, bytes_per_sample, outs):
        fn = lib.lib.SRLAMI355X_EncodeBatchPcm
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        fn.restype = C.c_int
        self.fn = fn
        n = self.n = len(frames)
        self._keep = (frames, outs)
        self.ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        self.num_samples = (C.c_uint32 * n)(*[int(x) for x in num_samples])
        self.bytes_per_sample = int(bytes_per_sample)
        self.data = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        self.data_size = (C.c_uint32 * n)(*[o.size for o in outs])
        self.results = (C.c_int * n)()

    def run(self, enc, out_sizes):
        return self.fn(enc, self.n, self.ptrs, self.num_samples, self.bytes_per_sample, self.data, self.data_size, out_sizes, self.results)


def encode_batch(lib, enc, pcms, caps=None):
    """-> (rc, [stream bytes or None], [per-stream result codes])"""
    outs = [np.zeros(int(caps[i] if caps else 2 * p.size * 4 + 1024), dtype=np.uint8) for i, p in enumerate(pcms)]
    call = BatchCall(lib, pcms, outs)
    sizes = (C.c_uint32 * len(pcms))()
    rc = call.run(enc, sizes)
    res = list(call.results)
    return rc, [outs[i][:sizes[i]].copy() if res[i] == OK else None for i in range(len(pcms))], res


def encode_batch_pcm(lib, enc, frames, num_samples, bytes_per_sample, caps=None):
    """SRLAMI355X_EncodeBatchPcm: `frames` = one bytes-like object per stream holding interleaved little-endian PCM frames (a
    WAV data chunk).  -> (rc, [stream bytes or None], [per-stream result codes])"""
    n = len(frames)
    fn = lib.lib.SRLAMI355X_EncodeBatchPcm
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    keep = [np.frombuffer(f, dtype=np.uint8) for f in frames]
    ptrs = (C.c_void_p * n)(*[k.ctypes.data for k in keep])
    nsmp = (C.c_uint32 * n)(*[int(x) for x in num_samples])
    outs = [np.zeros(int(caps[i] if caps else 2 * len(frames[i]) + 4096), dtype=np.uint8) for i in range(n)]
    data = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    sizes_in = (C.c_uint32 * n)(*[o.size for o in outs])
    sizes = (C.c_uint32 * n)()
    res = (C.c_int * n)()
    rc = fn(enc, n, ptrs, nsmp, int(bytes_per_sample), data, sizes_in, sizes, res)
    return rc, [outs[i][:sizes[i]].copy() if res[i] == OK else None for i in range(n)], list(res)

