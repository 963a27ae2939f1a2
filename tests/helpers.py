"""Shared test plumbing: library loaders, the oracle binding, the synthetic-signal generator.

Nothing here is product code.  `oracle()` loads oracle/liboracle.so (the CPU checker),
`reference()` loads oracle/_ref/libsrla_ref.so when it exists (this container only),
`synth()` calls tools/synth/libsynth.so.
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from srla_amd import capi  # noqa: E402

ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsrla_ref.so")
SYNTH_SO = os.path.join(ROOT, "tools", "synth", "libsynth.so")
PRODUCT_SO = os.environ.get("SRLA_PRODUCT_SO", os.path.join(ROOT, "srla_amd", "libsrla_mi355x.so"))   # override: A/B runs of two builds
GOLDEN = os.path.join(ROOT, "tests", "golden")

MAX_ORDER = 255


class OracleConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_channels", "bits_per_sample", "sampling_rate", "min_block",
                                          "max_block", "lookahead", "ltp_order", "preset")]


class OracleChannelParams(C.Structure):
    _fields_ = [("preemph_prev", C.c_int32), ("preemph_coef", C.c_int32), ("lpc_order", C.c_uint32),
                ("lpc_rshift", C.c_uint32), ("use_sum", C.c_uint32), ("ltp_period", C.c_uint32),
                ("ltp_coef", C.c_int32 * 3), ("code_length", C.c_uint32), ("res_code_type", C.c_uint32),
                ("res_porder", C.c_uint32), ("res_bits", C.c_uint32), ("lpc_coef", C.c_int32 * MAX_ORDER)]

    def as_dict(self):
        return dict(preemph_prev=self.preemph_prev, preemph_coef=self.preemph_coef, lpc_order=self.lpc_order,
                    lpc_rshift=self.lpc_rshift, use_sum=self.use_sum, ltp_period=self.ltp_period,
                    ltp_coef=list(self.ltp_coef), code_length=self.code_length,
                    res_code_type=self.res_code_type, res_porder=self.res_porder, res_bits=self.res_bits,
                    lpc_coef=list(self.lpc_coef)[:self.lpc_order])


class OracleBlockInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("block_type", "ch_method", "payload_bits", "block_bytes")]


def _ensure(path, cmd, cwd):
    if not os.path.exists(path):
        subprocess.check_call(cmd, cwd=cwd, shell=True)
    return path


_cache = {}


def oracle_lib():
    if "oracle" not in _cache:
        _ensure(ORACLE_SO, "make -s", os.path.join(ROOT, "oracle"))
        lib = C.CDLL(ORACLE_SO)
        lib.oracle_create.restype = C.c_void_p
        lib.oracle_create.argtypes = [C.POINTER(OracleConfig)]
        lib.oracle_destroy.argtypes = [C.c_void_p]
        lib.oracle_set_parameter.argtypes = [C.c_void_p, C.POINTER(OracleConfig)]
        lib.oracle_set_parameter.restype = C.c_int
        lib.oracle_set_offset_lshift.argtypes = [C.c_void_p, C.c_uint32]
        lib.oracle_set_svr_iterations.argtypes = [C.c_void_p, C.c_uint32]
        lib.oracle_svr_refine.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        pp = C.POINTER(C.POINTER(C.c_int32))
        lib.oracle_encode_whole.argtypes = [C.c_void_p, pp, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.oracle_encode_block.argtypes = lib.oracle_encode_whole.argtypes
        lib.oracle_compute_block_size.argtypes = [C.c_void_p, pp, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.oracle_search_partitions.argtypes = [C.c_void_p, pp, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
        lib.oracle_analyze_block.argtypes = [C.c_void_p, pp, C.c_uint32, C.POINTER(OracleBlockInfo),
                                             C.c_void_p, C.c_void_p, pp]
        lib.oracle_analyze_channel.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                               C.POINTER(OracleChannelParams)]
        lib.oracle_decode_whole.argtypes = [C.c_void_p, C.c_uint32, pp, C.c_uint32, C.c_uint32]
        lib.oracle_decode_header.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(OracleConfig),
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.oracle_list_blocks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_uint32, C.POINTER(C.c_uint32)]
        lib.oracle_fletcher16.restype = C.c_uint16
        lib.oracle_fletcher16.argtypes = [C.c_void_p, C.c_uint32]
        lib.oracle_offset_lshift.argtypes = [pp, C.c_uint32, C.c_uint32]
        lib.oracle_offset_lshift.restype = C.c_uint32
        lib.oracle_preemphasis_coef.argtypes = [C.c_void_p, C.c_uint32]
        lib.oracle_preemphasis_coef.restype = C.c_int32
        lib.oracle_preemphasis.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]
        lib.oracle_fft_real.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.oracle_autocorr.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        lib.oracle_levinson.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.oracle_select_order.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.oracle_select_order.restype = C.c_uint32
        lib.oracle_quantize.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
        lib.oracle_lpc_predict.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        lib.oracle_ltp_predict.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_uint32]
        lib.oracle_detect_pitch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.oracle_ltp_coefficients.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                C.POINTER(C.c_uint32)]
        lib.oracle_residual_code_search.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.oracle_rice_k.argtypes = [C.c_double]
        lib.oracle_rice_k.restype = C.c_uint32
        lib.oracle_recursive_rice_k2.argtypes = [C.c_double]
        lib.oracle_recursive_rice_k2.restype = C.c_uint32
        lib.oracle_coef_bits.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.oracle_coef_bits.restype = C.c_uint32
        lib.oracle_dijkstra.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double),
                                        C.c_void_p]
        _cache["oracle"] = lib
    return _cache["oracle"]


def have_reference():
    if os.path.exists(REF_SO):
        return True
    if os.path.isdir("/root/reference"):
        try:
            subprocess.check_call("make -s ref", cwd=os.path.join(ROOT, "oracle"), shell=True)
        except subprocess.CalledProcessError:
            return False
        return os.path.exists(REF_SO)
    return False


def reference_encoder():
    return capi.EncoderLib(REF_SO)


def reference_decoder():
    return capi.DecoderLib(REF_SO)


def synth(kind, seed, rate, nch, n, bps=16):
    if "synth" not in _cache:
        _ensure(SYNTH_SO, "gcc -O2 -fPIC -shared -o libsynth.so synth.c", os.path.join(ROOT, "tools", "synth"))
        _cache["synth"] = C.CDLL(SYNTH_SO)
    lib = _cache["synth"]
    a = np.zeros((nch, n), dtype=np.int32)
    rc = lib.synth_generate(C.c_uint32(kind), C.c_uint64(seed), C.c_uint32(rate), C.c_uint32(nch),
                            C.c_uint32(n), C.c_uint32(bps), capi.planar_ptrs(a))
    assert rc == 0
    return a


SINE, MUSIC, VARIED, NOISE = 0, 1, 2, 3


def synth_spec(sp):
    """input of a golden entry: {kind, seed, rate, nch, n, bps} [+ first, count: a slice of the n samples; + lshift: low bits cleared]"""
    a = synth(sp["kind"], sp["seed"], sp["rate"], sp["nch"], sp["n"], sp["bps"])
    if "first" in sp:
        a = np.ascontiguousarray(a[:, sp["first"]:sp["first"] + sp["count"]])
    if sp.get("lshift"):
        a = np.ascontiguousarray((a >> sp["lshift"]) << sp["lshift"])
    return a


def sha256(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


class Oracle:
    """Handle-style wrapper over oracle/liboracle.so taking the same knobs as `srla -e`."""

    def __init__(self, num_channels, bits_per_sample=16, sampling_rate=48000, preset=4, max_block=4096,
                 divisions=1, lookahead_factor=4, ltp_order=0, min_block=None, lookahead=None, svr_iterations=0):
        self.lib = oracle_lib()
        minb = (max_block >> divisions) if min_block is None else min_block
        look = lookahead_factor * max_block if lookahead is None else lookahead
        self.cfg = OracleConfig(num_channels, bits_per_sample, sampling_rate, minb, max_block, look, ltp_order, preset)
        self.h = self.lib.oracle_create(C.byref(self.cfg))
        if not self.h:
            raise ValueError("oracle_create rejected the configuration")
        if svr_iterations:
            self.lib.oracle_set_svr_iterations(self.h, svr_iterations)

    def close(self):
        if self.h:
            self.lib.oracle_destroy(self.h)
            self.h = None

    __del__ = close

    def set_parameter(self, preset=4, max_block=4096, divisions=1, lookahead_factor=4, ltp_order=0, min_block=None, lookahead=None,
                      svr_iterations=0):
        """SetEncodeParameter on the used handle: the persistent buffer stays (same maximum block and channels)"""
        minb = (max_block >> divisions) if min_block is None else min_block
        look = lookahead_factor * max_block if lookahead is None else lookahead
        cfg = OracleConfig(self.cfg.num_channels, self.cfg.bits_per_sample, self.cfg.sampling_rate, minb, max_block, look, ltp_order, preset)
        if self.lib.oracle_set_parameter(self.h, C.byref(cfg)) != 0:
            raise ValueError("oracle_set_parameter rejected the parameters")
        self.cfg = cfg
        self.lib.oracle_set_svr_iterations(self.h, svr_iterations)

    def set_offset_lshift(self, s):
        self.lib.oracle_set_offset_lshift(self.h, s)

    def encode_whole(self, pcm):
        cap = 2 * pcm.size * 4 + 4096
        buf = np.zeros(cap, dtype=np.uint8)
        out = C.c_uint32(0)
        rc = self.lib.oracle_encode_whole(self.h, capi.planar_ptrs(pcm), pcm.shape[1],
                                          buf.ctypes.data_as(C.c_void_p), cap, C.byref(out))
        if rc != 0:
            raise RuntimeError("oracle_encode_whole -> %d" % rc)
        return buf[:out.value].copy()

    def encode_block(self, pcm):
        cap = 2 * pcm.size * 4 + 4096
        buf = np.zeros(cap, dtype=np.uint8)
        out = C.c_uint32(0)
        rc = self.lib.oracle_encode_block(self.h, capi.planar_ptrs(pcm), pcm.shape[1],
                                          buf.ctypes.data_as(C.c_void_p), cap, C.byref(out))
        if rc != 0:
            raise RuntimeError("oracle_encode_block -> %d" % rc)
        return buf[:out.value].copy()

    def compute_block_size(self, pcm):
        out = C.c_uint32(0)
        rc = self.lib.oracle_compute_block_size(self.h, capi.planar_ptrs(pcm), pcm.shape[1], C.byref(out))
        if rc != 0:
            raise RuntimeError("oracle_compute_block_size -> %d" % rc)
        return out.value

    def search_partitions(self, pcm):
        parts = (C.c_uint32 * 1100)()
        n = C.c_uint32(0)
        rc = self.lib.oracle_search_partitions(self.h, capi.planar_ptrs(pcm), pcm.shape[1], C.byref(n), parts)
        if rc != 0:
            raise RuntimeError("oracle_search_partitions -> %d" % rc)
        return list(parts[:n.value])

    def analyze_block(self, pcm):
        """-> (info, chosen params per channel, [L,R,M,S] variant params, chosen residuals)"""
        nch, n = pcm.shape
        info = OracleBlockInfo()
        params = (OracleChannelParams * 8)()
        variants = (OracleChannelParams * 4)()
        res = np.zeros((nch, n), dtype=np.int32)
        rc = self.lib.oracle_analyze_block(self.h, capi.planar_ptrs(pcm), n, C.byref(info), params, variants,
                                           capi.planar_ptrs(res))
        if rc != 0:
            raise RuntimeError("oracle_analyze_block -> %d" % rc)
        return info, [params[c] for c in range(nch)], [variants[v] for v in range(4)], res

    def analyze_channel(self, samples):
        """samples: one channel variant (already shifted / M-S combined). -> (params, residual, filtered)"""
        buf = np.ascontiguousarray(samples, dtype=np.int32).copy()
        res = np.zeros_like(buf)
        out = OracleChannelParams()
        rc = self.lib.oracle_analyze_channel(self.h, buf.ctypes.data_as(C.c_void_p), buf.size,
                                             res.ctypes.data_as(C.c_void_p), C.byref(out))
        if rc != 0:
            raise RuntimeError("oracle_analyze_channel -> %d" % rc)
        return out, res, buf


def oracle_decode(data):
    lib = oracle_lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    cfg = OracleConfig()
    n = C.c_uint32(0)
    sh = C.c_uint32(0)
    rc = lib.oracle_decode_header(data.ctypes.data_as(C.c_void_p), data.size, C.byref(cfg), C.byref(n), C.byref(sh))
    if rc != 0:
        raise RuntimeError("oracle_decode_header -> %d" % rc)
    out = np.zeros((cfg.num_channels, n.value), dtype=np.int32)
    rc = lib.oracle_decode_whole(data.ctypes.data_as(C.c_void_p), data.size, capi.planar_ptrs(out),
                                 cfg.num_channels, n.value)
    if rc != 0:
        raise RuntimeError("oracle_decode_whole -> %d" % rc)
    return out


def list_blocks(data):
    lib = oracle_lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    cap = 1 << 16
    t = np.zeros(cap, np.uint32); ns = np.zeros(cap, np.uint32); nb = np.zeros(cap, np.uint32)
    cnt = C.c_uint32(0)
    rc = lib.oracle_list_blocks(data.ctypes.data_as(C.c_void_p), data.size, t.ctypes.data_as(C.c_void_p),
                                ns.ctypes.data_as(C.c_void_p), nb.ctypes.data_as(C.c_void_p), cap, C.byref(cnt))
    if rc != 0:
        raise RuntimeError("oracle_list_blocks -> %d" % rc)
    k = cnt.value
    return list(zip(t[:k].tolist(), ns[:k].tolist(), nb[:k].tolist()))


def reference_encode_fresh(pcm, bits_per_sample=16, sampling_rate=48000, **cli):
    """`srla -e` semantics: the compiled reference in a FRESH process.

    The reference reads never-written words of its malloc'ed work area (the middle sample of an
    odd-length block, lpc.c:260-264; auto_corr[263..264] in the pitch search, lpc.c:1508-1510), so in
    a long-lived process its output for such inputs depends on heap history.  A fresh process (what
    the CLI is) gets zero pages, which is also what the oracle models."""
    import pickle
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "in.pkl")
        dst = os.path.join(d, "out.npy")
        with open(src, "wb") as f:
            pickle.dump(dict(pcm=pcm, bps=bits_per_sample, rate=sampling_rate, cli=cli), f)
        code = ("import sys,pickle,numpy as np; sys.path.insert(0,%r); import helpers; "
                "a=pickle.load(open(%r,'rb')); "
                "np.save(%r, helpers.reference_encoder().encode(a['pcm'], bits_per_sample=a['bps'], sampling_rate=a['rate'], **a['cli']))"
                % (os.path.join(ROOT, "tests"), src, dst))
        subprocess.check_call([sys.executable, "-c", code])
        return np.load(dst)


REF_TOOL = os.path.join(ROOT, "oracle", "_ref", "srla_ref")


def write_wav(path, pcm, rate, bps):
    """planar int32 [ch][n] -> a PCMWAVEFORMAT RIFF file (what the reference's tool reads, libs/wav/src/wav.c)"""
    import struct
    nch = pcm.shape[0]
    inter = pcm.T.reshape(-1)
    if bps == 8:
        body = (inter + 128).astype(np.uint8).tobytes()
    elif bps == 16:
        body = inter.astype("<i2").tobytes()
    else:
        u = inter.astype(np.int64) & 0xFFFFFF
        body = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
    b = bps // 8
    fmt = struct.pack("<HHIIHH", 1, nch, rate, rate * nch * b, nch * b, bps)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def reference_tool_encode(pcm, bits_per_sample=16, sampling_rate=48000, preset=4, max_block=4096, divisions=1,
                          lookahead_factor=4, ltp_order=0, svr_iterations=0):
    """`srla -e` itself: the reference's own tool (oracle/_ref/srla_ref, `make -C oracle ref_tool`), WAV file in, .srl out.
    Its encoder's work area is one fresh allocation of zero pages, the behaviour the oracle models for the words the
    reference reads without having written them."""
    import tempfile
    if not os.path.exists(REF_TOOL):
        subprocess.check_call("make -s ref_tool", cwd=os.path.join(ROOT, "oracle"), shell=True)
    with tempfile.TemporaryDirectory() as d:
        wav, srl = os.path.join(d, "in.wav"), os.path.join(d, "out.srl")
        write_wav(wav, pcm, sampling_rate, bits_per_sample)
        cmd = [REF_TOOL, "-e", "-m", str(preset), "-B", str(max_block), "-V", str(divisions), "-L", str(lookahead_factor), "-P", str(ltp_order)]
        if svr_iterations:
            cmd += ["--svr-filter-learning-iteration", str(svr_iterations)]
        subprocess.check_call(cmd + [wav, srl], stdout=subprocess.DEVNULL)
        return np.fromfile(srl, dtype=np.uint8)
