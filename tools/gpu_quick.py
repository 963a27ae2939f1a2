"""Quick on-GPU bring-up check: product library vs oracle on short clips, with stage-level diffs."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import *
from srla_amd import capi

lib = capi.EncoderLib(PRODUCT_SO)
print(C.c_char_p.in_dll if False else "", flush=True)

def probe(pcm, n, **cli):
    cfg, par = capi.cli_setup(pcm.shape[0], 16, 48000, **cli)
    enc = lib.create(cfg); assert enc
    assert lib.set_parameter(enc, par) == 0
    nch = pcm.shape[0]; nv = nch + (2 if nch >= 2 else 0)
    blk = np.ascontiguousarray(pcm[:, :n])
    recs = np.zeros(nv * 1344, np.uint8); res = np.zeros((nv, n), np.int32); dbg = np.zeros((nv, 1040))
    fn = lib.lib.SRLAMI355X_ProbeBlock
    fn.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = fn(enc, capi.planar_ptrs(blk), n, recs.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p), dbg.ctypes.data_as(C.c_void_p))
    lib.destroy(enc)
    assert rc == 0, rc
    return recs.reshape(nv, 1344), res, dbg

def rec_fields(r):
    w = r[:64].view(np.int32)
    return dict(prev=int(w[0]), pcoef=int(w[1]), order=int(w[2]), rshift=int(w[3]), use_sum=int(w[4]), period=int(w[5]),
                ltp=[int(w[6]), int(w[7]), int(w[8])], code_length=int(w[9]), type=int(w[10]), porder=int(w[11]), res_bits=int(w[12]), flags=int(w[13]),
                coef=r[64:64+int(w[2])].view(np.int8).tolist())

def compare_block(pcm, n, **cli):
    recs, res, dbg = probe(pcm, n, **cli)
    o = Oracle(pcm.shape[0], **cli)
    blk = np.ascontiguousarray(pcm[:, :n])
    info, chosen, variants, ores = o.analyze_block(blk)
    nch = pcm.shape[0]
    order_map = list(range(nch)) + ([2, 3] if nch >= 2 else [])   # device: plain..., M, S ; oracle variants: L,R,M,S
    ok = True
    for v in range(recs.shape[0]):
        d = rec_fields(recs[v])
        ov = variants[v].as_dict() if nch >= 2 else chosen[0].as_dict()
        for a, b in (("prev", "preemph_prev"), ("pcoef", "preemph_coef"), ("order", "lpc_order"), ("rshift", "lpc_rshift"), ("use_sum", "use_sum"),
                     ("period", "ltp_period"), ("code_length", "code_length"), ("type", "res_code_type"), ("porder", "res_porder"), ("res_bits", "res_bits")):
            if d[a] != ov[b]:
                ok = False; print("  variant", v, a, "gpu", d[a], "oracle", ov[b])
        if d["coef"] != ov["lpc_coef"]:
            ok = False; print("  variant", v, "coefs differ", d["coef"][:8], ov["lpc_coef"][:8])
        if d["period"] and d["ltp"][:3] != ov["ltp_coef"]:
            ok = False; print("  variant", v, "ltp coefs", d["ltp"], ov["ltp_coef"])
    return ok

t0 = time.time()
pcm = synth(MUSIC, 1, 48000, 2, 48000 * 2)
print("probe 4096 -m4:", compare_block(pcm, 4096, preset=4, max_block=4096, divisions=1), flush=True)
print("probe 2048 -m4:", compare_block(pcm, 2048, preset=4, max_block=4096, divisions=1), flush=True)
print("probe 4096 -m2 P3:", compare_block(pcm, 4096, preset=2, max_block=4096, divisions=1, ltp_order=3), flush=True)
print("probe 3000 -m4:", compare_block(pcm, 3000, preset=4, max_block=4096, divisions=1), flush=True)
print("probe 8192 -m4 P3:", compare_block(pcm, 8192, preset=4, max_block=8192, divisions=2, ltp_order=3), flush=True)

nbad = 0
for kind in (MUSIC, VARIED):
    for nch in (2, 1):
        pcm = synth(kind, 5, 48000, nch, 48000 * 3 + 1000)
        for cli in [dict(preset=4, max_block=4096, divisions=1), dict(preset=0, max_block=2048, divisions=1), dict(preset=2, max_block=4096, divisions=0),
                    dict(preset=4, max_block=4096, divisions=2), dict(preset=4, max_block=4096, divisions=2, ltp_order=3), dict(preset=4, max_block=8192, divisions=2, ltp_order=3)]:
            t = time.time(); g = lib.encode(pcm, **cli); tg = time.time() - t
            o = Oracle(nch, **cli); d = o.encode_whole(pcm)
            same = g.size == d.size and bool((g == d).all())
            if not same:
                nbad += 1
                try:
                    back = oracle_decode(g); rt = bool((back == pcm).all())
                except Exception as e:
                    rt = "decode failed: %s" % e
                bg = list_blocks(g) if isinstance(rt, bool) else None; bd = list_blocks(d)
                print("MISMATCH", kind, nch, cli, g.size, d.size, "roundtrip", rt)
                if bg:
                    for i, (x, y) in enumerate(zip(bg, bd)):
                        if x != y: print("   first differing block", i, x, y); break
            else:
                print("ok", kind, nch, cli, g.size, "%.3fs" % tg, flush=True)
print("mismatches:", nbad, "elapsed %.1f" % (time.time() - t0))

