import sys, time, ctypes as C, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, helpers
from srla_amd import capi
lib = capi.EncoderLib(helpers.PRODUCT_SO)
rate, nch, bps = 48000, 2, 16
n = 600 * rate
pcm = helpers.synth(helpers.MUSIC, 1000, rate, nch, n, bps)
cfg, par = capi.cli_setup(nch, bps, rate, preset=4, max_block=4096, divisions=1)
enc = lib.create(cfg); assert lib.set_parameter(enc, par) == capi.OK
cap = 2 * pcm.size * 2 + 4096
for name, out in (("pageable out", np.zeros(cap, np.uint8)), ("pinned out", torch.empty(cap, dtype=torch.uint8).pin_memory().numpy())):
    for pin_in in (False, True):
        src = pcm if not pin_in else torch.from_numpy(pcm).pin_memory().numpy()
        osz = C.c_uint32(0)
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            rc = lib.lib.SRLAEncoder_EncodeWhole(enc, capi.planar_ptrs(src), n, out.ctypes.data_as(C.c_void_p), cap, C.byref(osz), None)
            ts.append(time.perf_counter() - t0)
            assert rc == capi.OK
        print(name, "pinned in" if pin_in else "pageable in", ["%.1f ms" % (1e3 * t) for t in ts], "%.0f Msamples/s" % (n / min(ts) / 1e6))
