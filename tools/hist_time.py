import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, helpers
from srla_amd import capi
lib = capi.EncoderLib(helpers.PRODUCT_SO)
pcm = helpers.synth(helpers.MUSIC, 1, 48000, 2, 480000)
for cli in (dict(preset=4, max_block=1024, divisions=2, ltp_order=3), dict(preset=4, max_block=4095, divisions=0), dict(preset=4, max_block=1000, divisions=3)):
    lib.encode(np.ascontiguousarray(pcm[:, :50000]), **cli)
    t0 = time.perf_counter(); d = lib.encode(pcm, **cli); dt = time.perf_counter() - t0
    print(cli, "%.1f ms, %.1f Msamples/s" % (dt * 1e3, pcm.size / dt / 1e6), flush=True)
