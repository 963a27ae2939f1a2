import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, ctypes as C
import bench, helpers
from srla_amd import capi
product = capi.EncoderLib(helpers.PRODUCT_SO)
def stats(enc):
    st = bench.Stats(); fn = product.lib.SRLAMI355X_GetStats; fn.argtypes = [C.c_void_p, C.POINTER(bench.Stats), C.c_int]; fn(enc, C.byref(st), 0); return st
cli = dict(preset=2, max_block=2048, divisions=1, svr_iterations=3)
for kind, n in ((helpers.MUSIC, 100000), (helpers.VARIED, 98304)):
    pcm = helpers.synth(kind, 33, 48000, 2, n)
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg); assert product.set_parameter(enc, par) == 0
    rc, got = product.encode_whole(enc, pcm)
    st = stats(enc); product.destroy(enc)
    want = helpers.Oracle(2, **cli).encode_whole(pcm)
    print(os.environ.get("SRLA_MI355X_TIE_TEST"), kind, "equal", np.array_equal(got, want), "ties", st.num_tie_items, "svr", st.num_svr_tie_items, "resolved", st.num_tie_resolved, "overrides", st.num_tie_overrides, "restarts", st.num_restarts, flush=True)
