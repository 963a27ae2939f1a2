import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, helpers
from srla_amd import capi
lib = capi.EncoderLib(helpers.PRODUCT_SO)
pcm = helpers.synth(helpers.MUSIC, 97, 48000, 2, 150001)
import os
ONLY=os.environ.get("ONLY")
for name, clip, cli in [("even, two blocks", pcm[:, :131070], dict(preset=3, max_block=65535, divisions=0, svr_iterations=1)),
                        ("B40000 svr", pcm[:, :80000], dict(preset=3, max_block=40000, divisions=0, svr_iterations=1)),
                        ("B32768 svr", pcm[:, :65536], dict(preset=3, max_block=32768, divisions=0, svr_iterations=1)),
                        ("B32768 plain", pcm[:, :65536], dict(preset=3, max_block=32768, divisions=0)),
                        ("B16384 svr", pcm[:, :65536], dict(preset=3, max_block=16384, divisions=0, svr_iterations=1)),
                        ("odd", pcm, dict(preset=3, max_block=65535, divisions=0, svr_iterations=1))]:
    if ONLY and ONLY not in name: continue
    clip = np.ascontiguousarray(clip)
    print("running", name, flush=True)
    got = lib.encode(clip, **cli)
    want = helpers.Oracle(2, **cli).encode_whole(clip)
    print(name, "equal" if np.array_equal(got, want) else "MISMATCH", got.size, want.size, flush=True)
