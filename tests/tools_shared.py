"""Input construction shared by the golden generator's consumers (mirrors tools/gen_golden.py)."""
import json
import os

import numpy as np

import helpers


def edge_signal(name, nch, n, bps):
    full = (1 << (bps - 1)) - 1
    a = np.zeros((nch, n), dtype=np.int32)
    if name == "silence":
        pass
    elif name == "const_pos":
        a[:] = full
    elif name == "const_neg":
        a[:] = -full - 1
    elif name == "nyquist":
        a[:, 0::2] = full
        a[:, 1::2] = -full - 1
    elif name == "impulse":
        a[:, 0] = 1
        a[:, n // 2] = -1
    elif name == "one_silent":
        a[0] = helpers.synth(helpers.MUSIC, 77, 48000, 1, n, bps)[0]
    elif name == "sign_flip":
        s = helpers.synth(helpers.SINE, 1, 48000, 1, n, bps)[0]
        a[0] = s
        if nch > 1:
            a[1] = -s
    elif name == "white_full":
        a[:] = np.random.RandomState(99).randint(-full - 1, full + 1, size=(nch, n)).astype(np.int32)
    elif name == "lshift3":
        a[:] = helpers.synth(helpers.MUSIC, 78, 48000, nch, n, bps)
        a[:] = (a >> 3) << 3
    else:
        raise ValueError(name)
    return a


def make_input(spec):
    if "edge" in spec:
        return edge_signal(spec["edge"], spec["nch"], spec["n"], spec["bps"])
    return helpers.synth(spec["kind"], spec["seed"], spec["rate"], spec["nch"], spec["n"], spec["bps"])


STREAMS = json.load(open(os.path.join(helpers.GOLDEN, "streams.json")))["streams"]
KATS = json.load(open(os.path.join(helpers.GOLDEN, "kats.json")))


def stages():
    return np.load(os.path.join(helpers.GOLDEN, "stages.npz"))
