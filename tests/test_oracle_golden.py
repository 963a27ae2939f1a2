"""The CPU oracle against the golden vectors generated from the compiled reference
(tools/gen_golden.py).  This is what pins the oracle on machines without /root/reference."""
import json
import os

import numpy as np
import pytest

import helpers
from tools_shared import make_input, STREAMS

FAST = [s for s in STREAMS if s["input"]["n"] <= 300000]
SLOW = [s for s in STREAMS if s["input"]["n"] > 300000]


def _run(case):
    pcm = make_input(case["input"])
    assert helpers.sha256(pcm) == case["input_sha256"], "synthetic input differs from the one the golden was made from"
    o = helpers.Oracle(pcm.shape[0], bits_per_sample=case["input"]["bps"], sampling_rate=case["input"].get("rate", 48000), **case["cli"])
    data = o.encode_whole(pcm)
    assert data.size == case["srl_size"]
    assert helpers.sha256(data) == case["srl_sha256"]
    if "file" in case:
        want = np.fromfile(os.path.join(helpers.GOLDEN, case["file"]), dtype=np.uint8)
        assert np.array_equal(data, want)
    return pcm, data


@pytest.mark.parametrize("case", FAST, ids=[c["name"] for c in FAST])
def test_oracle_reproduces_reference_stream(case):
    pcm, data = _run(case)
    assert np.array_equal(helpers.oracle_decode(data), pcm)
    blocks = helpers.list_blocks(data)
    assert len(blocks) == case["num_blocks"]
    assert sum(1 for b in blocks if b[0] == 2) == case["raw_blocks"]
    assert sum(1 for b in blocks if b[0] == 1) == case["silent_blocks"]


@pytest.mark.parametrize("case", SLOW, ids=[c["name"] for c in SLOW])
def test_oracle_reproduces_reference_stream_full_size(case):
    _run(case)


def test_committed_streams_decode_to_their_inputs():
    for case in STREAMS:
        if "file" not in case:
            continue
        data = np.fromfile(os.path.join(helpers.GOLDEN, case["file"]), dtype=np.uint8)
        assert helpers.sha256(data) == case["srl_sha256"]
        assert np.array_equal(helpers.oracle_decode(data), make_input(case["input"]))
