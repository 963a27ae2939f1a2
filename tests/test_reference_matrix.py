"""The reference's own integration matrix (/root/reference/test/srla_encode_decode/main.cpp:393-767: ten generators x
{1, 2, 8 ch} x {8, 16, 24 bit} x min 512 / max 1024 x look-ahead {2048, 1536} x {plain ; LTP 3 + 6 SVR iterations}, preset 0,
8 500 samples, encoded twice on one handle) and explicit (minimum, maximum, look-ahead) triples the `srla` tool cannot
express, against golden streams made from the compiled reference (tools/gen_golden_matrix.py ->
tests/golden/matrix_streams.json).  CPU: the inputs re-create, the oracle reproduces the goldens.  GPU: the library's bytes."""
import json
import os

import numpy as np
import pytest

import helpers
import refmatrix

GOLD = json.load(open(os.path.join(helpers.GOLDEN, "matrix_streams.json")))
MATRIX = GOLD["matrix"]
TRIPLES = GOLD["triples"]


@pytest.fixture(scope="module")
def matrix_inputs():
    """all 360 inputs, generated in the reference's order (one srand(0), rand() running on through the cases)"""
    made = refmatrix.generate_all()
    assert len(made) == len(MATRIX) == 360
    for (case, pcm), gold in zip(made, MATRIX):
        assert case["name"] == gold["name"]
        assert helpers.sha256(pcm) == gold["input_sha256"], "input %s differs from the one the golden was made from (C library rand / libm?)" % case["name"]
    return [pcm for _, pcm in made]


def _triple_input(t):
    sp = t["input"]
    pcm = helpers.synth_spec(sp)
    assert helpers.sha256(pcm) == t["input_sha256"]
    return pcm


def _oracle_for(case):
    return helpers.Oracle(case["nch"], bits_per_sample=case["bps"], sampling_rate=refmatrix.RATE, preset=case["preset"],
                          max_block=refmatrix.MAX_BLOCK, min_block=refmatrix.MIN_BLOCK, lookahead=case["lookahead"],
                          ltp_order=case["ltp_order"], svr_iterations=case["svr_iterations"])


# ------------------------------------------------------------------------------------------------------------------- CPU
def test_matrix_is_the_reference_tests_shape():
    assert len(MATRIX) == 360 and len(refmatrix.parameter_sets()) == 36
    assert {c["lookahead"] for c in MATRIX} == {2048, 1536}
    assert {(c["ltp_order"], c["svr_iterations"], c["preset"]) for c in MATRIX} == {(0, 0, 0), (3, 6, 0)}
    # the same handle encodes the same input to the same bytes (no history regime in the matrix)
    assert all(c["srl_sha256"] == c["second_srl_sha256"] for c in MATRIX)


def test_oracle_reproduces_the_reference_matrix(matrix_inputs):
    for gold, pcm in zip(MATRIX, matrix_inputs):
        data = _oracle_for(gold).encode_whole(pcm)
        assert data.size == gold["srl_size"] and helpers.sha256(data) == gold["srl_sha256"], gold["name"]


@pytest.mark.parametrize("t", [t for t in TRIPLES if t["input"].get("count", t["input"]["n"]) * t["input"]["nch"] <= 130000 and not (t["cli"].get("svr_iterations") and t["cli"]["preset"] > 0)],
                         ids=lambda t: t["name"])
def test_oracle_reproduces_the_reference_on_explicit_triples(t):
    pcm = _triple_input(t)
    sp = t["input"]
    data = helpers.Oracle(sp["nch"], bits_per_sample=sp["bps"], sampling_rate=sp["rate"], **t["cli"]).encode_whole(pcm)
    assert data.size == t["srl_size"] and helpers.sha256(data) == t["srl_sha256"]


# ------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("generator", refmatrix.GENERATORS)
def test_reference_matrix_on_the_mi355x(product, matrix_inputs, generator):
    bad = []
    for gold, pcm in zip(MATRIX, matrix_inputs):
        if gold["generator"] != generator:
            continue
        first, second = refmatrix.encode_twice(product, gold, pcm)
        if not (first.size == gold["srl_size"] and helpers.sha256(first) == gold["srl_sha256"]):
            bad.append((gold["name"], "first", int(first.size), gold["srl_size"]))
        if not (second.size == gold["second_srl_size"] and helpers.sha256(second) == gold["second_srl_sha256"]):
            bad.append((gold["name"], "second", int(second.size), gold["second_srl_size"]))
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("t", TRIPLES, ids=lambda t: t["name"])
def test_explicit_triples_on_the_mi355x(product, t):
    pcm = _triple_input(t)
    sp = t["input"]
    got = product.encode(pcm, bits_per_sample=sp["bps"], sampling_rate=sp["rate"], **t["cli"])
    assert got.size == t["srl_size"] and helpers.sha256(got) == t["srl_sha256"]
