"""The register-resident FFT schedule of srla_autocorr_w (srla_amd/csrc/autocorr_wave.hip) and the slot permutations chosen for
its transpositions, checked on the CPU: tools/fft_schedule_model.py runs the schedule in exact IEEE doubles against the
reference's Stockham transform (fft.c:71-136 as the oracle restates it) -- identical bits, every index formula, the pruned
inverse, the in-lane pairing of bins i and m - i."""
import os
import sys

import pytest

import helpers

sys.path.insert(0, os.path.join(helpers.ROOT, "tools"))
import fft_schedule_model as model  # noqa: E402
import lds_conflicts  # noqa: E402


@pytest.mark.parametrize("T", [16, 32, 64])
def test_schedule_computes_the_reference_transform_bit_for_bit(T):
    import random
    random.seed(7 + T)
    m = 32 * T
    model.check_pairing(T)
    x = [(random.uniform(-1, 1), random.uniform(-1, 1)) for _ in range(m)]
    for flag in (-1, 1):
        want = model.stockham(x, flag)
        for paired in (False, True):
            got = model.schedule(x, flag, T, paired)
            assert len(got) == m and all(got[k] == want[k] for k in range(m)), (T, flag, paired)
    want = model.stockham(x, 1)
    for need in (9, 33, 132):
        got = model.schedule(x, 1, T, True, need)
        assert all(k in got and got[k] == want[k] for k in range(need)), (T, need)


def test_output_slot_formula_is_the_inverse_of_the_digit_reversal():
    """k = rev4(v) + 256 wrev(c) <-> slot v C + c (the kernel stores pass 3's outputs by k and reads them by bin)"""
    for T in (32, 64, 128, 256):
        M, C = 16 * T, (16 * T) // 256
        seen = set()
        for v in range(256):
            for c in range(C):
                k = model.rev4(v) + 256 * model.wrev(c, C)
                assert 0 <= k < M and k not in seen
                seen.add(k)
        assert len(seen) == M


@pytest.mark.parametrize("T", [32, 64, 128])
def test_the_chosen_slot_permutations_are_conflict_free_in_the_bank_model(T):
    """sw1 / sw2 / sw3 of autocorr_wave.hip under tools/lds_conflicts.py's model of 16-byte LDS accesses: every store of a
    transposition at the conflict-free cost, every load within 1.5x of it"""
    sw1 = (lambda i: i ^ ((i >> 4) & 15)) if T < 256 else (lambda i: i)
    sw2 = lambda i: i ^ ((i >> 4) & (15 if T >= 256 else 7))
    sw3 = lambda i: i ^ ((i >> 4) & 1) ^ ((i >> 5) & 7)
    pats = lds_conflicts.patterns(T)
    for sw, parts in ((sw1, [("T1 write", "w"), ("T1 read / T2 write", "r")]), (sw2, [("T1 read / T2 write", "w"), ("T2 read", "r")]),
                      (sw3, [("T3 write", "w"), ("T3 read", "r")])):
        for name, kind in parts:
            rd, wr = lds_conflicts.total_cost(T, pats[name][1], kind, sw)
            if kind == "w":
                assert wr == 8.0, (T, name, wr)          # 8 LDS-array cycles: one per group of 8 lanes
            else:
                assert rd <= 6.0, (T, name, rd)          # 4 = conflict free
