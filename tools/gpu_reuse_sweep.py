#!/usr/bin/env python3
"""Randomised sweep of CALL SEQUENCES on one handle, on an MI355X: library bytes vs ONE oracle handle (which keeps the calculator's
FFT buffer from call to call like the reference, oracle/srla_oracle.c: fftbuf; pinned against the compiled reference by
tests/test_handle_reuse.py).

    python tools/gpu_reuse_sweep.py [sequences] [seed]

Every sequence: random parameters (regular and history regimes, LTP, now and then SVR; 8 / 16 / 24 bit), 3-9 calls of EncodeWhole /
SRLAMI355X_EncodeWholeDevice / ComputeBlockSize / EncodeBlock / EncodeOptimalPartitionedBlock (and SRLAMI355X_EncodeBatch, which must
neither read nor change the handle's buffer) with lengths around the block and window sizes (odd and even, clips
of less than a window, streams of several windows), now and then SetEncodeParameter in between, inputs of every kind incl.
identical channels, digital silence at the end and an offset left shift.
A call the library counts as SRLAMI355X_NONIDENTICAL_HANDLE_HISTORY may differ (reported separately); any other difference is a
mismatch (exit status 1)."""
import ctypes as C
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
import bench    # noqa: E402
from srla_amd import capi  # noqa: E402

HANDLE_HISTORY = 4


def sequences(count, seed):
    rnd = random.Random(seed)
    for case in range(count):
        nch = rnd.choice([1, 2, 2, 2, 3])
        preset = rnd.choice([1, 2, 3, 4, 4, 4, 5])
        order = [0, 8, 16, 32, 64, 128, 255][preset]
        shape = rnd.choice(["regular", "regular", "odd", "odd_v0", "ltp_short", "regular_ltp", "v0"])
        ltp = 0
        if shape == "regular":
            max_block = rnd.choice([1024, 2048, 4096, 4096, 8192]); divisions = rnd.choice([1, 1, 2, 3])
        elif shape == "regular_ltp":
            max_block = rnd.choice([2048, 4096, 8192]); divisions = rnd.choice([1, 2]); ltp = rnd.choice([1, 3])
        elif shape == "odd":
            max_block = rnd.choice([1000, 2000, 3000, 5000]); divisions = 3
        elif shape == "odd_v0":
            max_block = rnd.choice([999, 2047, 4095, 4097]); divisions = 0; ltp = rnd.choice([0, 0, 1, 3])
        elif shape == "ltp_short":
            max_block = rnd.choice([512, 1024]); divisions = rnd.choice([1, 2]); ltp = rnd.choice([1, 3])
        else:
            max_block = rnd.choice([1024, 4096]); divisions = 0; ltp = rnd.choice([0, 3])
        min_block = max_block >> divisions
        if order > min_block:
            preset = 2; order = 16
        look = rnd.choice([2, 4, 4]) * max_block if divisions else 4 * max_block
        cli = dict(preset=preset, max_block=max_block, divisions=divisions, ltp_order=ltp, lookahead_factor=look // max_block)
        if rnd.random() < 0.12:
            cli["svr_iterations"] = rnd.choice([1, 2])
        window = look if divisions else max_block
        bps = rnd.choice([16, 16, 16, 8, 24])
        steps = []
        for k in range(rnd.randint(3, 9)):
            if k > 0 and rnd.random() < 0.15:
                # SetEncodeParameter on the way: everything but the maximum block (the reference searches up to the one the encoder was
                # created for) and the channels
                d2 = rnd.choice([0, 1, 2, 3])
                if max_block % (1 << d2) != 0 or (max_block >> d2) < 32:
                    d2 = 0
                p2 = rnd.choice([1, 2, 3, 4, 4, 5])
                if [0, 8, 16, 32, 64, 128, 255][p2] > (max_block >> d2):
                    p2 = 2
                l2 = rnd.choice([0, 0, 1, 3])
                if l2 and max_block <= 256:
                    l2 = 0
                divisions, min_block = d2, max_block >> d2
                look = rnd.choice([2, 4]) * max_block if divisions else 4 * max_block
                window = look if divisions else max_block
                new = dict(preset=p2, max_block=max_block, divisions=d2, ltp_order=l2, lookahead_factor=look // max_block)
                if rnd.random() < 0.1:
                    new["svr_iterations"] = 1
                steps.append(dict(api="set", cli=new))
                cli_now = new
            api = rnd.choice(["whole", "whole", "whole", "block", "size", "partitioned", "whole_device", "batch"])
            if api in ("block", "size"):
                n = rnd.choice([max_block, max_block - 1, rnd.randint(1, max_block), rnd.randint(1, max_block) | 1, min_block + 1])
            elif api == "partitioned":
                n = rnd.choice([window, window - 1, rnd.randint(1, window), rnd.randint(1, window) | 1])
                if not divisions:
                    api = "block"; n = min(n, max_block)
            else:
                n = rnd.choice([rnd.randint(1, window), rnd.randint(1, window) | 1, window + rnd.randint(1, window), rnd.randint(2, 6) * window + rnd.randint(0, window),
                                (rnd.randint(2, 5) * window + rnd.randint(0, window)) | 1, rnd.randint(1, min_block) | 1])
            n = max(2, min(n, 60000 if ("svr_iterations" in cli or any("svr_iterations" in x.get("cli", {}) for x in steps)) else 200000))
            if api in ("block", "size"):
                n = min(n, max_block)
            kind = rnd.choice([helpers.MUSIC, helpers.VARIED, helpers.VARIED, helpers.NOISE, helpers.SINE])
            twist = rnd.choice(["none", "none", "none", "identical", "silent_end", "silent_all", "impulses", "shifted"])
            steps.append(dict(api=api, n=n, kind=kind, seed=seed * 1000 + case * 16 + k, twist=twist))
        yield case, nch, bps, cli, steps


def make_input(st, nch, bps=16):
    a = helpers.synth(st["kind"], st["seed"], 48000, nch, st["n"], bps)
    r = np.random.RandomState(st["seed"] % (1 << 31))
    t = st["twist"]
    if t == "identical":
        a[1:] = a[0]
    elif t == "silent_end":
        a[:, int(r.randint(0, a.shape[1])):] = 0
    elif t == "silent_all":
        a[:] = 0
    elif t == "shifted":
        a[:] = (a >> 3) << 3                       # an offset left shift: EncodeWhole leaves it in the handle's header for the block calls
    elif t == "impulses":
        keep = np.zeros(a.shape[1], dtype=bool)
        keep[::int(r.randint(40, 900))] = True
        a[:, ~keep] = 0
        a[1:] = a[0]
    return np.ascontiguousarray(a)


def oracle_call(o, api, pcm):
    if api == "whole_device":
        return o.encode_whole(pcm)
    if api == "size":
        return o.compute_block_size(pcm)
    if api == "block":
        return o.encode_block(pcm)
    if api == "partitioned":
        pos, chunks = 0, []
        for p in o.search_partitions(pcm):
            chunks.append(o.encode_block(np.ascontiguousarray(pcm[:, pos:pos + p])))
            pos += p
        return np.concatenate(chunks)
    return o.encode_whole(pcm)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    lib.lib.SRLAMI355X_GetStats.argtypes = [C.c_void_p, C.POINTER(bench.Stats), C.c_int]
    lib.lib.SRLAMI355X_EncodeWholeDevice.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
    calls = mismatches = flagged_calls = flagged_differ = seqs = 0
    for case, nch, bps, cli, steps in sequences(count, seed):
        cfg, par = capi.cli_setup(nch, bps, 48000, **cli)
        # created for everything the sequence's parameter changes may ask for (same maximum block)
        cfg.min_num_samples_per_block = min([par.min_num_samples_per_block] + [cli["max_block"] >> x["cli"]["divisions"] for x in steps if x["api"] == "set"])
        cfg.max_num_lookahead_samples = 4 * cli["max_block"]
        enc = lib.create(cfg)
        assert enc and lib.set_parameter(enc, par) == capi.OK
        o = helpers.Oracle(nch, bits_per_sample=bps, **cli)
        seqs += 1
        for k, st in enumerate(steps):
            if st["api"] == "set":
                assert lib.set_parameter(enc, capi.cli_setup(nch, bps, 48000, **st["cli"])[1]) == capi.OK
                o.set_parameter(**st["cli"])
                continue
            pcm = make_input(st, nch, bps)
            before = bench.Stats(); lib.lib.SRLAMI355X_GetStats(enc, C.byref(before), 0)
            if st["api"] == "batch":
                # EncodeBatch models one fresh handle per stream: the oracle's bytes of a fresh handle, and the handle's buffer untouched
                other = np.ascontiguousarray(pcm[:, :max(2, pcm.shape[1] // 2)])
                rc, outs, res = capi.encode_batch(lib, enc, [pcm, other])
                cur = dict(cli)
                for x in steps[:k]:
                    if x["api"] == "set":
                        cur = dict(x["cli"])
                ok = rc == capi.OK
                for a, got in zip((pcm, other), outs):
                    want = helpers.Oracle(nch, bits_per_sample=bps, **cur).encode_whole(a)
                    ok = ok and got is not None and got.size == want.size and np.array_equal(got, want)
                calls += 1
                if not ok:
                    mismatches += 1
                    print("MISMATCH sequence %d call %d (batch): %s  %s  nch %d" % (case, k, cur, st, nch), flush=True)
                    break
                continue
            if st["api"] == "whole_device":
                import torch
                d = torch.from_numpy(pcm).cuda(); torch.cuda.synchronize()
                buf = np.zeros(2 * pcm.size * 4 + 4096, np.uint8); out = C.c_uint32(0)
                rc = lib.lib.SRLAMI355X_EncodeWholeDevice(enc, C.c_void_p(d.data_ptr()), pcm.shape[1], pcm.shape[1], buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(out), None)
                got = buf[:out.value].copy()
                del d
            elif st["api"] == "size":
                rc, got = lib.compute_block_size(enc, pcm)
            elif st["api"] == "block":
                rc, got = lib.encode_block(enc, pcm)
            elif st["api"] == "partitioned":
                rc, got = lib.encode_partitioned(enc, pcm)
            else:
                rc, got = lib.encode_whole(enc, pcm)
            after = bench.Stats(); lib.lib.SRLAMI355X_GetStats(enc, C.byref(after), 0)
            flagged = after.num_nonidentical_calls > before.num_nonidentical_calls
            try:
                want = oracle_call(o, st["api"], pcm)
            except Exception as e:                      # (the oracle follows the reference into its failures)
                print("sequence %d call %d: the oracle fails (%s); the rest of the sequence is skipped  %s %s" % (case, k, e, cli, st), flush=True)
                break
            calls += 1
            same = rc == capi.OK and (got == want if st["api"] == "size" else (got.size == want.size and np.array_equal(got, want)))
            if flagged:
                flagged_calls += 1
                if not same:
                    flagged_differ += 1
                    print("sequence %d call %d: counted HANDLE_HISTORY and differs  %s %s" % (case, k, cli, st), flush=True)
                    break                               # (from here on the two handles' buffers are different ones)
            elif not same:
                mismatches += 1
                print("MISMATCH sequence %d call %d rc %d: %s  %s  nch %d  steps so far %s" % (case, k, rc, cli, st, nch, steps[:k + 1]), flush=True)
                break
        lib.destroy(enc)
    print("sequences %d, calls compared %d, mismatches %d; calls counted HANDLE_HISTORY %d, of which %d differ  (seed %d)"
          % (seqs, calls, mismatches, flagged_calls, flagged_differ, seed))
    sys.exit(1 if mismatches else 0)


if __name__ == "__main__":
    main()
