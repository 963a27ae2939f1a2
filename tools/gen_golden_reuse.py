#!/usr/bin/env python3
"""Golden outputs for sequences of calls on ONE encoder handle (tests/reuse.py), made with the compiled reference
(oracle/_ref/libsrla_ref.so) through its C API, one fresh process per sequence (so that the handle's buffer starts as zero
pages, as a process's first handle does).

    python tools/gen_golden_reuse.py          # writes tests/golden/reuse_sequences.json

Also records, per call, whether the bytes differ from the same call made on a FRESH handle in a fresh process -- the calls for which
carrying the buffer from call to call is visible.  Runs only where /root/reference exists."""
import json
import os
import pickle
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
import reuse    # noqa: E402

WORKER = r"""
import sys, pickle
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import helpers, reuse
cli, steps = pickle.load(open(sys.argv[1], 'rb'))
lib = helpers.reference_encoder()
outs, enc = reuse.run_on_library(lib, cli, steps)
lib.destroy(enc)
pickle.dump([reuse.digest(v) for v in outs], open(sys.argv[2], 'wb'))
"""


def fresh(cli, steps):
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in.pkl"), os.path.join(d, "out.pkl")
        pickle.dump((cli, steps), open(src, "wb"))
        subprocess.check_call([sys.executable, "-c", WORKER % dict(tests=os.path.join(ROOT, "tests"), root=ROOT), src, dst])
        return pickle.load(open(dst, "rb"))


def main():
    assert helpers.have_reference(), "needs the compiled reference (oracle/_ref)"
    out = {}
    for name, (cli, steps) in list(reuse.SEQUENCES.items()) + list(reuse.COUNTED.items()):
        got = fresh(cli, steps)
        # the same call alone on a fresh handle, under the parameters in force at that point
        cur = {k: v for k, v in cli.items() if k != "config"}
        alone = []
        for st in steps:
            if st["api"] == "set":
                cur = dict(st["cli"])
                alone.append(None)
                continue
            alone.append(fresh(cur, [st])[0])
        calls = []
        for st, g, a in zip(steps, got, alone):
            e = dict(api=st["api"])
            if g is not None:
                e.update(g)
                e["input_sha256"] = helpers.sha256(reuse.make_input(st["input"]))
                if a is not None:
                    e["differs_from_fresh_handle"] = (a != g)
            calls.append(e)
        out[name] = calls
        n = sum(1 for c in calls if c.get("differs_from_fresh_handle"))
        print("%-44s %2d calls, %d differ from a fresh handle's" % (name, len(calls), n), flush=True)
    path = os.path.join(ROOT, "tests", "golden", "reuse_sequences.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
