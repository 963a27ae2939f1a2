"""`srla -e` on the MI355X: the encode half of the reference's command line tool
(tools/srla_codec/srla_codec.c:75-158, options :38-62) on top of libsrla_mi355x.so.

    python -m srla_amd.cli -e [-m 4] [-B 4096] [-V 1] [-L 4] [-P 0] in.wav out.srl
    torchrun --nproc-per-node 8 -m srla_amd.cli -e ... --corpus IN_DIR --out OUT_DIR      # one rank per GPU

Same defaults, same parameter set-up, same output buffer rule (2 x the input file size) and the same summary line
as the reference; decoding stays with the reference's tool (this library is the encoder only)."""
import argparse
import os
import sys

import numpy as np

from . import capi, wavio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "srla_amd", "libsrla_mi355x.so")


def encode_file(lib, in_path, out_path, cli, device_index=None):
    pcm, rate, bps = wavio.read_wav(in_path)
    cfg, par = capi.cli_setup(pcm.shape[0], bps, rate, **cli)           # srla_codec.c:91-116
    enc = lib.create(cfg)
    if not enc:
        raise RuntimeError("Failed to create encoder handle.")
    try:
        rc = lib.set_parameter(enc, par)
        if rc != capi.OK:
            raise RuntimeError("Failed to set encode parameter: %d" % rc)
        in_size = os.path.getsize(in_path)
        rc, data = lib.encode_whole(enc, pcm, cap=2 * in_size)          # srla_codec.c:125-129
        if rc != capi.OK:
            raise RuntimeError("Failed to encode data: %d" % rc)
    finally:
        lib.destroy(enc)
    with open(out_path, "wb") as f:
        f.write(data.tobytes())
    return in_size, int(data.size)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="srla_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("-e", "--encode", action="store_true", help="Encode mode")
    ap.add_argument("-d", "--decode", action="store_true", help="(not available: use the reference's srla -d)")
    ap.add_argument("-m", "--mode", type=int, default=4)
    ap.add_argument("-L", "--lookahead-sample-factor", type=int, default=4)
    ap.add_argument("-B", "--max-block-size", type=int, default=4096)
    ap.add_argument("-V", "--variable-block-divisions", type=int, default=1)
    ap.add_argument("-P", "--long-term-prediction", type=int, default=0)
    ap.add_argument("--svr-filter-learning-iteration", type=int, default=0)
    ap.add_argument("--corpus", help="encode every .wav under this directory (sharded over the ranks of torchrun)")
    ap.add_argument("--out", help="output directory for --corpus")
    ap.add_argument("files", nargs="*")
    a = ap.parse_args(argv)
    if a.decode:
        print("srla_amd.cli: decoding is not part of this library; use the reference's `srla -d`.", file=sys.stderr)
        return 1
    if not a.encode:
        ap.print_usage()
        return 1
    if a.mode >= 7:
        print("srla_amd.cli: encode preset number is out of range.", file=sys.stderr)
        return 1
    cli = dict(preset=a.mode, max_block=a.max_block_size, divisions=a.variable_block_divisions,
               lookahead_factor=a.lookahead_sample_factor, ltp_order=a.long_term_prediction,
               svr_iterations=a.svr_filter_learning_iteration)
    lib = capi.EncoderLib(LIB)
    if a.corpus:
        from . import corpus
        return corpus.main_cli(lib, a.corpus, a.out or a.corpus, cli)
    if len(a.files) != 2:
        print("srla_amd.cli: input and output file must be specified.", file=sys.stderr)
        return 1
    try:
        in_size, out_size = encode_file(lib, a.files[0], a.files[1], cli)
    except (RuntimeError, wavio.WavError, OSError) as e:
        print(str(e), file=sys.stderr)
        print("srla_amd.cli: failed to encode %s." % a.files[0], file=sys.stderr)
        return 1
    print("finished: %d -> %d (%6.2f %%) " % (in_size, out_size, 100.0 * out_size / in_size))
    return 0


if __name__ == "__main__":
    sys.exit(main())
