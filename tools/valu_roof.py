#!/usr/bin/env python3
"""The VALU issue roof of a kernel from its own instruction mix (round 6, roofline.int_valu).

    python tools/valu_roof.py profiles/r06/valu_rates.txt [lib.so] > profiles/r06/valu_roof.json

`valu_rates.txt` is what tools/probes/valu_rates.hip printed on an MI355X: cycles per wave-instruction and SIMD for the
instruction classes the wide kernels are made of.  This tool disassembles the library's gfx950 code object (llvm-objdump), takes
the STATIC histogram of VALU mnemonics of the kernels named below (they are almost straight-line code: unrolled per-sample loops;
the k-block loop of the matrix-pipe FIR and the order-dependent parts repeat, which a static count cannot weigh -- said in the
output), prices every mnemonic with its measured cycles (the four-wavefronts-per-SIMD row; a mnemonic that was not measured takes
the rate of its encoding class and is listed as assumed), and reports

    cycles_per_valu_instruction = sum(count x cycles) / sum(count)
    peak_tera_lane_ops = 64 lanes / cycles_per_valu_instruction x 1024 SIMDs x 2.4 GHz / 1e12

-- the rate at which THIS mix could issue if nothing but VALU issue limited it.  The data sheet's 78.6 T lane-ops/s (32 lanes per
clock and SIMD) is reached by no instruction; two-operand adds / shifts / logic come to 2.5 - 2.7 cycles, three-operand, packed,
multiply, dot, DPP and fp64 forms to 4.3 - 4.5."""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNELS = {
    "srla_residual_cost<2, true>": "_Z18srla_residual_costILi2ELb1EE",
    "srla_residual_cost<4, true>": "_Z18srla_residual_costILi4ELb1EE",
    "srla_autocorr<2, 256, 4096, true>": "_Z13srla_autocorrILi2ELi256ELi4096ELb1EE",
    "srla_autocorr<1, 256, 2048, true>": "_Z13srla_autocorrILi1ELi256ELi2048ELb1EE",
}
FULL, HALF = "two-operand integer / logic (assumed like v_add_u32)", "three-operand / packed / multiply / fp64 / DPP (assumed like v_add3_u32)"


def rates(path):
    """instruction name -> cycles per wave-instruction and SIMD (the row with the most wavefronts per SIMD)"""
    out = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        m = re.match(r"(.+?)\s+(\d)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m:
            out[m.group(1).strip()] = float(m.group(4))          # later rows (more wavefronts) overwrite earlier ones
    return out


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="valu_roof_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp, check=False)
        text = ""
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                text += subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        return text
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def histogram(text, prefix):
    hist, on = collections.Counter(), False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            on = m.group(1).startswith(prefix)
            continue
        if on:
            m = re.match(r"^\s+([a-z_0-9]+)\s", line)
            if m:
                hist[m.group(1)] += 1
    return hist


def price(mn, table):
    """(cycles, measured?, class) of one VALU mnemonic"""
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    dpp = mn.endswith("dpp")
    if dpp and base in ("v_mov_b32",):
        return table["v_mov_b32 dpp row_shr"], True, "dpp"
    if dpp:
        return table["v_add_u32 dpp row_shr"], True, "dpp"
    if base.startswith("v_mfma"):
        return None, True, "matrix pipe (issues beside the VALU)"
    if base == "v_cndmask_b32" and mn.endswith("_e32") and "v_cmp_gt_u32 + v_cndmask_b32 (pair)" in table:
        # the VOP2 form behind a compare: what the pair test leaves after the compare's own cycles (alone, reading a VCC nobody wrote
        # in the loop, the probe's v_cndmask_b32 takes 23 cycles -- an artefact of the probe, not of the kernels)
        return 2.0 * table["v_cmp_gt_u32 + v_cndmask_b32 (pair)"] - table["v_cmp_gt_u32"], True, "measured (pair)"
    names = {"v_cndmask_b32": "v_cndmask_b32 (sgpr pair)", "v_sub_u32": "v_sub_u32", "v_pk_sub_u16": "v_pk_sub_u16 clamp"}
    key = names.get(base, base)
    if key in table:
        return table[key], True, "measured"
    if base.endswith("_f64") or base.startswith("v_cvt_f64") or base.startswith("v_cvt_i32_f64") or base.startswith("v_cvt_u32_f64"):
        return table["v_add_f64"], False, "fp64 (as v_add_f64)"
    if base.startswith("v_cmp"):
        return table.get("v_cmp_gt_u32", table["v_add_u32"]), False, "compare (as v_cmp_gt_u32)"
    if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32", "v_nop", "v_accvgpr_read_b32", "v_accvgpr_write_b32"):
        return table["v_add_u32"], False, FULL
    two = ("v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
           "v_mov_b32", "v_not_b32", "v_bfrev_b32")
    if base in two:
        return table["v_add_u32"], False, FULL
    return table["v_add3_u32"], False, HALF


def main():
    table = rates(sys.argv[1])
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "srla_amd", "libsrla_mi355x.so")
    text = disassemble(lib)
    out = {"_source": "tools/valu_roof.py over %s and the gfx950 code object of %s (static instruction histogram)" % (os.path.relpath(sys.argv[1], ROOT) if os.path.isabs(sys.argv[1]) else sys.argv[1], os.path.basename(lib)),
           "_clock_ghz": 2.4, "_simds": 1024,
           "_note": "static counts: loops (the k-blocks of the matrix-pipe FIR, order-dependent parts) are counted once; the two kernels are almost straight-line code"}
    for name, prefix in KERNELS.items():
        hist = histogram(text, prefix)
        valu = {k: v for k, v in hist.items() if k.startswith("v_")}
        if not valu:
            continue
        cyc, n, assumed, mfma, classes = 0.0, 0, 0, 0, collections.Counter()
        rows = []
        for mn, cnt in sorted(valu.items(), key=lambda kv: -kv[1]):
            c, measured, cls = price(mn, table)
            if c is None:
                mfma += cnt
                continue
            cyc += c * cnt
            n += cnt
            assumed += 0 if measured else cnt
            classes[round(c, 1)] += cnt
            rows.append([mn, cnt, round(c, 2), "measured" if measured else "assumed: " + cls])
        avg = cyc / n
        out[name] = {"valu_instructions_static": n, "mfma_instructions_static": mfma, "assumed_rate_share": round(assumed / n, 4),
                     "cycles_per_valu_instruction": round(avg, 3), "peak_tera_lane_ops": round(64.0 / avg * 1024 * 2.4e9 / 1e12, 2),
                     "histogram_by_cycles": {str(k): v for k, v in sorted(classes.items())}, "top_mnemonics": rows[:24],
                     "other_instructions_static": {k: v for k, v in sorted(hist.items(), key=lambda kv: -kv[1]) if not k.startswith("v_")and v >= 8}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
