"""bench.py's roofline book-keeping: every kernel the library can launch, and every kernel a committed profile saw, belongs to a
stage of bench.STAGES -- a stage is priced with what ran (profiles/pmc_summary.json), never with a list that went stale."""
import csv
import glob
import json
import os
import re

import bench
import helpers

ROOT = helpers.ROOT


def _library_kernels():
    names = set()
    for src in sorted(glob.glob(os.path.join(helpers.ROOT, "srla_amd", "csrc", "*.hip"))):
        text = open(src).read()
        for m in re.finditer(r"__global__[^;{]*?\b(srla_[a-z0-9_]+)\s*\(", text, re.S):
            names.add(m.group(1))
    return names


def test_every_kernel_of_the_library_has_a_stage():
    kernels = _library_kernels()
    assert len(kernels) >= 20 and "srla_residual_cost" in kernels and "srla_lpc_errvars_lean" in kernels
    for k in sorted(kernels):
        # templates appear as name<...> to the profiler
        assert bench.kernel_stage(k) or bench.kernel_stage(k + "<1>"), "kernel %s belongs to no stage of bench.STAGES" % k


def test_every_kernel_of_the_committed_profiles_has_a_stage():
    pmc = json.load(open(os.path.join(helpers.ROOT, "profiles", "pmc_summary.json")))
    seen = 0
    for cfg, entry in pmc.items():
        for k in entry:
            if k.startswith("_"):
                continue
            seen += 1
            assert bench.kernel_stage(k), "profiles/pmc_summary.json[%s]: kernel %s belongs to no stage" % (cfg, k)
    for path in glob.glob(os.path.join(helpers.ROOT, "profiles", "r0[3-9]", "*", "kernel_stats.csv")):
        for r in csv.DictReader(open(path)):
            seen += 1
            assert bench.kernel_stage(r["Name"]), "%s: kernel %s belongs to no stage" % (path, r["Name"])
    assert seen > 50


def test_no_kernel_belongs_to_two_stages():
    for k in sorted(_library_kernels()) + ["__amd_rocclr_copyBuffer", "__amd_rocclr_fillBufferAligned"]:
        hits = [s for s, _, _, prefixes in bench.STAGES if any(k.startswith(p) or (k + "<").startswith(p) for p in prefixes)]
        assert len(hits) == 1, (k, hits)


def test_the_solve_stage_is_priced_with_the_kernels_that_ran():
    """the default chain is srla_lpc_errvars(_lean) + srla_order_select + srla_lpc_taps"""
    class St:                                           # the fields roofline_object reads
        timed_jobs = 4; analyze_launches = 8; num_items = 8 * 15360
        autocorr_ms = 0.8; pitch_ms = 0.0; solve_ms = 0.4; residual_ms = 0.8; price_ms = 0.04; gather_ms = 1.0
    roof = bench.roofline_object(St, "M", 8, 3.6e6, 16.0 * 3.6e6, 100.0)
    solve = roof["stages"]["srla_lpc_solve"]
    assert any(k.startswith("srla_lpc_errvars") for k in solve["kernels"]) and any(k.startswith("srla_lpc_taps") for k in solve["kernels"])
    assert not any(k.startswith("srla_lpc_solve_regs") for k in solve["kernels"])
    assert solve["traffic_over_algorithmic"] > 0.4
    # the longest stage of ALL stages names the line; the single dominant kernel of the committed profile stands beside it
    assert roof["kernel"] == "srla_pack_blocks"
    assert roof["dominant_kernel"]["name"].startswith("srla_")
    assert roof["dominant_kernel"]["stage"] in roof["stages"]


def test_the_config_legs_cut_distinct_windows_out_of_the_runs_streams():
    """bench.py: leg_windows -- C5's nine 300 s files are nine different windows of the run's two 600 s streams, every window inside
    its source; a single-stream leg starts at 0; the legs exist for every configuration but the metric's own and C1"""
    import bench
    n600, n300 = 600 * 48000, 300 * 48000
    w = bench.leg_windows(9, [n600, n600], n300)
    assert len(set(w)) == 9 and all(k in (0, 1) and 0 <= off <= n600 - n300 and off % 2 == 0 for k, off in w)
    assert bench.leg_windows(1, [n600, n600], n300) == [(0, 0)] and bench.leg_windows(1, [n600], n600) == [(0, 0)]
    assert set(bench.LEG_CONFIGS) == set(bench.CONFIGS) - {"M", "C1"}
    assert set(bench.LEG_CPU_SECONDS) == set(bench.LEG_CONFIGS) == set(bench.LEG_NOMINAL)
    for name in bench.LEG_CONFIGS:
        assert bench.CONFIGS[name]["seconds"] <= 600.0 and bench.LEG_CPU_SECONDS[name] <= bench.CONFIGS[name]["seconds"]


def test_the_integer_roof_on_the_line_follows_from_the_committed_rates():
    """roofline.int_valu.peak_tera_lane_ops = profiles/r06/valu_roof.json, which tools/valu_roof.py derives from the committed issue
    rates (profiles/r06/valu_rates.txt, measured on an MI355X by tools/probes/valu_rates.hip) and the ISA histogram of the library
    as built here: rerun the tool and compare (the histogram is the compiler's: a different hipcc may move it by a per cent)."""
    import json
    import subprocess
    import sys
    import helpers
    rates = os.path.join(ROOT, "profiles", "r06", "valu_rates.txt")
    committed = json.load(open(os.path.join(ROOT, "profiles", "r06", "valu_roof.json")))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_roof.py"), rates, helpers.PRODUCT_SO], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    now = json.loads(out.stdout)
    for k in ("srla_residual_cost<2, true>", "srla_autocorr<2, 256, 4096, true>"):
        a, b = now[k], committed[k]
        assert abs(a["peak_tera_lane_ops"] - b["peak_tera_lane_ops"]) <= 0.03 * b["peak_tera_lane_ops"], (k, a["peak_tera_lane_ops"], b["peak_tera_lane_ops"])
        assert 30.0 < a["peak_tera_lane_ops"] < 78.6 and a["assumed_rate_share"] < 0.25
    # the two-operand integer forms issue faster than everything else the kernels use, and nothing reaches the data sheet's 2 cycles
    table = {}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_roof
    table = valu_roof.rates(rates)
    assert 2.0 < table["v_add_u32"] < 3.0 < table["v_add3_u32"] < 5.0 and 4.0 < table["v_add_f64"] < 5.0
