"""The committed evidence agrees with itself (no GPU needed): the bench line's roofline.traffic is what the counter passes
say, and the kernel's average launch time in the rocprofv3 summary is the one bench.py measured with HIP events."""
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _latest_round():
    rounds = sorted({m.group(1) for f in os.listdir(PROF) for m in [re.match(r"(r\d\d)_bench_default\.json$", f)] if m})
    return rounds[-1] if rounds else None


@pytest.mark.skipif(_latest_round() is None, reason="no committed bench line")
def test_bench_line_agrees_with_the_profiles():
    r = _latest_round()
    line = json.load(open(os.path.join(PROF, r + "_bench_default.json")))
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    kernel = roof["kernel"]
    # achieved = algorithmic bytes / average launch duration
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * roof["achieved"]
    # the rocprofv3 --kernel-trace summary of the same command: the kernel's average duration agrees within 3 %
    stats = os.path.join(PROF, r + "_kernel_stats_serialized.txt")
    if os.path.exists(stats):
        avg = None
        for ln in open(stats):
            parts = ln.split()
            if len(parts) > 4 and kernel in parts[0] and parts[1].isdigit():
                avg = float(parts[3]) * 1e-6  # avg_ns -> ms
                break
        assert avg is not None, "kernel %s not in %s" % (kernel, stats)
        assert abs(avg - roof["avg_launch_ms"]) < 0.03 * roof["avg_launch_ms"], (avg, roof["avg_launch_ms"])
    # traffic = the counter passes (FETCH_SIZE doubled + WRITE_SIZE, per launch of the kernel), as tools/pmc_json.py derives it
    pm = json.load(open(os.path.join(PROF, "pmc_latest.json")))
    if roof["traffic"] is not None:
        assert roof["traffic"] == pm["hbm_bytes_per_launch"][kernel]
        p1, p2 = os.path.join(PROF, r + "_pmc_pass1.txt"), os.path.join(PROF, r + "_pmc_pass2.txt")
        if os.path.exists(p1) and os.path.exists(p2):
            import pmc_json
            fetch, ex1 = pmc_json.parse(p1)
            calls1 = dict(pmc_json.own_calls)
            write, ex2 = pmc_json.parse(p2)
            calls2 = dict(pmc_json.own_calls)
            fk = fetch[kernel]["FETCH_SIZE"] / max(calls1.get(kernel, ex1), 1)
            wk = write[kernel]["WRITE_SIZE"] / max(calls2.get(kernel, ex2), 1)
            assert int((fk * 2.0 + wk) * 1024) == roof["traffic"]
    # wasted traffic is visible, not hidden: the line carries the algorithmic bytes next to it
    assert roof["traffic"] is None or roof["traffic"] >= roof["algorithmic_bytes_per_launch"] * 0.9


@pytest.mark.skipif(_latest_round() is None, reason="no committed bench line")
def test_bench_line_has_the_contract_keys():
    line = json.load(open(os.path.join(PROF, _latest_round() + "_bench_default.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert abs(line["value"] - 1024 ** 3 / (line["ms_per_step"] * 1e-3) / 1e6) < 0.01 * line["value"]
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert "workload" in line["config"] and "model" not in line["config"]
