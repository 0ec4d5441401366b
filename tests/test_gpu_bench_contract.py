"""GPU: bench.py prints ONE JSON line that carries the contract's keys (metric / value / config /
roofline / cpu_baseline, and the secondary bf16x3 measurement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(cuda):
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
         "--train-steps", "0", "--cpu-budget", "2"],
        cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "alt"):
        assert key in d, key
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "configs[1]" in d["config"]["workload"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 0.01
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert isinstance(c["sample"], str) and c["unit"] == "frames/s"
    assert [a["conv_math"] for a in d["alt"]] == ["bf16x8", "bf16x3"]
    for a in d["alt"]:
        assert a["max_err_vs_f32_rel_to_max"] < a["tolerance"] == 1e-4 and a["value"] > 0
    assert d["alt"][0]["max_err_vs_f32_rel_to_max"] < 2e-6  # three exact planes: fp32-grade
    assert d["value"] > 100 * c["value"]  # the north star's >= 100x the same-host CPU path
