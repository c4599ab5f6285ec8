"""bench.py on the GPU, as the driver runs it: ONE line on stdout, strict JSON, small enough for the driver's 8 KB of stdout
tail, with the contract's keys, `roofline` and (when asked) `cpu_baseline`; everything else in the extras file."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_compact_line(tmp_path):
    extras = tmp_path / "extras.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-box", "--extras-file", str(extras)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]                   # (RCCL's banner and the like go to stderr)
    line = lines[0]
    assert len(line) < 4096
    d = json.loads(line, parse_constant=lambda c: pytest.fail("non-strict JSON constant %s" % c))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "rccl", "path_floor", "value_200_steps"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["unit"] == "proposals/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["rccl"]["world"] == 1
    # value is whole-job throughput of exactly the timed steps
    assert abs(d["value"] - 300.0 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3
    assert 0.5 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["traffic"] is None or rf["traffic"] > rf["algorithmic_bytes"] * 0.9
    assert 0.5 < d["path_floor"]["frac"] <= 1.0
    full = json.load(open(extras))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and "kernel_table" in full
