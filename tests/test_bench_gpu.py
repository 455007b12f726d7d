"""bench.py's output contract on a real GPU: one JSON line with the fields the driver and the judge read, at N = 1
and (rehearsal: gloo control flow, both ranks on this one GPU) at N = 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            out.append(json.loads(ln))
    return out


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--nsamp", "256",
                        "--cpu-nsamp", "64"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "boxes/s" and d["value"] > 0 and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert rf["launches_timed"] >= 1 and rf["launches"] >= rf["launches_timed"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "boxes/s" and cb["sample"]


def test_two_ranks_on_one_gpu_rehearsal():
    env = dict(os.environ, FASTBOX_BENCH_BACKEND="gloo", FASTBOX_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "6", "--warmup", "2", "--nsamp", "128"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1                      # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["parallelism"] == "replicas x2"
