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
                        "--cpu-nsamp", "64", "--sizes", "128,512"], capture_output=True, text=True, timeout=900, cwd=ROOT)
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
    assert rf["timed_in"] and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "boxes/s" and cb["sample"]
    # the auxiliary legs: reference precision, BASELINE configs[2], other grid sizes, both byte counts of the step
    assert d["f64"]["value"] > 0 and d["f64"]["dtype"] == "f64"
    c3 = d["config3"]
    assert c3["ms_per_chain"] > 0 and c3["model_sweeps"] == 13.5 and 0 < c3["rsd_roofline"]["frac"] < 1
    assert set(d["sizes"]) == {"128", "512"} and all(v["value"] > 0 for v in d["sizes"].values())
    pr = d["pipeline_roofline"]
    assert pr["model_sweeps"] == 5.0 and pr["moved_sweeps"] == 4.5 and 0 < pr["frac_moved"] < pr["frac"] < 1
    assert "from_profiles" in d and "file" in d["from_profiles"] and "head" in d["from_profiles"]
    # the headline is the median of >= 5 fenced regions of `steps` steps, every region listed
    rg = d["regions"]
    assert rg["count"] >= 5 and len(rg["ms_per_step"]) == rg["count"] and all(x > 0 for x in rg["ms_per_step"])
    assert sorted(rg["ms_per_step"])[rg["count"] // 2] == pytest.approx(d["ms_per_step"], rel=1e-3) and rg["spread"] >= 0
    assert d["collective"] is None                 # one rank: no process group
    assert "best of" in c3["workload"]


def test_two_ranks_started_by_bench_itself():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks (rehearsal: gloo control
    flow, both ranks on this one GPU), rank 0 prints the line, and the strong-scaling leg runs the slab-decomposed box
    as a child job of two ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(FASTBOX_BENCH_BACKEND="gloo", FASTBOX_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--nsamp", "128", "--slab-sizes", "64"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [d for d in _json_lines(r.stdout) if d.get("scaling") == "weak"]
    assert len(lines) == 1                      # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["parallelism"] == "replicas x2"
    import math

    def group_ok(c):        # what the process group really was: backend, size as the group reports it, one entry per rank
        assert c["backend"] == "gloo" and c["world_size"] == 2 and len(c["devices"]) == 2
        assert sorted(g["rank"] for g in c["devices"]) == [0, 1] and all("device" in g and "pci_bus_id" in g for g in c["devices"])
    group_ok(d["collective"])
    assert d["regions"]["count"] >= 5
    ss = d["strong_scaling"]["64"]
    assert ss["n_gpus"] == 2 and ss["scaling"] == "strong" and math.isfinite(ss["value"]) and ss["value"] > 0
    assert math.isfinite(ss["ms_per_step"]) and ss["finite"] is True
    ex = ss["exchange"]
    assert ex["ms_per_step_pipelined"] > 0 and ss["config"]["parallelism"] == "slab x2"
    assert math.isfinite(ex["ms_per_step_one_realisation_at_a_time"]) and math.isfinite(ex["exposed_exchange_ms_per_step"])
    assert 2 <= ex["chunks_per_transform"] <= 4 and math.isfinite(ex["ms_per_step_one_realisation_at_a_time_unchunked"])
    group_ok(ss["collective"])


def test_two_ranks_under_an_external_launcher():
    """The driver's form: torch.distributed.run around bench.py (WORLD_SIZE set): no second launch."""
    env = dict(os.environ, FASTBOX_BENCH_BACKEND="gloo", FASTBOX_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "6", "--warmup", "2", "--nsamp", "128", "--no-extras"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0
    assert lines[0]["collective"]["world_size"] == 2 and lines[0]["collective"]["backend"] == "gloo"
