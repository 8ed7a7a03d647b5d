"""The bench.py contract (one JSON line, the fields the driver reads), on a small batch; also the N > 1 path with two ranks
sharing the GPU over gloo (the driver's real multi-GPU runs use RCCL with one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"}


def _run(cmd, env=None):
    e = dict(os.environ); e.update(env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line(hip):
    r = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "32", "--cpu-seconds", "1"])
    assert KEYS <= set(r) and r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["unit"] == "Mpx/s"
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert r["value"] > 0 and rf["achieved"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]


def test_two_ranks_one_line(hip):
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--gather"],
             env={"GAMUT_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["cpu_baseline"] is None and r["value"] > 0
    assert r["gather"].get("own_slice_intact") is True and r["gather"]["ms"] > 0, r["gather"]


def test_self_launch_two_ranks(hip):
    """`python bench.py --gpus 2` with no launcher: bench.py starts its own ranks (BASELINE.json metric: "at 1/2/4/8 GPUs")"""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16"], env={"GAMUT_BENCH_BACKEND": "gloo"})
    assert KEYS <= set(r) and r["n_gpus"] == 2 and r["cpu_baseline"] is None and r["value"] > 0 and r["scaling"] == "weak"
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--workload", "mixed", "--total-images", "14", "--steps", "2", "--warmup", "1",
              "--width", "320", "--height", "200"], env={"GAMUT_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["images_per_gpu_per_step"] == 7


def test_live_traffic_counters(hip):
    """roofline.traffic is measured in the run (rocprofv3 --pmc passes on this very workload), not replayed from a file"""
    r = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "32", "--no-cpu"])
    rf = r["roofline"]
    if rf["traffic"] is None or "replayed" in (rf.get("traffic_source") or ""):
        pytest.skip(f"rocprofv3 counters unavailable here: {rf.get('traffic_source')}")
    assert 0.9 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.2, rf


def test_mixed_workload_line(hip):
    """BASELINE.json configs[4] on one rank: JPEG / PNG / QOI by image index, per-format breakdown, parity checked inside"""
    r = _run([sys.executable, "bench.py", "--workload", "mixed", "--steps", "2", "--warmup", "1", "--batch", "7", "--width", "320", "--height", "200",
              "--cpu-seconds", "1"])
    pf = r["config"]["per_format"]
    assert [pf[k]["images"] for k in ("jpeg", "png", "qoi")] == [3, 2, 2] and all(pf[k]["ms"] > 0 for k in pf)
    assert r["value"] > 0 and r["cpu_baseline"]["value"] > 0
