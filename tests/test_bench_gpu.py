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
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line(hip):
    r = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "32", "--cpu-seconds", "1"])
    assert "max_zag" in r["config"]["workload"], "the headline launch passes m_mcu_block_max_zag, as a decode does"
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


def test_eight_ranks_one_device_no_rccl(hip):
    """The shape of the driver's 8-GPU run on a 1-GPU box: eight ranks (sharing the device here), control plane over gloo -- the
    default path creates no RCCL communicator at all (the data path has no collective) -- and EVERY rank checks its own batch."""
    r = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "16"])
    assert KEYS <= set(r) and r["n_gpus"] == 8 and r["cpu_baseline"] is None and r["value"] > 0 and r["scaling"] == "weak"
    assert "all 16 images == oracle" in r["config"]["parity_check"] and "gather" not in r
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count('init_process_group("gloo"') == 1 and 'init_process_group("nccl"' not in src, "RCCL only behind --gather (dist.new_group)"


def test_two_ranks_carry_the_other_configs(hip):
    """N > 1: one command yields the "1 vs 8 GPUs" figures of configs 3 / 4 / 5 too -- every rank runs each workload as a sub-run with a
    rendezvous of its own (small geometry here), rank 0 condenses the lines, a failure on any rank fails the line."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--also-args", "--width 320 --height 200"])
    sys.path.insert(0, ROOT)
    import bench
    assert r["n_gpus"] == 2 and len(r["also"]) == len(bench.ALSO_MULTI)
    for e in r["also"]:
        assert "error" not in e and "skipped" not in e and e["parity"].startswith("ok on every rank"), e
        assert e["n_gpus"] == 2 and e["value"] > 0
    assert r["also"][-1]["scaling"] == "strong" and r["also"][-1]["images_per_gpu_per_step"] == 4096 and r["also"][0]["scaling"] == "weak"


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


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1..4] at their STATED shapes, one step each: bench.py checks parity before it times anything and exits
# non-zero on a mismatch (per-image checksums over the whole batch + the oracle byte for byte on a sample), so a passing run IS
# the parity test at full size -- the launch shapes the small-geometry tests do not reach (512 x 4K takes the device-wide work
# queue of k_png_defilter_queue, 256 layers are converted in resident chunks, 1024 mixed images put three formats side by side).
def _stated(workload, *extra):
    return _run([sys.executable, "bench.py", "--workload", workload, "--steps", "1", "--warmup", "1", "--no-cpu", "--no-traffic"] + list(extra))


def test_config2_with_max_zag_from_files(hip):
    """1024 x 1080p: coefficients + max_zag entropy-decoded from libjpeg-written photographs; files -> pixels == the oracle's
    decompress_jpeg, coefficients and max_zag == the oracle's decode_next_row"""
    r = _stated("jpeg:photo")
    assert r["config"]["images_per_gpu_per_step"] == 1024 and "1920x1080" in r["config"]["workload"]
    assert 0.05 < r["config"]["waves_on_sparse_luma_passes"] < 0.95, "the workload is meant to exercise both the dense and the sparse luma passes"


@pytest.mark.parametrize("policy", ["random", "heuristic"])
def test_config3_512_images_of_4k(hip, policy):
    r = _stated("png:" + policy)
    assert r["config"]["images_per_gpu_per_step"] == 512 and "3840x2160" in r["config"]["workload"] and policy in r["config"]["workload"]
    assert r["roofline"]["algorithmic_bytes_per_launch"] == 512 * (2160 * (3840 * 4 + 1) + 3840 * 2160 * 4)


@pytest.mark.parametrize("pair", ["rgba16:rgbaf32", "rgbaf32:rgba8", "rgba8:rgba16", "rgbaf32:rgba16", "rgba8:rgbaf32", "rgba16:rgba8"])
def test_config4_256_layers_in_chunks(hip, pair):
    r = _stated("convert:" + pair, "--batch", "256")
    assert "256 layers of 8192x8192" in r["config"]["workload"]
    if pair in ("rgba16:rgbaf32", "rgbaf32:rgba16"):
        assert "launches" in r["config"]["workload"], "24 bytes per pixel x 256 layers exceed one GPU's HBM: converted in chunks"


def test_config5_1024_mixed_images(hip):
    """8192 images / 8 GPUs = 1024 per GPU (JPEG / PNG / QOI by index), per-format parity inside"""
    r = _stated("mixed")
    pf = r["config"]["per_format"]
    assert r["config"]["images_per_gpu_per_step"] == 1024 and [pf[k]["images"] for k in ("jpeg", "png", "qoi")] == [342, 341, 341]


def test_default_line_carries_the_other_configs(hip):
    """the driver's own `python bench.py` line: the headline + an `also` entry per other config, each with its parity verdict"""
    r = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-seconds", "2", "--no-traffic", "--also-seconds", "400"])
    sys.path.insert(0, ROOT)
    import bench
    n = len(bench.ALSO)                                                 # the other BASELINE.json configs (+ 4:4:4, -> rgb8, -> l8), inputs resident in HBM
    assert r["config"]["images_per_gpu_per_step"] == 1024 and len(r["also"]) == n + 4, [e.get("workload", e.get("what")) for e in r["also"]]
    for e in r["also"]:
        assert "error" not in e and "skipped" not in e and e.get("parity", "").startswith("ok"), e
    for e in r["also"][:n]:
        assert e["ms_per_step"] > 0 and 0 < e["roofline_frac"] < 1
    for e in r["also"][n:]:                                             # files -> pixels (tools/files_bench.py: baseline and progressive JPEG, two kinds of PNG): PCIe-inclusive, labelled so
        assert e["what"].startswith("files -> pixels") and e["unit"] == "Mpx/s" and e["value"] > 0 and "host memory" in e["inputs"]
