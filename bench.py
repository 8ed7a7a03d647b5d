#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Gamut hot path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): a batch of 1024 synthetic 1920x1080 baseline 4:2:0
JPEGs per GPU, already entropy-decoded to dense de-quantised coefficients resident in HBM;
one "step" = one pass of the hot path (8x8 IDCT + frequency-domain 4:2:0 chroma upsample +
YCbCr->RGBA8) over the whole batch through the C ABI (gamut_hip_jpeg_reconstruct_batch_device).
Images are independent, so ranks shard by image index with no data-path collective
("scaling": "weak": every rank owns a full 1024-image batch).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the launch
stream; `cpu_baseline` is the CPU oracle (a scalar C restatement of the reference's loops,
oracle/) timed on this box's host cores over a bounded sample of the same coefficients.
Other kernels of the path: --workload convert:<src>:<dst> | png | mixed (configs[4]: JPEG / PNG / QOI)  (same JSON shape).
Measurement knobs (environment): GAMUT_BENCH_CONVERT_GB (resident chunk of the convert workloads, default 128), GAMUT_BENCH_STEP_MS (every
timed step on stderr), GAMUT_BENCH_BACKEND=gloo (the optional gather without RCCL), GAMUT_BENCH_NOCHECK (experiments with deliberately
wrong kernels only: the line says so in config.parity_check).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

# dmabuf IPC: RCCL and device memory shared across processes need it on this pool -- set before the first HIP call on EVERY path (under
# the driver's own `torch.distributed.run ... bench.py --gpus N` nothing else would)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU per step")
    ap.add_argument("--width", type=int, default=0, help="default: the BASELINE.json geometry of the workload (1920 x 1080 JPEG / mixed, 3840 x 2160 PNG, 8192 x 8192 convert)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--workload", default="jpeg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic falls back to profiles/)")
    ap.add_argument("--serial-formats", action="store_true", help="mixed workload: one stream, one format after the other")
    ap.add_argument("--total-images", type=int, default=0, help="mixed workload: images over ALL ranks (8192 = BASELINE.json configs[4]); "
                    "each rank takes total / N (strong scaling) instead of --batch")
    ap.add_argument("--no-also", action="store_true", help="default workload only: skip the `also` lines (the other BASELINE.json configs at their stated shapes)")
    ap.add_argument("--also-seconds", type=float, default=420.0, help="wall-clock budget of the `also` lines; what does not fit is reported as skipped")
    ap.add_argument("--gather", action="store_true", help="N > 1: also time an all_gather of output slices over RCCL (after the timed region; the only use of RCCL in this script)")
    ap.add_argument("--also-args", default="", help="N > 1 test aid: extra arguments for every `also` sub-run (e.g. a small geometry); implies the `also` lines at any --batch")
    return ap.parse_args()


def host_cores():
    """cores this process may use: the affinity mask, capped by a cgroup CPU quota (the GPU box: 256 threads visible, quota 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def traffic_from_profiles(workload, kernel_substr):
    """(HBM bytes per image, file) from the committed rocprofv3 --pmc summary of THIS workload (profiles/<tag>_bench.json names the
    workload, <tag>_traffic.json holds the counters), or (None, None) when the workload has not been profiled."""
    import glob
    import re
    norm = lambda w: re.sub(r"^batch \d+ x |, \d+ layers of ", "|", w)
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench.json")), reverse=True):      # newest round first
        try:
            line = [ln for ln in open(p).read().splitlines() if ln.startswith("{")][-1]
            if norm(json.loads(line)["config"]["workload"]) != norm(workload):
                continue
            tf = p[:-len("_bench.json")] + "_traffic.json"
            rows = json.load(open(tf)).get("kernels", [])
            rows = [r for r in rows if kernel_substr.split(" + ")[0] in r.get("kernel", "") and r.get("hbm_bytes_per_image")]
            if rows:
                return max(r["hbm_bytes_per_image"] for r in rows), os.path.join("profiles", os.path.basename(tf))
        except Exception:
            pass
    return None, None


def live_traffic(kernel_substr, pmc_batch):
    """HBM bytes per image of the dominant kernel, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE: they
    do not fit one pass) over a short run of this same script and workload at a smaller batch.  Corrections as
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: counters are in KB, and on gfx950 FETCH_SIZE counts a wide
    coalesced read stream at half its bytes.  Returns (bytes_per_image, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    inner = [a for a in sys.argv[1:]]
    for flag in ("--steps", "--warmup", "--batch", "--gpus", "--cpu-seconds"):       # replaced below
        while flag in inner:
            i = inner.index(flag); del inner[i:i + 2]
    inner = [a for a in inner if a not in ("--no-cpu", "--gather")]
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gamut_pmc_", dir="/tmp")
        try:
            # (counters on this library's kernels only: with the torch kernels that synthesise 6.4 GB of input instrumented too, a pass at
            # the timed batch took longer than its time limit)
            cmd = [exe, "--output-format", "csv", "--kernel-include-regex", "gamut", "--pmc", ctr, "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__)] + inner + \
                  ["--steps", "3", "--warmup", "1", "--no-cpu", "--no-traffic", "--no-also", "--batch", str(pmc_batch)]
            env = dict(os.environ, TMPDIR="/tmp", GAMUT_BENCH_NOCHECK="1")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == ctr and kernel_substr.split(" + ")[0] in row.get("Kernel_Name", ""):
                        tot += float(row["Counter_Value"]); n += 1
            if not n:
                return None, f"no {ctr} rows (rocprofv3 rc {r.returncode})"
            vals[ctr] = tot / n
        except Exception as e:                                     # never fail the bench line over the counters
            return None, repr(e)[:120]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    per_launch = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    return per_launch / pmc_batch, f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in this run, batch {pmc_batch}, FETCH_SIZE x2 (gfx950), KB -> B"


ALSO = [  # the other BASELINE.json configs at their stated shapes, each through this same script (its own parity check included)
    ("config 2, what a caller has: coefficients + max_zag from libjpeg-written photographs", "jpeg:photo", []),
    ("config 2's kernel family on 4:4:4 files (k_jpeg_cols)", "jpeg:4:1", []),
    ("the headline batch -> rgb8, what loadJPEG produces when the caller sets no flags (plugins/jpeg.d:48-86)", "jpeg:3", []),
    ("the headline batch -> l8 (LOAD_GREYSCALE)", "jpeg:1", []),
    ("config 3: 512 x 3840x2160 RGBA8, random row filters (a fifth Paeth)", "png", []),
    ("config 3: the same with the encoder heuristic's filters (no Paeth rows on this data)", "png:heuristic", []),
    ("config 4: rgba16 -> rgbaf32, 256 layers of 8192x8192 in resident chunks", "convert:rgba16:rgbaf32", ["--batch", "256"]),
    ("config 4: rgbaf32 -> rgba8, 256 layers", "convert:rgbaf32:rgba8", ["--batch", "256"]),
    ("config 4: rgba8 -> rgba16, 256 layers", "convert:rgba8:rgba16", ["--batch", "256"]),
    ("config 4, the reverse directions: rgbaf32 -> rgba16 (scanline.d:731-746), 256 layers", "convert:rgbaf32:rgba16", ["--batch", "256"]),
    ("config 4: rgba8 -> rgbaf32 (scanline.d:428-443), 256 layers", "convert:rgba8:rgbaf32", ["--batch", "256"]),
    ("config 4: rgba16 -> rgba8 (through the rgbaf32 intermediate, scanline.d:25-31 / image.d:1238-1241), 256 layers", "convert:rgba16:rgba8", ["--batch", "256"]),
    ("config 5 at its stated size on this GPU: 8192 mixed 1080p images", "mixed", ["--total-images", "8192"]),
    ("config 5, one GPU's share of the 8-GPU run: 1024 mixed images (342 JPEG + 341 PNG + 341 QOI: the four-wave QOI kernel)", "mixed", []),
]


def also_lines(budget_s):
    """Runs every ALSO workload as `bench.py --workload ... --steps 20 --warmup 5` in a process of its own (fresh HBM) and condenses
    its JSON line.  A non-zero exit code is a parity failure or an error and is reported as such, never dropped."""
    import subprocess
    t_end = time.perf_counter() + budget_s
    res = []
    for what, wl, extra in ALSO:
        left = t_end - time.perf_counter()
        if left < 20:
            res.append({"what": what, "workload": wl, "skipped": "the --also-seconds budget ran out"})
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", "20", "--warmup", "5", "--no-cpu", "--no-also"] + extra
        # every line takes its HBM traffic live, at its own timed batch (two rocprofv3 --pmc passes restricted to this library's kernels),
        # as long as the budget has room for the lines behind it; a line that replays a committed summary names the file.  The mixed
        # step is three kernels side by side: no single-kernel traffic figure
        if wl == "mixed" or left < 150:
            cmd.append("--no-traffic")
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                res.append({"what": what, "workload": wl, "parity": "FAILED" if "PARITY" in (r.stderr + r.stdout) else None,
                            "error": ((r.stderr or r.stdout).strip().splitlines() or ["no output"])[-1][:200], "rc": r.returncode})
                continue
            j = json.loads(line[-1])
            e = {"what": what, "workload": j["config"]["workload"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                 "roofline_frac": j["roofline"]["frac"], "achieved_GB/s": j["roofline"]["achieved"], "kernel": j["roofline"]["kernel"],
                 "kernel_ms_avg": j["roofline"]["kernel_ms_avg"], "parity": "ok: " + str(j["config"].get("parity_check", "checked against the oracle before timing")), "steps": j["steps"],
                 "wall_s": round(time.perf_counter() - t0, 1)}
            if j["roofline"].get("traffic"):
                e["traffic"] = j["roofline"]["traffic"]; e["traffic_over_algorithmic"] = round(j["roofline"]["traffic"] / j["roofline"]["algorithmic_bytes_per_launch"], 4)
                e["traffic_source"] = j["roofline"].get("traffic_source")
            for k in ("per_format", "waves_on_sparse_luma_passes", "y_blocks_max_zag_le_10"):
                if k in j["config"]:
                    e[k] = j["config"][k]
            res.append(e)
        except subprocess.TimeoutExpired:
            res.append({"what": what, "workload": wl, "skipped": "did not finish inside the --also-seconds budget"})
        except Exception as ex:                                          # the headline line must survive anything here
            res.append({"what": what, "workload": wl, "error": repr(ex)[:200]})
    # what a caller with FILES sees (the feeders in front of the hot path) -- tools/files_bench.py, one line per case
    left = t_end - time.perf_counter()
    if left < 45:
        res.append({"what": "files -> pixels (tools/files_bench.py)", "skipped": "the --also-seconds budget ran out"})
    else:
        try:
            r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "files_bench.py")],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left, text=True)
            lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
            res.extend(lines)
            if r.returncode != 0 or not lines:
                res.append({"what": "files -> pixels (tools/files_bench.py)", "error": ((r.stderr or r.stdout).strip().splitlines() or ["no output"])[-1][:200], "rc": r.returncode})
        except subprocess.TimeoutExpired:
            res.append({"what": "files -> pixels (tools/files_bench.py)", "skipped": "did not finish inside the --also-seconds budget"})
        except Exception as ex:
            res.append({"what": "files -> pixels (tools/files_bench.py)", "error": repr(ex)[:200]})
    return res


ALSO_MULTI = [  # N > 1: what BASELINE.json quotes "at 1 / 8 GPUs" beside the headline -- config 3 and 4 weak (every rank a full batch), config 5 strong (8192 images over all ranks)
    ("config 3: 512 x 3840x2160 RGBA8 per GPU, random row filters", "png", []),
    ("config 3: the same with the encoder heuristic's filters", "png:heuristic", []),
    ("config 4: rgba16 -> rgbaf32, 256 layers of 8192x8192 per GPU", "convert:rgba16:rgbaf32", ["--batch", "256"]),
    ("config 4: rgbaf32 -> rgba8, 256 layers per GPU", "convert:rgbaf32:rgba8", ["--batch", "256"]),
    ("config 4: rgba8 -> rgba16, 256 layers per GPU", "convert:rgba8:rgba16", ["--batch", "256"]),
    ("config 5: 8192 mixed 1080p images over all ranks (strong scaling)", "mixed", ["--total-images", "8192"]),
]


def also_lines_multi(args, rank, world, dist):
    """Every rank runs each ALSO_MULTI workload as a sub-run of this script with the launcher's RANK / LOCAL_RANK / WORLD_SIZE and a
    rendezvous port of its own (rank 0 picks it, broadcast over the gloo group); rank 0 condenses its child's JSON line."""
    import shlex
    import socket
    import subprocess
    t_end = time.perf_counter() + args.also_seconds
    res = []
    for k, (what, wl, extra) in enumerate(ALSO_MULTI):
        left = [t_end - time.perf_counter()]
        port = [0]
        if rank == 0:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port[0] = sock.getsockname()[1]
        ctl = [port[0], left[0]]
        dist.broadcast_object_list(ctl, src=0)                    # (rank 0's clock decides: every rank takes the same branch)
        if ctl[1] < 30:
            res.append({"what": what, "workload": wl, "skipped": "the --also-seconds budget ran out"})
            continue
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(ctl[0]))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)             # the sub-run's rank 0 hosts its own store (the launcher's agent store is on the parent's port)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--workload", wl, "--steps", "20", "--warmup", "5", "--no-cpu", "--no-also",
               "--no-traffic"] + extra + shlex.split(args.also_args)
        t0 = time.perf_counter()
        e = {"what": what, "workload": wl}
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=ctl[1] + 60, text=True, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or (rank == 0 and not line):
                e.update({"parity": "FAILED" if "PARITY" in (r.stderr + r.stdout) else None, "error": ((r.stderr or r.stdout).strip().splitlines() or ["no output"])[-1][:200], "rc": r.returncode})
            elif rank == 0:
                j = json.loads(line[-1])
                e = {"what": what, "workload": j["config"]["workload"], "n_gpus": j["n_gpus"], "scaling": j["scaling"], "value": j["value"], "unit": j["unit"],
                     "ms_per_step": j["ms_per_step"], "roofline_frac_rank0": j["roofline"]["frac"], "kernel": j["roofline"]["kernel"], "kernel_ms_avg_rank0": j["roofline"]["kernel_ms_avg"],
                     "images_per_gpu_per_step": j["config"]["images_per_gpu_per_step"],
                     "parity": "ok on every rank: " + str(j["config"].get("parity_check", "")), "steps": j["steps"], "wall_s": round(time.perf_counter() - t0, 1)}
                if "per_format" in j["config"]:
                    e["per_format"] = j["config"]["per_format"]
        except subprocess.TimeoutExpired:
            e["skipped"] = "did not finish inside the --also-seconds budget"
        except Exception as ex:
            e["error"] = repr(ex)[:200]
        # a failure on ANY rank is the line's failure
        bad = [None] * world
        dist.all_gather_object(bad, e.get("error") or e.get("skipped"))
        worst = next((f"rank {r_}: {b}" for r_, b in enumerate(bad) if b), None)
        if worst and "error" not in e and "skipped" not in e:
            e = {"what": what, "workload": wl, "error": worst}
        res.append(e)
    return res


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this script under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and let rank 0's JSON line through.  The driver's own
    `python -m torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE, so this is skipped there."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)                                         # (HSA_ENABLE_IPC_MODE_LEGACY=0 is set at import, on every path)
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus) and int(os.environ.get("RANK", "0")) == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # One process per GPU; a box with fewer GPUs than ranks (the 1-GPU test boxes) lets ranks share devices.  The data path has NO
    # collective (images are independent, sharded by index), so the control plane -- rendezvous, the barriers around the
    # timed region, the max-reduce of the elapsed time -- runs over gloo on the host: an RCCL communicator problem cannot cost the
    # scaling curve.  RCCL is created only for --gather (the one exchange the path has), and only when every rank has a GPU of its own.
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")        # one node: never resolve the container's hostname
        sys.stdout.flush()
        saved = os.dup(1)                                            # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."):
        os.dup2(2, 1)                                                # stdout carries ONE JSON line, so they go to stderr
        try:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
            dist.barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    gather_backend = "nccl" if (ndev >= int(os.environ.get("LOCAL_WORLD_SIZE", world)) and os.environ.get("GAMUT_BENCH_BACKEND", "nccl") == "nccl") else "gloo"

    from gamut_amd import _capi, synth
    L = _capi.lib()
    _capi.check(L.gamut_hip_init(dev_index))
    stream = torch.cuda.current_stream().cuda_stream

    # ------------------------------------------------------------------ workload
    w, h, B = args.width or 1920, args.height or 1080, args.batch
    wl = args.workload
    check = None
    check_note = {}                                              # what check() compared, for the JSON line
    _host = {}                                                   # host copies of the cpu_baseline sample (made once, shared by the threads)
    if wl == "jpeg" or wl.startswith("jpeg:"):
        jp = wl.split(":")                                       # jpeg[:out_comps[:scan_type]]  |  jpeg:photo[:out_comps]
        photo = len(jp) > 1 and jp[1] == "photo"
        if photo:
            jp = [jp[0]] + jp[2:]
        oc = int(jp[1]) if len(jp) > 1 else 4                    # 4 = rgba8 (headline), 3 = rgb8, 1 = l8
        st = int(jp[2]) if len(jp) > 2 else 4                    # jpgd scan type: 4 = 4:2:0 (headline), 3 = 4:4:0, 2 = 4:2:2, 1 = 4:4:4, 0 = grey
        comps_in = 1 if st == 0 else 3
        files = None
        if photo:
            # What a caller has after decode_next_row: coefficients AND m_mcu_block_max_zag of real files.  Synthetic photographs
            # (synth.photo_rgb: 1/f detail on ~45 % of the frame, smooth elsewhere) written by libjpeg (Pillow) as baseline 4:2:0 at
            # q 75 / 85 / 90, entropy-decoded on the GPU (gamut_hip_jpeg_entropy_decode_device) straight into the resident buffers
            # the timed launch reads.
            import io
            from PIL import Image
            assert st == 4
            nd = max(1, min(B, 6))
            files = []
            for i in range(nd):
                bio = io.BytesIO()
                Image.fromarray(synth.photo_rgb(w, h, 1000 * (1 + rank) + i)).save(bio, "JPEG", quality=(75, 85, 90)[i % 3], subsampling=2)
                files.append(np.frombuffer(bio.getvalue(), np.uint8))
            nblk = ((w + 15) // 16) * ((h + 15) // 16) * 6
            coeffs = torch.empty((B, nblk, 64), dtype=torch.int16, device=dev)
            zag = torch.empty((B, nblk), dtype=torch.uint8, device=dev)
            bufs = [files[i % nd] for i in range(B)]
            ptrs = (C.c_void_p * B)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * B)(*[b.size for b in bufs])
            co_off = np.arange(B, dtype=np.int64) * nblk * 64; zz_off = np.arange(B, dtype=np.int64) * nblk
            info = (_capi.JpegFrame * B)()
            _capi.check(L.gamut_hip_jpeg_entropy_decode_device(ptrs, lens, B, co_off.ctypes.data_as(C.POINTER(C.c_int64)), zz_off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                                coeffs.data_ptr(), zag.data_ptr(), None, info, None, stream))
            torch.cuda.synchronize()
        else:
            coeffs = synth.jpeg_coeff_batch(B, w, h, dev, seed=1 + rank, scan_type=st)
            zag = synth.jpeg_max_zag(coeffs)                     # m_mcu_block_max_zag, as decode_next_row leaves it (jpegload.d:2512)
            nblk = coeffs.shape[1]
        out = torch.empty((B, h, w * oc), dtype=torch.uint8, device=dev)
        px_per_step = B * w * h
        bytes_per_step = B * (nblk * 128 + w * h * oc)           # 6 266 880 + 8 294 400 per 1080p image (rgba8); the max_zag bytes (48 960) are not counted
        kernel_name = "k_jpeg_h2v2" if st == 4 else "k_jpeg_"
        zstat = {}
        if st == 4:                                              # which share of the kernel's waves (2 MCUs = 8 Y blocks) takes the Row!4 / Col!4 passes
            yz = zag.view(B, -1, 6)[:, :, :4]
            mrow = (w + 15) // 16
            pair = (yz <= 10).all(dim=2).view(B, -1, mrow)[:, :, :mrow // 2 * 2].reshape(B, -1, 2).all(dim=2)
            zstat = {"y_blocks_max_zag_le_10": round(float((yz <= 10).float().mean()), 4), "y_max_zag_median": float(yz.float().median()),
                     "waves_on_sparse_luma_passes": round(float(pair.float().mean()), 4)}
        workload = (f"batch {B} x {w}x{h} baseline JPEG { {4: '4:2:0', 3: '4:4:0', 2: '4:2:2', 1: '4:4:4', 0: 'grey'}[st] }, IDCT"
                    f"{' + freq-domain chroma upsample' if st == 4 else ''} + YCbCr->{ {4: 'RGBA8', 3: 'RGB8', 1: 'L8'}[oc] }"
                    + (", coefficients + max_zag entropy-decoded from libjpeg-written synthetic photographs (q 75/85/90)" if photo else ", coefficients + max_zag of the q 90 smooth-plus-noise generator"))

        def step():
            _capi.check(L.gamut_hip_jpeg_reconstruct_batch_device(coeffs.data_ptr(), nblk * 64, zag.data_ptr(), nblk, out.data_ptr(), w * oc,
                                                                   h * w * oc, w, h, st, oc, B, stream))

        def check():
            """EVERY image of the batch against the oracle (the oracle is C behind ctypes, which releases the GIL: one image per host
            thread at a time, ~1.5-3 s for 1024 frames on the GPU box's 16-core quota)."""
            import oracle_lib as O
            from concurrent.futures import ThreadPoolExecutor
            step()
            torch.cuda.synchronize()

            def one(i):
                co, zz = coeffs[i].cpu().numpy(), zag[i].cpu().numpy()
                got = out[i].cpu().numpy()
                if not np.array_equal(got, O.jpeg_reconstruct(w, h, comps_in, st, co, zz, oc)):
                    return f"PARITY FAILURE on image {i}"
                if photo and i < 2 * len(files):                 # the whole file through the oracle's own decoder: entropy decode included
                    d = O.DecodedJpeg(bytes(files[i % len(files)]))
                    if not (np.array_equal(d.coeffs, co) and np.array_equal(d.max_zag, zz)):
                        return f"PARITY FAILURE (coefficients / max_zag of file {i})"
                    if oc == 4 and not np.array_equal(got, O.decompress_jpeg(bytes(files[i % len(files)]), 4)[0]):
                        return f"PARITY FAILURE (file {i} vs decompress_jpeg)"
                return None
            t0 = time.perf_counter()
            with ThreadPoolExecutor(host_cores()) as pool:
                bad = [r for r in pool.map(one, range(B)) if r]
            if bad:
                raise SystemExit(bad[0] + (f" (+ {len(bad) - 1} more)" if len(bad) > 1 else ""))
            check_note["checked"] = f"all {B} images == oracle, {time.perf_counter() - t0:.1f} s on {host_cores()} host threads"

        def cpu_leg(seconds):
            import oracle_lib as O
            if "h" not in _host:
                _host["h"] = [(coeffs[i].cpu().numpy(), zag[i].cpu().numpy()) for i in range(min(B, 64))]
            host = _host["h"]
            O.jpeg_reconstruct(w, h, comps_in, st, host[0][0], host[0][1], oc)             # warm
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                O.jpeg_reconstruct(w, h, comps_in, st, host[n % len(host)][0], host[n % len(host)][1], oc)
                n += 1
            dt = time.perf_counter() - t0
            return n * w * h / dt / 1e6, f"{n} of the batch's {w}x{h} coefficient frames (with max_zag: the sparse Row!N / Col!N paths), coefficients -> pixels, single thread, {dt:.1f} s"
        dtype = "int32"
    elif wl.startswith("convert:"):
        import oracle_lib as O
        _, s, d = wl.split(":")
        st, dt_ = O.PT[s], O.PT[d]
        B = args.batch if args.batch != 1024 else 8
        w = h = args.width or 8192
        npx = w * h
        g = torch.Generator(device=dev); g.manual_seed(7 + rank)
        sdt = O.PT_DTYPE[st]
        # BASELINE.json configs[3] is a batch of 256 layers: 412 GB for rgba16 -> rgbaf32, more than one GPU holds.  The layers
        # are converted in chunks: R resident layers (<= 128 GB of source + destination), a step = ceil(B / R) launches over them.
        R = max(1, min(B, int(float(os.environ.get("GAMUT_BENCH_CONVERT_GB", "128")) * 1e9 // (npx * (O.PT_SIZE[st] + O.PT_SIZE[dt_])))))
        if sdt == np.float32:
            src = torch.rand((R, npx * O.PT_CHANNELS[st]), device=dev, generator=g, dtype=torch.float32)
        elif sdt == np.uint16:
            src = torch.randint(0, 65536, (R, npx * O.PT_CHANNELS[st]), device=dev, generator=g, dtype=torch.int32).to(torch.uint16)
        else:
            src = torch.randint(0, 256, (R, npx * O.PT_CHANNELS[st]), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
        out = torch.empty((R, npx * O.PT_SIZE[dt_]), dtype=torch.uint8, device=dev)
        px_per_step = B * npx
        bytes_per_step = px_per_step * (O.PT_SIZE[st] + O.PT_SIZE[dt_])
        kernel_name = f"k_convert_vec<{st}, {dt_}>"
        launches = (B + R - 1) // R
        workload = f"convertTo {s}->{d}, {B} layers of {w}x{h}, gapless" + (f", {launches} launches of <= {R} resident layers" if launches > 1 else "")
        sp, dp = w * O.PT_SIZE[st], w * O.PT_SIZE[dt_]

        def step():
            for c in range(0, B, R):
                _capi.check(L.gamut_hip_scanlines_convert_device(st, src.data_ptr(), sp, sp * h, dt_, out.data_ptr(), dp, dp * h,
                                                                  w, h, min(R, B - c), stream))

        def check():
            """whole layers against the oracle: the first, the middle and the last resident layer and the last one the step's final launch
            converted, every row of each, in bands of 128 rows on the host threads"""
            from concurrent.futures import ThreadPoolExecutor
            step()
            torch.cuda.synchronize()
            rows = 128
            last = (B - 1) % R if B % R else R - 1                     # a layer the step's last launch converted
            layers = sorted({0, R // 2, last, R - 1})                  # (R - 1: the library splits more layers than 32 bits index into several launches -- the last of them)

            def one(job):
                layer, r0 = job
                n = min(rows, h - r0)
                a = src[layer].view(torch.uint8)[r0 * sp:(r0 + n) * sp].cpu().numpy()
                exp = O.scanlines_convert(st, a, dt_, w, n)
                return None if np.array_equal(out[layer][r0 * dp:(r0 + n) * dp].cpu().numpy(), exp) else f"PARITY FAILURE (layer {layer}, rows {r0}..)"
            t0 = time.perf_counter()
            with ThreadPoolExecutor(host_cores()) as pool:
                bad = [r for r in pool.map(one, [(l, r0) for l in layers for r0 in range(0, h, rows)]) if r]
            if bad:
                raise SystemExit(bad[0])
            check_note["checked"] = f"layers {layers} of the {R} resident ones, every row == oracle, {time.perf_counter() - t0:.1f} s"

        def cpu_leg(seconds):
            rows = 256
            if "h" not in _host:
                _host["h"] = src[0].view(torch.uint8)[:rows * sp].cpu().numpy()
            a = _host["h"]
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                O.scanlines_convert(st, a, dt_, w, rows)
                n += 1
            dt = time.perf_counter() - t0
            return n * rows * w / dt / 1e6, f"{n} x {rows} rows of {w} px, scanlinesConvert, single thread, {dt:.1f} s"
        dtype = "f32"
    elif wl == "png" or wl.startswith("png:"):
        import oracle_lib as O
        parts = wl.split(":")                                  # png[:policy[:channels]]
        # default: a random filter per row (a fifth of the rows Paeth, every band mixed: what adaptive encoders produce on photographs).
        # "heuristic" = the minimum-sum-of-absolute-differences choice on THIS synthetic data, which never picks Paeth: the cheap case.
        policy = parts[1] if len(parts) > 1 and parts[1] else "random"
        ch = int(parts[2]) if len(parts) > 2 else 4
        on = int(parts[3]) if len(parts) > 3 else ch           # out_n: ch or ch + 1 (alpha inserted, stbdec.d:1467-1480)
        if policy.isdigit():
            policy = int(policy)
        if not (args.width and args.height):
            w, h = 3840, 2160
        B = args.batch if args.batch != 1024 else 512
        raw, sums = synth.png_raw_batch(B, w, h, dev, seed=3 + rank, policy=policy, channels=ch)
        raw_len = raw.shape[1]
        out = torch.empty((B, h * w * on), dtype=torch.uint8, device=dev)
        status = torch.zeros((B,), dtype=torch.int32, device=dev)
        if os.environ.get("GAMUT_BENCH_ADDR"):
            print(f"[bench] raw {raw.data_ptr():#x} out {out.data_ptr():#x}", file=sys.stderr)
        px_per_step = B * w * h
        bytes_per_step = B * (raw_len + w * h * on)            # 33 179 760 + 33 177 600 per 3840x2160 RGBA8 image
        kernel_name = "k_png_defilter"
        workload = f"batch {B} x {w}x{h} PNG 8-bit {['', 'grey', 'grey+alpha', 'RGB', 'RGBA'][ch]}, post-inflate de-filter{' + alpha insert' if on != ch else ''} ({policy} row filters)"

        def step():
            _capi.check(L.gamut_hip_png_defilter_batch_device(raw.data_ptr(), raw_len, raw_len, out.data_ptr(), w * h * on, w, h, ch, on, 8, {1: 0, 2: 4, 3: 2, 4: 6}[ch],
                                                               B, status.data_ptr(), stream))

        def check():
            step()
            torch.cuda.synchronize()
            assert int(status.abs().sum()) == 0
            got = out.view(B, -1).to(torch.int64).sum(dim=1) if B <= 64 else torch.stack([out[i].to(torch.int64).sum() for i in range(B)])
            if not torch.equal(got, sums + (on - ch) * 255 * w * h):           # inserted alpha = 255
                raise SystemExit("PARITY FAILURE: checksum of de-filtered pixels != checksum of the source pixels")
            from concurrent.futures import ThreadPoolExecutor

            def one(i):                                         # every image byte for byte against the oracle, on the host threads
                exp = O.png_create_image_raw(raw[i].cpu().numpy(), ch, on, w, h, 8, {1: 0, 2: 4, 3: 2, 4: 6}[ch])
                return None if np.array_equal(out[i].cpu().numpy(), exp) else f"PARITY FAILURE vs oracle on image {i}"
            t0 = time.perf_counter()
            with ThreadPoolExecutor(host_cores()) as pool:
                bad = [r for r in pool.map(one, range(B)) if r]
            if bad:
                raise SystemExit(bad[0])
            check_note["checked"] = f"all {B} images == oracle byte for byte (+ checksum of the source pixels), {time.perf_counter() - t0:.1f} s"

        def cpu_leg(seconds):
            if "h" not in _host:
                _host["h"] = [raw[i].cpu().numpy() for i in range(min(B, 8))]
            host = _host["h"]
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                O.png_create_image_raw(host[n % len(host)], ch, on, w, h, 8, {1: 0, 2: 4, 3: 2, 4: 6}[ch])
                n += 1
            dt = time.perf_counter() - t0
            return n * w * h / dt / 1e6, f"{n} of the batch's {w}x{h} filtered streams, stbi__create_png_image_raw, single thread, {dt:.1f} s"
        dtype = "u8"
    elif wl == "mixed":
        # BASELINE.json configs[4] on one rank: image i of the batch is a JPEG (i % 3 == 0), a PNG (1) or a QOI file (2), all
        # 1920x1080 -> rgba8.  Inputs resident in HBM in the form each GPU stage starts from: dense coefficients, inflated
        # filtered streams, QOI files as they are.  One launch per format; the per-format times are in config.per_format.
        import oracle_lib as O
        if args.batch == 1024 and args.total_images:               # BASELINE.json configs[4]: 8192 images over all ranks
            B = (args.total_images + world - 1) // world
        nj, npn, nq = (B + 2) // 3, (B + 1) // 3, B // 3
        coeffs = synth.jpeg_coeff_batch(nj, w, h, dev, seed=1 + rank)
        nblk = coeffs.shape[1]
        raw, sums = synth.png_raw_batch(npn, w, h, dev, seed=3 + rank, policy="heuristic", channels=4)
        raw_len = raw.shape[1]
        nd = max(1, min(8, nq))
        rgb = synth.synth_rgb_batch(nd, w, h, dev, seed=5 + rank).permute(0, 2, 3, 1).to(torch.uint8).cpu().numpy()
        rgb[:, h // 4:h // 2, w // 4:w // 2] = rgb[:, h // 4:h // 4 + 1, w // 4:w // 4 + 1]     # a flat patch: RUN ops
        files = [synth.qoi_encode(np.ascontiguousarray(rgb[i])) for i in range(nd)]
        slack = 160                                                   # GAMUT_HIP_QOI_SLACK
        q_begin = np.zeros(max(nq, 1), np.int64); q_size = np.zeros(max(nq, 1), np.int32); parts = []; pos = 0
        for i in range(nq):
            f = files[i % nd]
            q_begin[i] = pos; q_size[i] = len(f); parts += [f, bytes(slack)]; pos += len(f) + slack
        blob = torch.from_numpy(np.frombuffer(b"".join(parts) or bytes(1), np.uint8).copy()).to(dev)
        q_descs = (_capi.QoiDesc * max(nq, 1))()
        for i in range(nq):
            _capi.check(L.gamut_hip_qoi_read_header(files[i % nd], len(files[i % nd]), C.byref(q_descs[i])))
        out = torch.empty((B, h * w * 4), dtype=torch.uint8, device=dev)     # image i of the batch at out[i]
        idx = np.arange(B)
        q_off = (idx[idx % 3 == 2].astype(np.int64) * (h * w * 4))
        # JPEG and PNG outputs go to every third slot as well: image pitch = 3 slots
        status = torch.zeros((max(npn, 1),), dtype=torch.int32, device=dev)
        px_per_step = B * w * h
        fmt_bytes = {"jpeg": nj * (nblk * 128 + w * h * 4), "png": npn * (raw_len + w * h * 4), "qoi": int(q_size[:nq].sum()) + nq * w * h * 4}
        bytes_per_step = sum(fmt_bytes.values())
        kernel_name = "k_jpeg_h2v2 + k_png_defilter + k_qoi_decode"
        workload = (f"mixed batch of {B} x {w}x{h} images -> rgba8, image i: JPEG 4:2:0 / PNG RGBA8 / QOI RGB by i % 3 "
                    f"({nj} + {npn} + {nq}); the QOI launch (a few waves per file, only its INDEX ops are serial) bounds the step"
                    + ("" if args.serial_formats else " and runs on a second stream beside the JPEG and PNG launches"))
        fmt_ev = {k: [] for k in ("jpeg", "png", "qoi")}
        img = h * w * 4

        # The QOI launch keeps a few waves per file busy for as long as its serial walk takes and leaves most of the chip idle: it
        # goes out first, on a second HIP stream, and the JPEG and PNG launches run beside it on the caller's stream, which then
        # waits for it (--serial-formats: one stream, one format after the other, as in round 1).
        main_stream = torch.cuda.current_stream()
        side = None if args.serial_formats else torch.cuda.Stream()
        q_stream = stream if side is None else side.cuda_stream

        def step():
            e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            if side is not None:
                side.wait_stream(main_stream)
                e[3].record(side)
                if nq:
                    _capi.check(L.gamut_hip_qoi_decode_resident_device(blob.data_ptr(), blob.numel(), q_begin.ctypes.data_as(C.POINTER(C.c_int64)), q_size.ctypes.data_as(C.POINTER(C.c_int)),
                                                                       q_descs, nq, 4, q_off.ctypes.data_as(C.POINTER(C.c_int64)), out.data_ptr(), q_stream))
                e[4].record(side)
            e[0].record()
            _capi.check(L.gamut_hip_jpeg_reconstruct_batch_device(coeffs.data_ptr(), nblk * 64, None, 0, out.data_ptr(), w * 4, 3 * img, w, h, 4, 4, nj, stream))
            e[1].record()
            if npn:
                _capi.check(L.gamut_hip_png_defilter_batch_device(raw.data_ptr(), raw_len, raw_len, out.data_ptr() + img, 3 * img, w, h, 4, 4, 8, 6, npn, status.data_ptr(), stream))
            e[2].record()
            if side is None:
                e[3].record()
                if nq:
                    _capi.check(L.gamut_hip_qoi_decode_resident_device(blob.data_ptr(), blob.numel(), q_begin.ctypes.data_as(C.POINTER(C.c_int64)), q_size.ctypes.data_as(C.POINTER(C.c_int)),
                                                                       q_descs, nq, 4, q_off.ctypes.data_as(C.POINTER(C.c_int64)), out.data_ptr(), stream))
                e[4].record()
            else:
                main_stream.wait_stream(side)
            for k, (a, b) in zip(("jpeg", "png", "qoi"), ((e[0], e[1]), (e[1], e[2]), (e[3], e[4]))):
                fmt_ev[k].append((a, b))

        def check():
            step()
            torch.cuda.synchronize()
            fmt_ev["jpeg"].clear(); fmt_ev["png"].clear(); fmt_ev["qoi"].clear()
            assert int(status.abs().sum()) == 0
            exp = O.jpeg_reconstruct(w, h, 3, 4, coeffs[nj - 1].cpu().numpy(), None, 4)
            if not np.array_equal(out[3 * (nj - 1)].cpu().numpy().reshape(exp.shape), exp):
                raise SystemExit("PARITY FAILURE (JPEG)")
            if npn:
                exp = O.png_create_image_raw(raw[npn - 1].cpu().numpy(), 4, 4, w, h, 8, 6)
                if not np.array_equal(out[3 * (npn - 1) + 1].cpu().numpy(), exp):
                    raise SystemExit("PARITY FAILURE (PNG)")
            for k in sorted(set(range(min(nd, nq))) | ({nq - 1} if nq else set())):      # every distinct file once (+ the last image)
                got = out[3 * k + 2].cpu().numpy().reshape(h, w, 4)
                if not (np.array_equal(got[:, :, :3], rgb[k % nd]) and (got[:, :, 3] == 255).all()):
                    raise SystemExit("PARITY FAILURE (QOI)")
                exp, _, _ = O.qoi_decode(files[k % nd], 4)
                if not np.array_equal(got.reshape(exp.shape), exp):
                    raise SystemExit("PARITY FAILURE (QOI vs oracle)")
            # every image of the step: JPEG / PNG checksums against the checked ones' generators are not available, so compare the
            # images that share an input (QOI files repeat every nd) and the PNG source checksums
            if npn:
                got = torch.stack([out[3 * i + 1].to(torch.int64).sum() for i in range(npn)])
                if not torch.equal(got, sums):
                    raise SystemExit("PARITY FAILURE (PNG checksums)")
            for i in range(nq):
                if i >= nd and not torch.equal(out[3 * i + 2], out[3 * (i % nd) + 2]):
                    raise SystemExit(f"PARITY FAILURE (QOI image {i} != its twin {i % nd})")
            check_note["checked"] = "last image of each format == oracle; every PNG == its source checksum; every QOI image == the oracle-checked twin of its file"

        def cpu_leg(seconds):
            if "h" not in _host:
                _host["h"] = (coeffs[0].cpu().numpy(), raw[0].cpu().numpy() if npn else None)
            cj, rp = _host["h"]
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                k = n % 3
                if k == 0: O.jpeg_reconstruct(w, h, 3, 4, cj, None, 4)
                elif k == 1 and npn: O.png_create_image_raw(rp, 4, 4, w, h, 8, 6)
                elif nq: O.qoi_decode(files[0], 4)
                n += 1
            dt = time.perf_counter() - t0
            return n * w * h / dt / 1e6, f"{n} images of the same three kinds in turn (coefficients / filtered stream / QOI file -> rgba8), single thread, {dt:.1f} s"
        dtype = "u8"
    else:
        raise SystemExit(f"unknown workload {wl}")

    if check is not None:                                  # EVERY rank checks its own batch: no rank times unverified output
        if not os.environ.get("GAMUT_BENCH_NOCHECK"):      # experiments with deliberately wrong kernels (tools/variant.sh) only
            check()

    # ------------------------------------------------------------------ timing
    def barrier():
        if world > 1:
            dist.barrier()

    # Untimed, in front of the W warm-up steps: the GPU has idled through the parity check (seconds of host work) and its clocks come back
    # over the first tens of milliseconds of load -- with a 3 ms step the first dozen launches ran up to 40 % slow
    # (profiles/r06_step_ms.txt).  Steps until 150 ms of load have passed (at most 100), none of them counted or timed.
    t_pre = time.perf_counter()
    for _ in range(100):
        step()
        torch.cuda.synchronize()
        if time.perf_counter() - t_pre > 0.15:
            break
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    if os.environ.get("GAMUT_BENCH_STEP_MS") and rank == 0:        # every timed step on stderr (looking for slow steps among fast ones)
        print("[bench] step ms: " + " ".join(f"{x:.3f}" for x in kern_ms), file=sys.stderr)

    gather = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # The only exchange the path has: gathering decoded outputs.  Outside the timed region and reported on
        # its own -- an all-gather of a slice of every rank's output (<= 256 MiB each) over RCCL / xGMI, on a process group of its
        # own that exists only for this.
        try:
            if not args.gather:
                raise StopIteration
            pg = dist.new_group(backend="nccl", device_id=dev) if gather_backend == "nccl" else None
            piece = out.reshape(-1).view(torch.uint8)[:256 << 20].contiguous()
            if gather_backend != "nccl":
                piece = piece.cpu()
            dst = torch.empty(world * piece.numel(), dtype=torch.uint8, device=piece.device)
            dist.all_gather_into_tensor(dst, piece, group=pg)            # warm-up (communicator set-up)
            torch.cuda.synchronize(); dist.barrier()
            t0g = time.perf_counter()
            dist.all_gather_into_tensor(dst, piece, group=pg)
            torch.cuda.synchronize(); dist.barrier()
            dtg = time.perf_counter() - t0g
            ok = bool(torch.equal(dst[rank * piece.numel():(rank + 1) * piece.numel()], piece))
            gather = {"bytes_per_rank": int(piece.numel()), "ms": round(dtg * 1e3, 3), "GB/s_received_per_rank": round((world - 1) * piece.numel() / dtg / 1e9, 1),
                      "own_slice_intact": ok, "backend": "rccl" if gather_backend == "nccl" else "gloo (ranks share a device)",
                      "note": "all_gather of output slices, outside the timed region"}
            del dst, piece
        except StopIteration:
            gather = None
        except Exception as e:                                          # the headline number does not depend on it
            gather = {"error": repr(e)[:200]}

    # N > 1: the other BASELINE.json configs on the same N ranks ("1 vs 8 GPUs", "sharded across 8 MI355X"), each as a sub-run of
    # this script per rank -- fresh HBM, its own parity check on every rank, its own gloo rendezvous on a port rank 0 picks
    also_multi = None
    if world > 1 and wl == "jpeg" and not args.no_also and ((B == 1024 and not (args.width or args.height)) or args.also_args):
        del coeffs, out, zag
        torch.cuda.empty_cache()
        also_multi = also_lines_multi(args, rank, world, dist)

    if rank == 0:
        avg_kernel_s = float(np.mean(kern_ms)) * 1e-3
        achieved = bytes_per_step / avg_kernel_s / 1e9
        traffic, traffic_src = None, None
        if world == 1 and not args.no_traffic and wl != "mixed":
            # the counters are taken on the launch that is TIMED: the same batch (round 4 took them at batch 64, where the PNG kernel's
            # second touch of a line still hit the L2: 1.05 x the algorithmic bytes there, 1.28 x at the timed 512).  convert: layers are
            # independent 268 MB streams read once -- two of them.
            pmc_batch = B if not wl.startswith("convert:") else min(B, 2)
            per_image, traffic_src = live_traffic(kernel_name, pmc_batch)
            traffic = None if per_image is None else round(per_image * B)
        if traffic is None:
            t, tfile = traffic_from_profiles(workload, kernel_name)
            traffic = None if t is None else round(t * B)
            traffic_src = (f"replayed from {tfile} (live pass unavailable: {traffic_src})" if traffic_src else f"replayed from {tfile}") if t is not None else traffic_src
        res = {
            "metric": "Mpixels/sec decoded (batched 1080p JPEG 4:2:0)" if wl == "jpeg" else f"Mpixels/sec ({wl})",
            "value": round(world * px_per_step * args.steps / elapsed / 1e6, 1),
            "unit": "Mpx/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if (wl == "mixed" and args.batch == 1024 and args.total_images) else "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": dict({"workload": workload, "images_per_gpu_per_step": B, "sharding": "image-index, no collective",
                            "parity_check": check_note.get("checked", "see check() of this workload" if not os.environ.get("GAMUT_BENCH_NOCHECK") else "SKIPPED (GAMUT_BENCH_NOCHECK)")},
                           **(zstat if wl.startswith("jpeg") else {})),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "algorithmic_bytes_per_launch": bytes_per_step,
                         "kernel_ms_avg": round(avg_kernel_s * 1e3, 4), "kernel_ms_min": round(min(kern_ms), 4)},
        }
        if gather is not None:
            res["gather"] = gather
        if wl == "mixed":
            torch.cuda.synchronize()
            res["config"]["formats_overlap"] = not args.serial_formats      # per-format times then overlap (QOI on its own stream)
            res["config"]["per_format"] = {}
            for k, cnt in (("jpeg", nj), ("png", npn), ("qoi", nq)):
                ms = float(np.mean([a.elapsed_time(b) for a, b in fmt_ev[k][-args.steps:]])) if fmt_ev[k] else 0.0
                res["config"]["per_format"][k] = {"images": cnt, "ms": round(ms, 4), "Mpx/s": round(cnt * w * h / ms / 1e3, 1) if ms > 0 else None,
                                                 "GB/s": round(fmt_bytes[k] / ms / 1e6, 1) if ms > 0 else None}
        if world == 1 and not args.no_cpu:
            # (i) one thread -- how the reference measures itself (examples/qoix/source/main.d:146-152); (ii) every host core, one
            # image per thread at a time (the oracle is C called through ctypes, which releases the GIL).  SURVEY.md 8d.
            v, sample = cpu_leg(args.cpu_seconds * 0.6)
            res["cpu_baseline"] = {"value": round(v, 2), "unit": "Mpx/s", "cores": 1, "kind": "port", "sample": sample}
            ncores = host_cores()                                # a cgroup CPU quota is the real number of cores this process gets
            if ncores > 1:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(ncores) as pool:
                    t0 = time.perf_counter()
                    rates = list(pool.map(lambda _: cpu_leg(args.cpu_seconds * 0.4)[0], range(ncores)))
                    dt = time.perf_counter() - t0
                res["cpu_baseline"]["all_cores"] = {"value": round(float(sum(rates)), 1), "unit": "Mpx/s", "cores": ncores,
                                                    "sample": f"the same loop on {ncores} threads at once, {dt:.1f} s"}
        else:
            res["cpu_baseline"] = None
        if world == 1 and wl == "jpeg" and not args.no_also and B == 1024 and not (args.width or args.height):
            del coeffs, out, zag                                          # the sub-runs want the HBM
            torch.cuda.empty_cache()
            res["also"] = also_lines(args.also_seconds)
        if also_multi is not None:
            res["also"] = also_multi
        print(json.dumps(res), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
