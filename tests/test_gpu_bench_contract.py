"""bench.py prints ONE JSON line with the agreed keys (short run, no CPU baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, check=True).stdout
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["steps"] == 5 and rec["warmup"] == 2 and rec["n_gpus"] == 1 and rec["unit"] == "images/s"
    assert rec["value"] > 0 and rec["higher_is_better"] is True and rec["vs_baseline"] is None
    assert "workload" in rec["config"] and "model" not in rec["config"]
    roof = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert 0 < rec["roofline_stress"]["frac"] < 1.0
    # the extras beside the headline: other contracts of the same path, and the matcher together with its consumers
    for key in ("flat_path", "per_layer_calls", "per_layer_calls_prefetched", "criterion_step"):
        assert rec[key]["ms_per_step"] > 0, key
    assert rec["criterion_step"]["loss"] > 0


def _bench(*flags):
    """`python bench.py ...` on its own, as the driver runs it: no launcher around it."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags, "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=300, check=True).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out[:400]    # rank 0's JSON line and nothing else on stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_two_ranks():
    """`bench.py --gpus 2` starts its two ranks itself (fences, MAX over ranks, rank 0 prints); here the
    ranks share the one GPU of the test box over gloo, the real runs use one GPU per rank over RCCL."""
    rec = _bench("--gpus", "2", "--backend", "gloo", "--share-gpu")
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["parallelism"] == "dp2"
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 64 * 5 / (rec["ms_per_step"] * 5e-3)) < 1e-6 * rec["value"]
    assert "cpu_baseline" not in rec and "flat_path" not in rec      # rank 0 at N = 1 only


def test_bench_under_an_external_launcher():
    """The same through torch.distributed.run, which sets RANK / LOCAL_RANK / WORLD_SIZE itself."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "2", "--backend", "gloo", "--share-gpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, check=True).stdout
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2"


def test_bench_strong_scaling_splits_the_batch():
    """--scaling strong: the workload's 64 images in total, dealt to the two ranks by target count."""
    rec = _bench("--gpus", "2", "--backend", "gloo", "--share-gpu", "--scaling", "strong")
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong"
    assert "B=32 images per GPU (64 in total)" in rec["config"]["workload"]
    assert abs(rec["value"] - 64 * 5 / (rec["ms_per_step"] * 5e-3)) < 1e-6 * rec["value"]


def test_rehearsal_eight_ranks_weak():
    """What the driver's `python bench.py --gpus 8` runs, minus the hardware: eight self-started ranks (here all on
    the one GPU, over gloo), every rank its own 64 images, fences and the MAX over ranks."""
    rec = _bench("--gpus", "8", "--backend", "gloo", "--share-gpu")
    assert rec["n_gpus"] == 8 and rec["scaling"] == "weak" and rec["config"]["parallelism"] == "dp8"
    assert "B=64 images per GPU (512 in total)" in rec["config"]["workload"]
    assert abs(rec["value"] - 8 * 64 * 5 / (rec["ms_per_step"] * 5e-3)) < 1e-6 * rec["value"]


def test_rehearsal_cfg5_share_of_a_rank():
    """cfg5 as BASELINE states it, 128 images per GPU (the most the in-place target table holds), four ranks."""
    rec = _bench("--gpus", "4", "--backend", "gloo", "--share-gpu", "--workload", "cfg5", "--batch", "128")
    assert rec["n_gpus"] == 4 and "B=128 images per GPU (512 in total)" in rec["config"]["workload"]
    assert rec["step_kernels"]["one_kernel_per_step"] is True


def test_rehearsal_cfg4_strong_over_eight_ranks():
    """cfg4's 256 images dealt to eight ranks by target count (32 each): the split BASELINE's config 4 describes."""
    rec = _bench("--gpus", "8", "--backend", "gloo", "--share-gpu", "--workload", "cfg4", "--scaling", "strong")
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong"
    assert "B=32 images per GPU (256 in total)" in rec["config"]["workload"]
    assert abs(rec["value"] - 256 * 5 / (rec["ms_per_step"] * 5e-3)) < 1e-6 * rec["value"]


def test_a_rank_that_dies_ends_the_run():
    """A rank that dies before the first fence: the parent ends the ranks that wait for it and exits non-zero,
    without a JSON line, well inside its timeout."""
    import time

    t0 = time.time()
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--share-gpu",
                          "--steps", "5", "--warmup", "2"], capture_output=True, text=True, timeout=200,
                         env=dict(os.environ, DETR_AMD_BENCH_DIE_RANK="2"))
    assert run.returncode != 0 and "rank 2 exited with status 7" in run.stderr
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 120


def test_bench_two_ranks_over_rccl():
    """One GPU per rank over RCCL (backend nccl): runs wherever the box has two GPUs."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    rec = _bench("--gpus", "2")
    assert rec["n_gpus"] == 2 and rec["value"] > 0


_RCCL_ONE_RANK = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import detr_amd
from detr_amd import synth
from detr_amd.dist import matched_box_counts, start_num_boxes_all_reduce
wl, sets, targets = synth.make_workload("cfg2", device=dev, batch=8)
counts = matched_box_counts([len(t["labels"]) for t in targets], wl.set_queries) + [0]
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
plain = m.match_many(sets, targets)
pending = []
for _ in range(12):      # more steps than the helper's rings of staging buffers and events hold
    hooked = m.match_many(sets, [dict(t) for t in targets],
                          while_waiting=lambda: pending.append(start_num_boxes_all_reduce(counts, dev, _always_reduce=True)))
    got = pending[-1].result()
    assert got.device == dev and got.tolist() == [float(max(c, 1)) for c in counts], got
    for a, b in zip(plain, hooked):
        for (a0, a1), (b0, b1) in zip(a, b):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)
assert len(pending) == 12
torch.cuda.synchronize()
dist.destroy_process_group()
print("ok")
'''


def test_collective_beside_the_kernels_over_rccl_one_rank():
    """The step's collective issued from match_many's `while_waiting` hook, on its own stream, through RCCL (a group of
    one rank: the box has one GPU): right values on the caller's stream, indices unchanged."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, ROOT, str(port)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
