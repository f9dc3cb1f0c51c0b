"""The C ABI from a plain C program (tests/c_abi/abi_demo.c, built by build_ext.py with gcc against
include/detr_match.h, the HIP runtime and libdetr_match.so -- no torch, no Python in that process): the same indices
as the Python matcher on the same inputs, through dm_match_targets_f32 and through dm_pack_targets + dm_match_f32."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "tests", "c_abi", "abi_demo")


def test_demo_is_built():
    assert os.path.exists(DEMO), "python build_ext.py builds tests/c_abi/abi_demo"


def _fixture(path, wl, sets, targets, result):
    S, (B, Q, C) = len(sets), sets[0]["pred_logits"].shape
    sizes = [len(t["labels"]) for t in targets]
    M = sum(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    kw = wl.matcher_kwargs
    src = np.full((S, M), -7, dtype=np.int64)
    tgt = np.full((S, M), -7, dtype=np.int64)
    for s in range(S):
        for b, (i, j) in enumerate(result[s]):
            src[s, off[b]:off[b] + len(i)] = i.numpy()
            tgt[s, off[b]:off[b] + len(j)] = j.numpy()
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", S, B, Q, C, 0 if kw["cls_loss"] == "ce" else 1, M))
        f.write(struct.pack("<5f", kw["cost_class"], kw["cost_bbox"], kw["cost_giou"], kw.get("focal_alpha", 0.25),
                            kw.get("focal_gamma", 2.0)))
        f.write(np.asarray(sizes, dtype=np.int32).tobytes())
        f.write(np.stack([o["pred_logits"].cpu().numpy() for o in sets]).astype(np.float32).tobytes())
        f.write(np.stack([o["pred_boxes"].cpu().numpy() for o in sets]).astype(np.float32).tobytes())
        f.write(np.concatenate([t["labels"].cpu().numpy() for t in targets] + [np.zeros(0, np.int64)]).astype(np.int64).tobytes())
        f.write(np.concatenate([t["boxes"].cpu().numpy().reshape(-1, 4) for t in targets] +
                               [np.zeros((0, 4), np.float32)]).astype(np.float32).tobytes())
        f.write(src.tobytes())
        f.write(tgt.tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch,n_sets,queries", [
    ("cfg2", 8, 3, None),        # one kernel, targets read in place
    ("cfg2", 130, 2, None),      # more images than the in-place table holds: packed through the pinned table
    ("cfg4", 5, 2, None),        # 91 classes: cost kernel + assignment kernel
    ("cfg1", 6, 2, 12),          # images with more targets than queries: the list-order kernel
])
def test_plain_c_program_gets_the_matchers_indices(tmp_path, name, batch, n_sets, queries):
    import dataclasses

    import detr_amd
    from detr_amd import synth

    if queries is not None:
        base = synth.WORKLOADS[name]
        synth.WORKLOADS["_c_abi"] = dataclasses.replace(base, name="_c_abi", set_queries=[queries] * n_sets)
        name = "_c_abi"
    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    sets = sets[:n_sets]
    result = detr_amd.HungarianMatcher(**wl.matcher_kwargs).match_many(sets, targets)
    path = str(tmp_path / "fixture.bin")
    _fixture(path, wl, sets, targets, result)
    del sets, targets
    torch.cuda.synchronize()
    # the C program brings its own HIP runtime (/opt/rocm's, by rpath): nothing of this process's torch leaks into it
    env = {k: v for k, v in os.environ.items() if k != "LD_LIBRARY_PATH"}
    run = subprocess.run([DEMO, path], capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), (run.stdout[-600:], run.stderr[-600:])
    assert "pass 0" in run.stdout and "pass 1" in run.stdout
