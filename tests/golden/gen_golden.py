#!/usr/bin/env python3
"""Generates the golden vectors in this directory by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the resulting
``*.npz`` files are committed and are the only thing that travels.

How the reference is loaded: ``models/matcher.py`` and ``utils/box_ops.py``
are executed from /root/reference by path.  Importing ``models`` as a package
would pull in torchvision backbones, and ``utils/box_ops.py:14`` imports
``torchvision.ops.boxes.box_area``; torchvision is not installed in the
container, so ``box_area`` is provided by a one-line stand-in with
torchvision's documented definition, ``(x1 - x0) * (y1 - y0)``.  Every other
instruction that produces these vectors is the reference's own code plus the
installed torch 2.10 and scipy 1.15.3.

The per-image cost blocks are captured by wrapping the
``linear_sum_assignment`` name inside the loaded matcher module: the wrapper
records the ``[Q, T_b]`` argument (matcher.py:94) and delegates to scipy.

Usage:  python tests/golden/gen_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_reference():
    tv, ops, bx = (types.ModuleType(n) for n in ("torchvision", "torchvision.ops", "torchvision.ops.boxes"))
    bx.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    ops.boxes = bx
    tv.ops = ops
    sys.modules.update({"torchvision": tv, "torchvision.ops": ops, "torchvision.ops.boxes": bx})
    pkg = types.ModuleType("utils")
    pkg.__path__ = []
    sys.modules["utils"] = pkg

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    load("utils.box_ops", os.path.join(REF, "utils/box_ops.py"))
    return load("ref_matcher", os.path.join(REF, "models/matcher.py"))


def boxes(n, gen):
    return torch.cat([torch.rand(n, 2, generator=gen), torch.rand(n, 2, generator=gen) * 0.30 + 0.01], -1)


def regular_case(seed, cls_loss, C, n_label_classes, B, Q, sizes, weights):
    gen = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, Q, C, generator=gen) * 2.0
    pboxes = boxes(B * Q, gen).view(B, Q, 4)
    targets = [{"labels": torch.randint(0, n_label_classes, (t,), generator=gen), "boxes": boxes(t, gen)}
               for t in sizes]
    return dict(cls_loss=cls_loss, weights=weights, logits=logits, pboxes=pboxes, targets=targets)


def build_cases():
    cases = {}
    S = [16, 0, 37, 5]
    cases["ce_c3_q50"] = regular_case(11, "ce", 3, 2, 4, 50, S, (2.0, 5.0, 2.0))
    cases["ce_c3_q600"] = regular_case(12, "ce", 3, 2, 4, 600, S, (2.0, 5.0, 2.0))
    cases["ce_c3_q300_w1"] = regular_case(13, "ce", 3, 2, 4, 300, S, (1.0, 5.0, 2.0))
    cases["focal_c2_q50"] = regular_case(21, "focal", 2, 2, 4, 50, S, (2.0, 5.0, 2.0))
    cases["focal_c2_q300"] = regular_case(22, "focal", 2, 2, 4, 300, S, (2.0, 5.0, 2.0))
    cases["focal_c2_q600_w1"] = regular_case(23, "focal", 2, 2, 4, 600, S, (1.0, 5.0, 2.0))
    cases["focal_c91_q100"] = regular_case(31, "focal", 91, 91, 4, 100, S, (2.0, 5.0, 2.0))
    cases["ce_c92_q100"] = regular_case(32, "ce", 92, 91, 4, 100, S, (1.0, 5.0, 2.0))
    # non-default focal parameters (general pow path)
    c = regular_case(33, "focal", 4, 4, 2, 64, [9, 3], (2.0, 5.0, 2.0))
    c["alpha"], c["gamma"] = 0.4, 1.5
    cases["focal_c4_gamma15"] = c
    # more targets than queries, and exactly as many
    cases["focal_t_gt_q"] = regular_case(41, "focal", 2, 2, 3, 5, [9, 5, 2], (2.0, 5.0, 2.0))
    cases["ce_t_gt_q"] = regular_case(42, "ce", 3, 2, 2, 7, [12, 7], (2.0, 5.0, 2.0))
    # all images empty
    cases["all_empty"] = regular_case(43, "ce", 3, 2, 3, 20, [0, 0, 0], (2.0, 5.0, 2.0))

    # duplicated ground-truth boxes -> exact ties between target rows
    c = regular_case(51, "focal", 2, 2, 2, 80, [8, 6], (2.0, 5.0, 2.0))
    for t in c["targets"]:
        t["boxes"][1::2] = t["boxes"][0::2][: len(t["boxes"][1::2])]
        t["labels"][1::2] = t["labels"][0::2][: len(t["labels"][1::2])]
    cases["dup_gt"] = c
    # identical prediction rows: zero-initialised box heads give every query
    # (0.5, 0.5, 0.5, 0.5) and equal logits at step 0 (detr/detr.py:70-71,96)
    c = regular_case(52, "ce", 3, 2, 2, 40, [6, 11], (2.0, 5.0, 2.0))
    c["pboxes"][:] = 0.5
    c["logits"][:] = 0.0
    cases["identical_preds"] = c
    # partly identical predictions (blocks of equal queries)
    c = regular_case(53, "focal", 2, 2, 2, 60, [7, 10], (2.0, 5.0, 2.0))
    c["pboxes"][:, 10:40] = c["pboxes"][:, 10:11]
    c["logits"][:, 10:40] = c["logits"][:, 10:11]
    cases["blocks_of_equal_preds"] = c
    # masked first-stage proposals are (0, 0, 0, 0) boxes (dino_detr/transformer.py:162-164)
    c = regular_case(54, "focal", 2, 2, 2, 90, [5, 13], (2.0, 5.0, 2.0))
    c["pboxes"][:, 60:] = 0.0
    cases["masked_first_stage"] = c

    # the stress shapes: a slice of the COCO-like workload (91 classes, 900 queries, up to 100 targets) ...
    cases["focal_c91_q900_t100"] = regular_case(55, "focal", 91, 91, 4, 900, [0, 3, 64, 100], (2.0, 5.0, 2.0))
    # ... a first stage over all 25 x 42 encoder tokens, the padded ones masked to (0, 0, 0, 0) ...
    c = regular_case(56, "focal", 2, 2, 2, 1050, [12, 30], (2.0, 5.0, 2.0))
    c["pboxes"][0, 800:] = 0.0
    c["pboxes"][1, 1000:] = 0.0
    cases["focal_c2_q1050_first_stage"] = c
    # ... and saturated logits (sigma = 8): sigmoid(x) rounds to 0 or 1 for many entries, the regime in which
    # the reference's log(1 - p + 1e-8) depends on the last bit of p
    c = regular_case(57, "focal", 2, 2, 2, 100, [9, 14], (2.0, 5.0, 2.0))
    c["logits"] *= 4.0
    cases["focal_c2_saturated"] = c

    # class heads beyond what a 64-row logit tile in LDS holds (an LVIS-size head): the kernel that reads
    # the logits from memory
    cases["focal_c1203_q70"] = regular_case(58, "focal", 1203, 1203, 3, 70, [11, 0, 25], (2.0, 5.0, 2.0))
    cases["ce_c1204_q70"] = regular_case(59, "ce", 1204, 1203, 3, 70, [11, 0, 25], (1.0, 5.0, 2.0))

    # error cases: the reference must raise
    c = regular_case(61, "ce", 3, 2, 2, 30, [4, 3], (2.0, 5.0, 2.0))
    c["logits"][1, 7, 1] = float("nan")
    cases["err_nan_logit"] = c
    c = regular_case(62, "ce", 3, 2, 2, 30, [4, 3], (2.0, 5.0, 2.0))
    c["pboxes"][0, 3, 2] = float("nan")
    cases["err_nan_pred_box"] = c
    c = regular_case(63, "focal", 2, 2, 2, 30, [4, 3], (2.0, 5.0, 2.0))
    c["pboxes"][1, 2, 2] = -0.1
    cases["err_negative_width"] = c
    c = regular_case(64, "focal", 2, 2, 2, 30, [4, 3], (2.0, 5.0, 2.0))
    c["targets"][1]["boxes"][0, 3] = -0.2
    cases["err_negative_gt_height"] = c
    c = regular_case(65, "focal", 2, 2, 2, 30, [4, 3], (2.0, 5.0, 2.0))
    c["pboxes"][0, 5] = 0.0
    c["targets"][0]["boxes"][2] = 0.0      # zero-area prediction on a zero-area target: 0/0
    cases["err_zero_over_zero"] = c
    c = regular_case(66, "focal", 2, 2, 2, 30, [4, 3], (2.0, 5.0, 2.0))
    c["targets"][0]["labels"][1] = 2
    cases["err_label_out_of_range"] = c
    # degenerate prediction in an image WITHOUT targets still trips the batch-wide assert
    c = regular_case(67, "focal", 2, 2, 2, 30, [0, 3], (2.0, 5.0, 2.0))
    c["pboxes"][0, 2, 3] = -0.3
    cases["err_negative_height_in_empty_image"] = c
    # NaN logit in an image without targets never reaches scipy: no error
    c = regular_case(68, "ce", 3, 2, 2, 30, [0, 3], (2.0, 5.0, 2.0))
    c["logits"][0, 2, 0] = float("nan")
    cases["nan_logit_in_empty_image"] = c
    return cases


def run_case(ref, case):
    blocks = []
    real = ref.linear_sum_assignment

    def recording(c):
        blocks.append(c.detach().numpy().astype(np.float32, copy=True))
        return real(c)

    ref.linear_sum_assignment = recording
    w = case["weights"]
    matcher = ref.HungarianMatcher(cls_loss=case["cls_loss"], cost_class=w[0], cost_bbox=w[1], cost_giou=w[2],
                                   focal_alpha=case.get("alpha", 0.25), focal_gamma=case.get("gamma", 2.0))
    err = ""
    indices = []
    try:
        indices = matcher({"pred_logits": case["logits"], "pred_boxes": case["pboxes"]}, case["targets"])
    except Exception as e:  # noqa: BLE001 -- the exception TYPE is the golden value
        err = type(e).__name__
    finally:
        ref.linear_sum_assignment = real
    return blocks, indices, err


def main():
    ref = load_reference()
    index = {}
    for name, case in build_cases().items():
        blocks, indices, err = run_case(ref, case)
        sizes = np.array([len(t["labels"]) for t in case["targets"]], dtype=np.int64)
        arrays = dict(
            logits=case["logits"].numpy(), pboxes=case["pboxes"].numpy(), sizes=sizes,
            labels=np.concatenate([t["labels"].numpy() for t in case["targets"]]).astype(np.int64),
            tboxes=np.concatenate([t["boxes"].numpy().reshape(-1, 4) for t in case["targets"]]).astype(np.float32),
            meta=np.array(json.dumps(dict(cls_loss=case["cls_loss"], weights=list(case["weights"]),
                                          alpha=case.get("alpha", 0.25), gamma=case.get("gamma", 2.0),
                                          error=err))),
        )
        if not err:
            # cost blocks [Q, T_b] flattened and concatenated (all images, T_b = 0 gives an empty block)
            arrays["cost"] = np.concatenate([b.reshape(-1) for b in blocks]) if blocks else np.zeros(0, np.float32)
            arrays["src"] = np.concatenate([i.numpy() for i, _ in indices]).astype(np.int64)
            arrays["tgt"] = np.concatenate([j.numpy() for _, j in indices]).astype(np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        index[name] = dict(error=err, sizes=sizes.tolist(), Q=int(case["logits"].shape[1]),
                           C=int(case["logits"].shape[2]))
        print(f"{name:40s} err={err or '-':16s} pairs={0 if err else len(arrays['src'])}")
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(dict(torch=torch.__version__, scipy=__import__("scipy").__version__, cases=index), f, indent=1)


if __name__ == "__main__":
    main()
