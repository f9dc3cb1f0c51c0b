#!/usr/bin/env python3
"""Golden vectors for inputs that are not float32, made by RUNNING THE REFERENCE (see gen_golden.py for how it is
loaded; build container only).  The reference's matcher computes in whatever dtype it is handed
(/root/reference/models/matcher.py:59-90): float64 predictions and targets give float64 cost blocks; float64 logits
beside float32 boxes promote the sum to float64; float64 prediction boxes against float32 target boxes raise
RuntimeError (torch.cdist wants one dtype); float16 / bfloat16 raise NotImplementedError on the CPU.

Writes tests/golden/dtype/*.npz: inputs in their dtype, the float64 cost blocks the reference handed to scipy, the
indices, or the exception's type name.

Usage:  python tests/golden/gen_golden_dtypes.py
"""
import json
import os

import numpy as np
import torch

import gen_golden as g

OUT = os.path.join(g.HERE, "dtype")


def cast(case, logits=None, pboxes=None, tboxes=None):
    if logits is not None:
        case["logits"] = case["logits"].to(logits)
    if pboxes is not None:
        case["pboxes"] = case["pboxes"].to(pboxes)
    if tboxes is not None:
        for t in case["targets"]:
            t["boxes"] = t["boxes"].to(tboxes)
    return case


def finer(case):
    """float64 values that float32 cannot hold (the float32 draw plus a float64 perturbation far below its ulp)."""
    gen = torch.Generator().manual_seed(99)
    for key in ("logits", "pboxes"):
        x = case[key]
        if x.dtype == torch.float64:
            case[key] = x * (1.0 + 1e-9 * torch.rand(x.shape, generator=gen, dtype=torch.float64))
    for t in case["targets"]:
        if t["boxes"].dtype == torch.float64:
            t["boxes"] = t["boxes"] * (1.0 + 1e-9 * torch.rand(t["boxes"].shape, generator=gen, dtype=torch.float64))
    return case


def build_cases():
    f64, f32 = torch.float64, torch.float32
    S = [16, 0, 37, 5]
    cases = {}
    cases["f64_focal_c2_q300"] = finer(cast(g.regular_case(81, "focal", 2, 2, 4, 300, S, (2.0, 5.0, 2.0)), f64, f64, f64))
    cases["f64_ce_c3_q100"] = finer(cast(g.regular_case(82, "ce", 3, 2, 4, 100, S, (2.0, 5.0, 2.0)), f64, f64, f64))
    cases["f64_focal_c91_q100"] = finer(cast(g.regular_case(83, "focal", 91, 91, 4, 100, S, (2.0, 5.0, 2.0)), f64, f64, f64))
    cases["f64_logits_only_focal_c2_q50"] = finer(cast(g.regular_case(84, "focal", 2, 2, 4, 50, S, (2.0, 5.0, 2.0)), f64))
    cases["f64_t_gt_q"] = finer(cast(g.regular_case(85, "ce", 3, 2, 2, 7, [12, 7], (2.0, 5.0, 2.0)), f64, f64, f64))
    cases["err_f64_preds_f32_targets"] = cast(g.regular_case(86, "focal", 2, 2, 3, 20, [4, 0, 6], (2.0, 5.0, 2.0)), f64, f64)
    cases["err_f32_preds_f64_targets"] = cast(g.regular_case(87, "focal", 2, 2, 3, 20, [4, 0, 6], (2.0, 5.0, 2.0)), None, None, f64)
    cases["err_f16"] = cast(g.regular_case(88, "focal", 2, 2, 3, 20, [4, 0, 6], (2.0, 5.0, 2.0)), torch.float16, torch.float16, torch.float16)
    cases["err_bf16"] = cast(g.regular_case(89, "ce", 3, 2, 3, 20, [4, 0, 6], (2.0, 5.0, 2.0)), torch.bfloat16, torch.bfloat16, torch.bfloat16)
    return cases


def run_case(ref, case):
    blocks, real = [], ref.linear_sum_assignment

    def recording(c):
        blocks.append(c.detach().to(torch.float64).numpy().copy())
        return real(c)

    ref.linear_sum_assignment = recording
    w = case["weights"]
    matcher = ref.HungarianMatcher(cls_loss=case["cls_loss"], cost_class=w[0], cost_bbox=w[1], cost_giou=w[2])
    err, indices = "", []
    try:
        indices = matcher({"pred_logits": case["logits"], "pred_boxes": case["pboxes"]}, case["targets"])
    except Exception as e:  # noqa: BLE001 -- the exception TYPE is the golden value
        err = type(e).__name__
    finally:
        ref.linear_sum_assignment = real
    return blocks, indices, err


def as_numpy(x):
    # numpy has no bfloat16: such inputs are stored as float32 with the dtype's name beside them
    return x.to(torch.float32).numpy() if x.dtype == torch.bfloat16 else x.numpy()


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = g.load_reference()
    for name, case in build_cases().items():
        blocks, indices, err = run_case(ref, case)
        sizes = np.array([len(t["labels"]) for t in case["targets"]], dtype=np.int64)
        tb = torch.cat([t["boxes"].reshape(-1, 4) for t in case["targets"]])
        arrays = dict(
            logits=as_numpy(case["logits"]), pboxes=as_numpy(case["pboxes"]), sizes=sizes,
            labels=np.concatenate([t["labels"].numpy() for t in case["targets"]]).astype(np.int64), tboxes=as_numpy(tb),
            meta=np.array(json.dumps(dict(cls_loss=case["cls_loss"], weights=list(case["weights"]), error=err,
                                          dtypes=[str(case["logits"].dtype), str(case["pboxes"].dtype), str(tb.dtype)]))))
        if not err:
            arrays["cost"] = np.concatenate([b.reshape(-1) for b in blocks]) if blocks else np.zeros(0)
            arrays["src"] = np.concatenate([i.numpy() for i, _ in indices]).astype(np.int64)
            arrays["tgt"] = np.concatenate([j.numpy() for _, j in indices]).astype(np.int64)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        print(f"{name:36s} err={err or '-':20s} pairs={0 if err else len(arrays['src'])} "
              f"cost dtype={blocks[0].dtype if blocks else '-'}")


if __name__ == "__main__":
    main()
