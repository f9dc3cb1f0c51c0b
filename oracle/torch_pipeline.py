"""The reference's CPU formulation restated with stock torch ops + scipy -- TEST INFRASTRUCTURE ONLY.

This is what ``bench.py`` times as ``cpu_baseline`` (kind "port"): the same
sequence of whole-batch tensor operations the reference matcher executes on a
CPU device, followed by one scipy LAP per image.

  * whole-batch [B*Q, sum(T)] cost matrix:      /root/reference/models/matcher.py:59-91
  * pairwise GIoU with [N, M, 2] temporaries:    /root/reference/utils/box_ops.py:32-69
  * per-image scipy.optimize.linear_sum_assignment on the diagonal blocks:
                                                 /root/reference/models/matcher.py:93-95

Unlike oracle_match.c (per-image evaluation) this keeps the reference's
B-times-oversized batched matrix, because that is where the reference's CPU
time goes.
"""
import torch
from scipy.optimize import linear_sum_assignment


def _corners(b):
    # utils/box_ops.py:17-21
    half_w, half_h = 0.5 * b[..., 2], 0.5 * b[..., 3]
    return torch.stack((b[..., 0] - half_w, b[..., 1] - half_h, b[..., 0] + half_w, b[..., 1] + half_h), dim=-1)


def _pairwise_giou(p, g):
    # utils/box_ops.py:48-69 (with box_iou :32-45 inlined)
    assert bool((p[:, 2:] >= p[:, :2]).all())
    assert bool((g[:, 2:] >= g[:, :2]).all())
    area_p = (p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 1])
    area_g = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    inner = (torch.min(p[:, None, 2:], g[:, 2:]) - torch.max(p[:, None, :2], g[:, :2])).clamp(min=0)
    inter = inner[..., 0] * inner[..., 1]
    union = area_p[:, None] + area_g - inter
    iou = inter / union
    outer = (torch.max(p[:, None, 2:], g[:, 2:]) - torch.min(p[:, None, :2], g[:, :2])).clamp(min=0)
    hull = outer[..., 0] * outer[..., 1]
    return iou - (hull - union) / hull


@torch.no_grad()
def batched_cost(pred_logits, pred_boxes, labels, tboxes, cls_loss="ce", cost_class=1.0, cost_bbox=1.0,
                 cost_giou=1.0, focal_alpha=0.25, focal_gamma=2.0):
    """[B, Q, sum(T)] cost tensor exactly as matcher.py:59-91 builds it."""
    B, Q = pred_logits.shape[:2]
    flat_boxes = pred_boxes.reshape(B * Q, 4)
    flat_logits = pred_logits.reshape(B * Q, -1)
    if cls_loss == "ce":
        cls_term = -flat_logits.softmax(-1)[:, labels]
    else:
        p = flat_logits.sigmoid()
        neg = (1 - focal_alpha) * (p ** focal_gamma) * (-(1 - p + 1e-8).log())
        pos = focal_alpha * ((1 - p) ** focal_gamma) * (-(p + 1e-8).log())
        cls_term = pos[:, labels] - neg[:, labels]
    l1_term = torch.cdist(flat_boxes, tboxes, p=1)
    giou_term = -_pairwise_giou(_corners(flat_boxes), _corners(tboxes))
    total = cost_bbox * l1_term + cost_class * cls_term + cost_giou * giou_term
    return total.view(B, Q, -1)


@torch.no_grad()
def match(pred_logits, pred_boxes, targets, **kw):
    """list of (src, tgt) int64 CPU tensors, one per image (matcher.py:93-95)."""
    labels = torch.cat([t["labels"] for t in targets])
    tboxes = torch.cat([t["boxes"] for t in targets])
    cost = batched_cost(pred_logits, pred_boxes, labels, tboxes, **kw).cpu()
    sizes = [len(t["boxes"]) for t in targets]
    out = []
    for b, blk in enumerate(cost.split(sizes, -1)):
        r, c = linear_sum_assignment(blk[b])
        out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(c, dtype=torch.int64)))
    return out
