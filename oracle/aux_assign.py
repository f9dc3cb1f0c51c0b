"""CPU restatement (numpy, float32) of the Co-DETR auxiliary label assigners -- TEST
INFRASTRUCTURE ONLY: never imported by the product path.

Follows /root/reference/models/co_detr/auxiliary_head.py:
  atss          :22-60   (k closest predictions per target by centre distance, IoU threshold
                          mean + std of those k, centre-inside-target test)
  faster_rcnn   :69-96   (positives: IoU > pos_threshold or the best prediction of a target;
                          negatives: predictions whose best IoU is < neg_threshold)
with IoU = torchvision.ops.box_iou's documented definition, inter / (area1 + area2 - inter).

Where the reference leaves the result to implementation detail, this restatement fixes it:
  * the centre distance is sqrt(dx^2 + dy^2) computed directly (torch.cdist switches to a
    matmul formulation above 25 rows whose cancellation error reorders near-equal distances);
    equal distances are ordered by query index (torch.topk leaves ties unspecified);
  * pairs of one target are listed in ascending query index for faster_rcnn (the reference's
    order is the iteration order of a Python set).
Pinned by tests/golden/aux/*.npz, produced by running the reference's own module.
"""
import numpy as np

F = np.float32


def _xyxy(b):
    b = np.asarray(b, dtype=F).reshape(-1, 4)
    return np.stack([b[:, 0] - F(0.5) * b[:, 2], b[:, 1] - F(0.5) * b[:, 3],
                     b[:, 0] + F(0.5) * b[:, 2], b[:, 1] + F(0.5) * b[:, 3]], axis=1)


def box_iou(a, b):
    """[len(a), len(b)] IoU of xyxy boxes."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.maximum(rb - lt, F(0))
    inter = wh[..., 0] * wh[..., 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / (area_a[:, None] + area_b[None, :] - inter)).astype(F)


def centre_distance(gt_boxes, pred_boxes):
    d = np.asarray(gt_boxes, F).reshape(-1, 4)[:, None, :2] - np.asarray(pred_boxes, F).reshape(-1, 4)[None, :, :2]
    return np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(F)


def _seq_sum(x):
    """fp32 sum in index order (the order is part of the definition: the threshold is compared with >=)"""
    acc = F(0)
    for v in x:
        acc = F(acc + v)
    return acc


def atss_image(pred_boxes, gt_boxes, k):
    """(src, tgt) int64 arrays of one image, auxiliary_head.py:29-58."""
    pred_boxes = np.asarray(pred_boxes, F).reshape(-1, 4)
    gt_boxes = np.asarray(gt_boxes, F).reshape(-1, 4)
    src, tgt = [], []
    if gt_boxes.shape[0] == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    if k > pred_boxes.shape[0]:
        raise RuntimeError("selected index k out of range")
    dist = centre_distance(gt_boxes, pred_boxes)
    for i, gt in enumerate(gt_boxes):
        cand = np.argsort(dist[i], kind="stable")[:k]
        ious = box_iou(_xyxy(gt[None]), _xyxy(pred_boxes[cand]))[0]
        mean = _seq_sum(ious) / F(k)
        with np.errstate(divide="ignore", invalid="ignore"):
            dev = ious - mean
            std = np.sqrt(_seq_sum(dev * dev) / F(k - 1)) if k > 1 else F(np.nan)
        thr = mean + std
        x1, y1, x2, y2 = _xyxy(gt[None])[0]
        for j, q in enumerate(cand):
            cx, cy = pred_boxes[q, 0], pred_boxes[q, 1]
            if ious[j] >= thr and x1 <= cx <= x2 and y1 <= cy <= y2:
                src.append(int(q))
                tgt.append(i)
    return np.asarray(src, np.int64), np.asarray(tgt, np.int64)


def atss(pred_boxes_bq4, target_boxes, k):
    return [atss_image(p, g, k) for p, g in zip(pred_boxes_bq4, target_boxes)]


def faster_rcnn_image(pred_boxes, gt_boxes, pos_threshold=0.7, neg_threshold=0.3):
    """((src, tgt), neg) of one image, auxiliary_head.py:74-94."""
    pred_boxes = np.asarray(pred_boxes, F).reshape(-1, 4)
    gt_boxes = np.asarray(gt_boxes, F).reshape(-1, 4)
    if gt_boxes.shape[0] == 0:
        return (np.zeros(0, np.int64), np.zeros(0, np.int64)), np.arange(pred_boxes.shape[0], dtype=np.int64)
    ious = box_iou(_xyxy(gt_boxes), _xyxy(pred_boxes))
    neg = np.nonzero(ious.max(axis=0) < F(neg_threshold))[0].astype(np.int64)
    src, tgt = [], []
    for i, row in enumerate(ious):
        pos = set(np.nonzero(row > F(pos_threshold))[0].tolist())
        pos.add(int(np.argmax(row)))
        for q in sorted(pos):
            src.append(q)
            tgt.append(i)
    return (np.asarray(src, np.int64), np.asarray(tgt, np.int64)), neg


def faster_rcnn(pred_boxes_bq4, target_boxes, pos_threshold=0.7, neg_threshold=0.3):
    res = [faster_rcnn_image(p, g, pos_threshold, neg_threshold) for p, g in zip(pred_boxes_bq4, target_boxes)]
    return [r[0] for r in res], [r[1] for r in res]


def faster_rcnn_class_targets(pos, neg, labels, Q, n_cls):
    """[B, Q] int64 class targets and the pair count, /root/reference/models/co_detr/loss.py:151-159:
    -1 everywhere, then the labels of the positive pairs in list order (a later pair overwrites an
    earlier one on the same prediction, as an in-order scatter does), then n_cls on the negatives."""
    B = len(neg)
    out = np.full((B, Q), -1, dtype=np.int64)
    pairs = 0
    for b in range(B):
        src, tgt = pos[b]
        for q, t in zip(src.tolist(), tgt.tolist()):
            out[b, q] = labels[b][t]
        pairs += len(src)
        out[b, neg[b]] = n_cls
    return out, pairs
