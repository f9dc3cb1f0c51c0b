"""CPU restatement (numpy, float32) of what the reference's criterion derives from the matching
indices -- TEST INFRASTRUCTURE ONLY: never imported by the product path.

Follows /root/reference/models/detr/loss.py:
  src_permutation_idx   :71-75    (batch index, query index of every matched pair)
  box_losses            :97-114   (L1 and GIoU loss of the matched pairs, each / num_boxes)
  target_classes        :121-125  (dense [B, Q] class map, no-object everywhere else)
and /root/reference/utils/box_ops.py:17-21, 32-45, 48-69 for the pair geometry.  The gradients
are the analytic derivatives with torch autograd's conventions (maximum/minimum split the
gradient evenly at ties, clamp passes it at the boundary, sign(0) = 0).

Pinned by tests/golden/loss/*.npz, produced by running the reference's own methods.
"""
import numpy as np

F = np.float32


def src_permutation_idx(indices):
    batch_idx = np.concatenate([np.full_like(src, b) for b, (src, _) in enumerate(indices)] + [np.zeros(0, np.int64)])
    src_idx = np.concatenate([src for src, _ in indices] + [np.zeros(0, np.int64)])
    return batch_idx.astype(np.int64), src_idx.astype(np.int64)


def matched_pairs(pred_boxes, target_boxes, indices):
    bi, si = src_permutation_idx(indices)
    sel = pred_boxes[bi, si].astype(F).reshape(-1, 4)
    tgt = np.concatenate([np.asarray(tb, dtype=F).reshape(-1, 4)[j] for tb, (_, j) in zip(target_boxes, indices)] +
                         [np.zeros((0, 4), F)])
    return bi, si, sel, tgt


def _xyxy(b):
    cx, cy, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return cx - F(0.5) * w, cy - F(0.5) * h, cx + F(0.5) * w, cy + F(0.5) * h


def _hi_share(a, b):
    """share of the gradient of max(a, b) that goes to a"""
    return (a > b).astype(F) + F(0.5) * (a == b).astype(F)


def _lo_share(a, b):
    """share of the gradient of min(a, b) that goes to a"""
    return (a < b).astype(F) + F(0.5) * (a == b).astype(F)


def pair_giou(sel, tgt):
    """GIoU of pair i (the diagonal of generalized_box_iou, loss.py:108-111) and its gradient
    w.r.t. the cxcywh prediction."""
    ax0, ay0, ax1, ay1 = _xyxy(sel)
    bx0, by0, bx1, by1 = _xyxy(tgt)
    area_a = (ax1 - ax0) * (ay1 - ay0)
    area_b = (bx1 - bx0) * (by1 - by0)
    ltx, lty = np.maximum(ax0, bx0), np.maximum(ay0, by0)
    rbx, rby = np.minimum(ax1, bx1), np.minimum(ay1, by1)
    dx, dy = rbx - ltx, rby - lty
    wx, wy = np.maximum(dx, F(0)), np.maximum(dy, F(0))
    inter = wx * wy
    union = (area_a + area_b) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union
        ex0, ey0 = np.minimum(ax0, bx0), np.minimum(ay0, by0)
        ex1, ey1 = np.maximum(ax1, bx1), np.maximum(ay1, by1)
        edx, edy = ex1 - ex0, ey1 - ey0
        ewx, ewy = np.maximum(edx, F(0)), np.maximum(edy, F(0))
        earea = ewx * ewy
        giou = iou - (earea - union) / earea

        # d giou / d (inter, union, earea)
        g_union = -inter / (union * union) + F(1) / earea
        g_inter = F(1) / union - g_union
        g_earea = -union / (earea * earea)
    g_area_a = g_union
    g_wx, g_wy = g_inter * wy, g_inter * wx
    g_dx, g_dy = g_wx * (dx >= 0), g_wy * (dy >= 0)
    g_ewx, g_ewy = g_earea * ewy, g_earea * ewx
    g_edx, g_edy = g_ewx * (edx >= 0), g_ewy * (edy >= 0)
    g_ax1 = g_dx * _lo_share(ax1, bx1) + g_edx * _hi_share(ax1, bx1) + g_area_a * (ay1 - ay0)
    g_ax0 = -g_dx * _hi_share(ax0, bx0) - g_edx * _lo_share(ax0, bx0) - g_area_a * (ay1 - ay0)
    g_ay1 = g_dy * _lo_share(ay1, by1) + g_edy * _hi_share(ay1, by1) + g_area_a * (ax1 - ax0)
    g_ay0 = -g_dy * _hi_share(ay0, by0) - g_edy * _lo_share(ay0, by0) - g_area_a * (ax1 - ax0)
    grad = np.stack([g_ax0 + g_ax1, g_ay0 + g_ay1, F(0.5) * (g_ax1 - g_ax0), F(0.5) * (g_ay1 - g_ay0)], axis=1)
    return giou.astype(F), grad.astype(F)


def box_losses(pred_boxes, target_boxes, indices):
    """loss.py:97-114.  Returns l1_loss, giou_loss and their gradients w.r.t. pred_boxes [B, Q, 4]."""
    pred_boxes = np.asarray(pred_boxes, dtype=F)
    bi, si, sel, tgt = matched_pairs(pred_boxes, target_boxes, indices)
    num_boxes = max(len(bi), 1)
    diff = sel - tgt
    l1 = np.abs(diff).sum(dtype=F) / F(num_boxes)
    giou, ggrad = pair_giou(sel, tgt)
    giou_loss = (F(1) - giou).sum(dtype=F) / F(num_boxes)
    grad_l1 = np.zeros_like(pred_boxes)
    grad_giou = np.zeros_like(pred_boxes)
    # (accumulated: indices that did not come from the matcher may name a prediction in several pairs)
    np.add.at(grad_l1, (bi, si), np.sign(diff) / F(num_boxes))
    np.add.at(grad_giou, (bi, si), -ggrad / F(num_boxes))
    return F(l1), F(giou_loss), grad_l1, grad_giou


def target_classes(shape_bq, target_labels, indices, no_object):
    """loss.py:121-125."""
    bi, si = src_permutation_idx(indices)
    dense = np.full(shape_bq, no_object, dtype=np.int64)
    picked = np.concatenate([np.asarray(tl, dtype=np.int64)[j] for tl, (_, j) in zip(target_labels, indices)] +
                            [np.zeros(0, np.int64)])
    dense[bi, si] = picked
    return dense
