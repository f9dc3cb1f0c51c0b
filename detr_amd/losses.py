"""What the reference's criterion does with the matching indices, fused and device-resident
(/root/reference/models/detr/loss.py:97-125 and its twins in the other criteria):

    matched_box_losses(outputs_list, flats)   -> [(l1_loss, giou_loss)] per output set, differentiable
    matched_box_loss_total(outputs_list, flats, l1_weight, giou_weight) -> their weighted sum, one scalar
    target_class_maps(flats, no_object)       -> [dense [B, Q] int64 class map] per output set

Both take the `FlatMatches` of ONE `HungarianMatcher.match_flat` call and handle all of its output
sets in one launch per operation through the C ABI; nothing synchronises with the host.
"""
import ctypes
from typing import List, Sequence, Tuple

import torch

from . import _lib, ops
from ._lib import DmIndexSet, lib
from .matcher import FlatMatches


def _same_launch(flats: Sequence[FlatMatches]):
    if not flats:
        raise ValueError("no output sets")
    out, S = flats[0].raw
    for f in flats:
        if f.raw is None or f.raw[0] is not out:
            raise ValueError("all FlatMatches must come from the same match_flat call")
    return out, S, flats[0].packed


def _index_sets(flats, M, pred=None, grad=None, cls=None):
    sets = (DmIndexSet * len(flats))()
    for i, f in enumerate(flats):
        sets[i].pred_boxes = pred[i].data_ptr() if pred is not None else None
        sets[i].boxes_stride = ops.image_stride(pred[i]) if pred is not None else 0
        sets[i].grad_boxes = grad[i].data_ptr() if grad is not None else None
        sets[i].target_cls = cls[i].data_ptr() if cls is not None else None
        sets[i].out_off, sets[i].Q = f.set_index * M, f.queries
    return sets


class _PairLossSums(torch.autograd.Function):
    """[len(flats), 2] sums of |pred - target| and (1 - giou) over each set's matched pairs."""

    @staticmethod
    def forward(ctx, flats, *pred_boxes):
        out, S, packed = _same_launch(flats)
        M, B = packed.total, packed.batch
        device = out.device
        boxes = []
        for f, p in zip(flats, pred_boxes):
            if p.dtype != torch.float32 or p.dim() != 3 or tuple(p.shape) != (B, f.queries, 4) or p.device != device:
                raise ValueError("pred_boxes must be float32 [B, Q, 4] on the device of the indices")
            p = p.detach()
            # rows dense inside every image are read in place (the tail views of the denoising models); else a copy
            stride = ops.image_stride(p)
            if stride is None or stride % 4 or p.data_ptr() % 16:
                p = p.contiguous()
                if p.data_ptr() % 16:
                    p = p.clone()
            boxes.append(p)
        n = len(flats)
        sums = torch.empty(n, 2, dtype=torch.float32, device=device)
        pair_grad = torch.empty(n, max(M, 1), 8, dtype=torch.float32, device=device)
        sets = _index_sets(flats, M, pred=boxes)
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            rc = lib.dm_pair_loss_f32(sets, n, B, M, packed.boxes.data_ptr(), packed.offsets_ptr, out.data_ptr(),
                                      out.data_ptr() + 8 * S * M, sums.data_ptr(), pair_grad.data_ptr(), stream)
        _lib.check(rc, "dm_pair_loss_f32")
        ctx.flats, ctx.raw = flats, (out, S, packed)
        ctx.save_for_backward(pair_grad)
        return sums

    @staticmethod
    def backward(ctx, grad_sums):
        (pair_grad,) = ctx.saved_tensors
        flats, (out, S, packed) = ctx.flats, ctx.raw
        M, B = packed.total, packed.batch
        device = out.device
        grads = [torch.empty(B, f.queries, 4, dtype=torch.float32, device=device) for f in flats]
        sets = _index_sets(flats, M, grad=grads)
        gs = grad_sums.contiguous().to(torch.float32)
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            rc = lib.dm_pair_loss_backward_f32(sets, len(flats), B, M, packed.offsets_ptr, out.data_ptr(),
                                               out.data_ptr() + 8 * S * M, pair_grad.data_ptr(), gs.data_ptr(), stream)
        _lib.check(rc, "dm_pair_loss_backward_f32")
        return (None, *grads)


def matched_box_losses(outputs_list, flats: Sequence[FlatMatches]) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """`calc_box_loss` (loss.py:97-114) of every output set: `(l1_loss, giou_loss)`, each divided by
    the set's own `max(number of matched pairs, 1)` like the reference (:100)."""
    sums = _PairLossSums.apply(list(flats), *[o["pred_boxes"] for o in outputs_list])
    result = []
    for i, f in enumerate(flats):
        n = float(max(f.num_matched, 1))
        result.append((sums[i, 0] / n, sums[i, 1] / n))
    return result


def matched_box_loss_total(outputs_list, flats: Sequence[FlatMatches], l1_weight: float = 1.0,
                           giou_weight: float = 1.0, reduction: str = "mean") -> torch.Tensor:
    """`l1_weight * l1_loss + giou_weight * giou_loss` over the output sets as one scalar with two
    autograd nodes instead of a dozen per set.

    reduction='mean' (default): the MEAN over the sets -- the box part of the reference's total loss when
    `outputs_list` holds the sets of ONE output group (`calc_total_loss` takes the mean of the per-layer
    losses of each key before weighting them, /root/reference/models/detr/loss.py:150-154; call it once per
    key for `'model'` and `'first_stage'`).  reduction='sum': the plain sum over the sets, len(flats) times
    the mean."""
    if reduction not in ("mean", "sum"):
        raise ValueError("reduction must be 'mean' or 'sum'")
    sums = _PairLossSums.apply(list(flats), *[o["pred_boxes"] for o in outputs_list])
    share = 1.0 / max(len(flats), 1) if reduction == "mean" else 1.0
    scale = torch.tensor([[share * l1_weight / max(f.num_matched, 1), share * giou_weight / max(f.num_matched, 1)]
                          for f in flats], dtype=torch.float32).to(sums.device, non_blocking=True)
    return (sums * scale).sum()


@torch.no_grad()
def target_class_maps(flats: Sequence[FlatMatches], no_object: int) -> List[torch.Tensor]:
    """Dense `[B, Q]` int64 class map of every output set (loss.py:121-125), one launch."""
    out, S, packed = _same_launch(flats)
    M, B = packed.total, packed.batch
    maps = [torch.empty(B, f.queries, dtype=torch.int64, device=out.device) for f in flats]
    sets = _index_sets(flats, M, cls=maps)
    stream = torch.cuda.current_stream(out.device).cuda_stream
    with torch.cuda.device(out.device):
        rc = lib.dm_target_classes_i64(sets, len(flats), B, M, packed.labels.data_ptr(), packed.offsets_ptr,
                                       out.data_ptr(), out.data_ptr() + 8 * S * M, int(no_object), stream)
    _lib.check(rc, "dm_target_classes_i64")
    return maps
