"""What the reference's criterion does with the matching indices, fused and device-resident
(/root/reference/models/detr/loss.py:97-125 and its twins in the other criteria):

    matched_box_losses(outputs_list, flats)   -> [(l1_loss, giou_loss)] per output set, differentiable
    matched_box_loss_total(outputs_list, flats, l1_weight, giou_weight) -> their weighted sum, one scalar
    target_class_maps(flats, no_object)       -> [dense [B, Q] int64 class map] per output set

These take the `FlatMatches` of ONE `HungarianMatcher.match_flat` call and handle all of its output
sets in one launch per operation through the C ABI; nothing synchronises with the host.

The classification loss on the dense class map (the criterion's `cls_criterion`, loss.py:127, 185-242):

    focal_loss(logits, target_cls, num_boxes, alpha, gamma)   FocalLoss / sigmoid_focal_loss, differentiable
    cross_entropy(logits, target_cls, weight)                 CrossEntropy (nn.CrossEntropyLoss(weight)), differentiable
    FocalLoss, CrossEntropy                                   modules with the reference's call convention
    check_cls_status()                                        raises for class values the reference refuses

one pass over the logits per call (loss and gradient together) instead of a dozen elementwise torch kernels
forward and as many backward.

    set_losses(matcher, outputs_list, targets, n_cls, ...)    -> [(l1_loss, giou_loss, cls_loss)] per output set

is the body of a criterion's loop over its Hungarian-matched output sets (loss.py:164-176) on top of all of the above:
one matching launch for the step, nothing synchronises.
"""
import ctypes
from typing import List, Optional, Sequence, Tuple

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


class _PairListSums(torch.autograd.Function):
    """[2] sums of |pred - target| and (1 - giou) over an explicit list of (image, query, packed target) pairs."""

    @staticmethod
    def forward(ctx, pred_boxes, tgt_boxes, pair_b, pair_q, pair_t):
        device = pred_boxes.device
        if pred_boxes.dtype != torch.float32 or pred_boxes.dim() != 3 or pred_boxes.shape[-1] != 4 or not pred_boxes.is_cuda:
            raise ValueError("pred_boxes must be a float32 [B, Q, 4] tensor on a GPU")
        if tgt_boxes.dtype != torch.float32 or tgt_boxes.dim() != 2 or tgt_boxes.shape[-1] != 4 or tgt_boxes.device != device:
            raise ValueError("tgt_boxes must be a float32 [M, 4] tensor on the device of the predictions")
        n = int(pair_b.shape[0])
        for idx in (pair_b, pair_q, pair_t):
            if idx.dtype != torch.int64 or idx.dim() != 1 or idx.shape[0] != n or idx.device != device:
                raise ValueError("the pair indices must be int64 [N] tensors on the device of the predictions")
        B, Q = int(pred_boxes.shape[0]), int(pred_boxes.shape[1])
        p = pred_boxes.detach()
        stride = ops.image_stride(p)
        if stride is None or stride % 4 or p.data_ptr() % 16:
            p = p.contiguous()
            if p.data_ptr() % 16:
                p = p.clone()
            stride = 0
        g = tgt_boxes.detach().contiguous()
        if g.data_ptr() % 16:
            g = g.clone()
        pair_b, pair_q, pair_t = pair_b.contiguous(), pair_q.contiguous(), pair_t.contiguous()
        sums = torch.empty(2, dtype=torch.float32, device=device)
        want_grad = ctx.needs_input_grad[0]
        pair_grad = torch.empty(max(n, 1), 8, dtype=torch.float32, device=device) if want_grad else None
        with torch.cuda.device(device):
            partials, status, stream = _cls_buffers(device)
            rc = lib.dm_pair_list_loss_f32(p.data_ptr(), stride, B, Q, g.data_ptr(), int(g.shape[0]), pair_b.data_ptr(),
                                           pair_q.data_ptr(), pair_t.data_ptr(), n, sums.data_ptr(),
                                           pair_grad.data_ptr() if want_grad else None, partials.data_ptr(),
                                           status.data_ptr(), stream)
        _lib.check(rc, "dm_pair_list_loss_f32")
        ctx.shape = pred_boxes.shape
        if want_grad:
            ctx.save_for_backward(pair_grad, pair_b, pair_q)
        return sums

    @staticmethod
    def backward(ctx, grad_sums):
        pair_grad, pair_b, pair_q = ctx.saved_tensors
        device = pair_grad.device
        grad = torch.zeros(ctx.shape, dtype=torch.float32, device=device)
        gs = grad_sums.contiguous().to(torch.float32)
        with torch.cuda.device(device):
            # (a query may occur in several pairs: the kernel adds with float atomics)
            rc = lib.dm_pair_list_backward_f32(pair_grad.data_ptr(), gs.data_ptr(), pair_b.data_ptr(), pair_q.data_ptr(),
                                               int(pair_b.shape[0]), int(ctx.shape[0]), int(ctx.shape[1]), grad.data_ptr(),
                                               torch.cuda.current_stream(device).cuda_stream)
        _lib.check(rc, "dm_pair_list_backward_f32")
        return grad, None, None, None, None


def pair_list_box_loss_sums(pred_boxes: torch.Tensor, tgt_boxes: torch.Tensor, pair_b: torch.Tensor, pair_q: torch.Tensor,
                            pair_t: torch.Tensor) -> torch.Tensor:
    """`[sum |pred - target|, sum (1 - giou)]` over the pairs (pred_boxes[pair_b[k], pair_q[k]], tgt_boxes[pair_t[k]]) --
    the two sums of `calc_box_loss` (/root/reference/models/detr/loss.py:105-112 before the division by num_boxes) for
    indices that did not come from the matcher: the denoising groups of DN-DETR / DINO / Co-DETR bring their own
    (/root/reference/models/dino_detr/dino_func.py:117-124), where the reference's `[N, N]` GIoU matrix has N = groups x
    targets of the whole batch.  O(N); a query may occur in several pairs (its gradients add up); an index out of range
    is reported by `check_cls_status()`.  Differentiable in `pred_boxes`."""
    return _PairListSums.apply(pred_boxes, tgt_boxes, pair_b, pair_q, pair_t)


def _takes_boxes(pred_boxes) -> bool:
    """Whether the pair kernels take these prediction boxes (anything else goes to the caller's own implementation)."""
    return pred_boxes.is_cuda and pred_boxes.dtype == torch.float32 and pred_boxes.dim() == 3 and pred_boxes.shape[-1] == 4


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


# ---- the classification loss on the dense class map ------------------------------------------------------------

_cls_partials = {}       # (device index, stream handle) -> float64 workspace of the reduction
_cls_status = {}         # device index -> int32 word the kernels OR their findings into


def _cls_buffers(device):
    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(device).cuda_stream
    partials = _cls_partials.get((index, stream))
    if partials is None:
        partials = _cls_partials[(index, stream)] = torch.empty(_lib.CLS_LOSS_PARTIALS, dtype=torch.float64, device=device)
    status = _cls_status.get(index)
    if status is None:
        status = _cls_status[index] = torch.zeros(1, dtype=torch.int32, device=device)
    return partials, status, stream


def check_cls_status(device=None) -> None:
    """Raises RuntimeError when a `focal_loss` / `cross_entropy` call on `device` (default: the current one) since the
    last check met a class value outside the range the reference accepts (its `one_hot` / `cross_entropy` raise for
    such a value at once; here the row counts as ignored and the finding waits for this call), or a
    `pair_list_box_loss_sums` call met an index outside its tensor.  Synchronises."""
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    status = _cls_status.get(index)
    if status is not None and int(status.item()):
        status.zero_()
        raise RuntimeError("class values must lie in [0, number of classes] (the last one meaning no object), "
                           "pair indices inside their tensors")


def _checked(logits, target_cls):
    if not logits.is_cuda:
        raise RuntimeError("detr_amd.losses: the classification loss runs on the GPU (there is no CPU path)")
    if logits.dtype != torch.float32:
        raise TypeError(f"logits must be float32, got {logits.dtype}")
    if target_cls.dtype != torch.int64 or target_cls.device != logits.device:
        raise TypeError("target_cls must be an int64 tensor on the device of the logits")
    if logits.dim() < 1 or tuple(logits.shape[:-1]) != tuple(target_cls.shape):
        raise ValueError(f"logits {tuple(logits.shape)} must be target_cls {tuple(target_cls.shape)} + [C]")
    x = logits.detach()
    if not x.is_contiguous() or x.data_ptr() % 16:
        x = x.contiguous()
        if x.data_ptr() % 16:
            x = x.clone()
    return x, target_cls.contiguous()


class _FocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target_cls, alpha, gamma, scale, ignore_index):
        x, cls = _checked(logits, target_cls)
        device = x.device
        grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        out = torch.empty(1, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            partials, status, stream = _cls_buffers(device)
            rc = lib.dm_focal_loss_f32(x.data_ptr(), cls.data_ptr(), cls.numel(), x.shape[-1], alpha, gamma, scale,
                                       0 if ignore_index is None else int(ignore_index), 0 if ignore_index is None else 1,
                                       out.data_ptr(), grad.data_ptr() if grad is not None else None, partials.data_ptr(),
                                       status.data_ptr(), stream)
        _lib.check(rc, "dm_focal_loss_f32")
        ctx.shape = logits.shape
        if grad is not None:
            ctx.save_for_backward(grad)
        return out.view(())

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return (grad * grad_out).view(ctx.shape), None, None, None, None, None


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target_cls, weight, ignore_index):
        x, cls = _checked(logits, target_cls)
        device = x.device
        C = x.shape[-1]
        if weight is not None:
            if weight.dtype != torch.float32 or weight.device != device or tuple(weight.shape) != (C,):
                raise ValueError("weight must be a float32 [C] tensor on the device of the logits")
            weight = weight.detach().contiguous()
        grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        out = torch.empty(2, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            partials, status, stream = _cls_buffers(device)
            rc = lib.dm_cross_entropy_f32(x.data_ptr(), cls.data_ptr(), weight.data_ptr() if weight is not None else None,
                                          cls.numel(), C, int(ignore_index), out.data_ptr(),
                                          grad.data_ptr() if grad is not None else None, partials.data_ptr(),
                                          status.data_ptr(), stream)
        _lib.check(rc, "dm_cross_entropy_f32")
        ctx.shape = logits.shape
        if grad is not None:
            ctx.save_for_backward(grad, out)
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        grad, out = ctx.saved_tensors
        return (grad * (grad_out * out[1])).view(ctx.shape), None, None, None


def focal_loss(logits: torch.Tensor, target_cls: torch.Tensor, num_boxes, alpha: float = 0.25, gamma: float = 2.0,
               ignore_index: Optional[int] = None) -> torch.Tensor:
    """`FocalLoss.forward` (/root/reference/models/detr/loss.py:198-213 with `sigmoid_focal_loss` :217-242) on
    class-last `logits [..., C]` and the class map `target_cls [...]` (C = no object): the sum of the focal terms of all
    logits divided by `num_boxes` -- the reference's `loss.mean(1).sum() / num_boxes * Q`.  `alpha < 0`: no alpha
    weighting.  `ignore_index`: rows of that class contribute nothing (with -1, the foreground selection of
    /root/reference/models/co_detr/loss.py:153-166).  A 0-dim tensor; differentiable in `logits`."""
    return _FocalLoss.apply(logits, target_cls, float(alpha), float(gamma), 1.0 / float(num_boxes), ignore_index)


def cross_entropy(logits: torch.Tensor, target_cls: torch.Tensor, weight: Optional[torch.Tensor] = None,
                  ignore_index: int = -100) -> torch.Tensor:
    """`CrossEntropy.forward` (/root/reference/models/detr/loss.py:185-195): `nn.CrossEntropyLoss(weight)` with the
    default 'mean' reduction, on class-last `logits [..., C]`.  A 0-dim tensor; differentiable in `logits`."""
    return _CrossEntropy.apply(logits, target_cls, weight, ignore_index)


def _takes(x, targets) -> bool:
    """Whether the kernels take these tensors as they are (anything else goes to the caller's own implementation)."""
    return x.is_cuda and x.dtype == torch.float32 and targets.dtype == torch.int64 and targets.device == x.device


def _class_last(inputs, targets, C, channels_first):
    """`inputs` as a [B, Q, C] tensor, or None when its layout cannot be told: the reference's criteria hand their
    `cls_criterion` [B, C, Q] (a transposed view, detr/loss.py:127) or [B, Q, C] (co_detr/loss.py:135)."""
    if channels_first is not None:
        return inputs.transpose(1, 2) if channels_first else inputs
    if inputs.dim() != 3 or targets.dim() != 2:
        return None
    B, Q = targets.shape
    if Q == C:
        return None
    if tuple(inputs.shape) == (B, Q, C):
        return inputs
    if tuple(inputs.shape) == (B, C, Q):
        return inputs.transpose(1, 2)
    return None


class FocalLoss(torch.nn.Module):
    """Takes the place of the reference's `FocalLoss(alpha, gamma, n_cls)` as a criterion's `cls_criterion`:
    `forward(inputs, targets, num_boxes)` with `inputs` [B, n_cls, Q] (/root/reference/models/detr/loss.py:208-213) or
    [B, Q, n_cls] (/root/reference/models/co_detr/loss.py:272-276) -- `channels_first` says which, None tells them
    apart by their shape -- and `targets` the [B, Q] class map.  Inputs the kernel does not take (not float32, not on
    a GPU, a layout that cannot be told) go to `fallback` when one is given."""

    def __init__(self, alpha: float, gamma: float, n_cls: int, channels_first: Optional[bool] = None, fallback=None):
        super().__init__()
        self.alpha, self.gamma, self.n_cls, self.channels_first = alpha, gamma, n_cls, channels_first
        self.fallback = fallback

    def forward(self, inputs, targets, num_boxes):
        x = _class_last(inputs, targets, self.n_cls, self.channels_first)
        if x is None or x.shape[-1] != self.n_cls or not _takes(x, targets):
            if self.fallback is None:
                raise RuntimeError("detr_amd.losses.FocalLoss: float32 GPU logits of [B, Q, n_cls] or [B, n_cls, Q] expected")
            return self.fallback(inputs, targets, num_boxes)
        return focal_loss(x, targets, num_boxes, self.alpha, self.gamma)


class CrossEntropy(torch.nn.Module):
    """Takes the place of the reference's `CrossEntropy(weight)` (/root/reference/models/detr/loss.py:185-195, input
    [B, C, Q]; /root/reference/models/co_detr/loss.py:249-259, input [B, Q, C])."""

    def __init__(self, weight: Optional[torch.Tensor] = None, channels_first: Optional[bool] = None,
                 ignore_index: int = -100, fallback=None):
        super().__init__()
        self.register_buffer("weight", None if weight is None else weight.detach().clone().float())
        self.channels_first, self.ignore_index, self.fallback = channels_first, ignore_index, fallback

    def forward(self, inputs, targets, num_boxes=None):
        C = self.weight.shape[0] if self.weight is not None else \
            (inputs.shape[1] if self.channels_first else inputs.shape[-1] if self.channels_first is False else None)
        x = _class_last(inputs, targets, C, self.channels_first) if C is not None else None
        if x is None or not _takes(x, targets) or (self.weight is not None and self.weight.device != x.device):
            if self.fallback is None:
                raise RuntimeError("detr_amd.losses.CrossEntropy: float32 GPU logits of [B, Q, C] or [B, C, Q] expected")
            return self.fallback(inputs, targets, num_boxes)
        return cross_entropy(x, targets, self.weight, self.ignore_index)


def set_losses(matcher, outputs_list, targets, n_cls: int, cls_loss: str = "focal", focal_alpha: float = 0.25,
               focal_gamma: float = 2.0, class_weight: Optional[torch.Tensor] = None
               ) -> List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """What the loop of the reference's `Criterion.forward` computes for its Hungarian-matched output sets
    (/root/reference/models/detr/loss.py:164-176: matcher, `calc_box_loss`, `calc_cls_loss`), for all of
    `outputs_list` (dicts with 'pred_logits' [B, Q, C] and 'pred_boxes' [B, Q, 4]) at once: `(l1_loss, giou_loss,
    cls_loss)` per set, the same values and gradients, device scalars.  One matching launch for the step, one launch for
    the pair losses of all sets, one for their class maps, one pass over each set's logits; no host synchronisation
    (`matcher.check_status()` and `check_cls_status()` raise what the reference would have raised for bad inputs).
    `cls_loss` 'ce': weighted cross entropy over n_cls + 1 classes with `class_weight` (the reference's ones with
    `noobj_weight` last, :59-62); anything else: the focal loss over n_cls logits."""
    outputs_list = list(outputs_list)
    flats = matcher.match_flat(outputs_list, targets)
    if not flats:
        return []
    box = matched_box_losses(outputs_list, flats)
    maps = target_class_maps(flats, n_cls)
    result = []
    for o, f, (l1, giou), cls_map in zip(outputs_list, flats, box, maps):
        logits = o["pred_logits"]
        if cls_loss == "ce":
            c = cross_entropy(logits, cls_map, class_weight)
        else:
            c = focal_loss(logits, cls_map, max(f.num_matched, 1), focal_alpha, focal_gamma)
        result.append((l1, giou, c))
    return result


class _ClsLossOfSets(torch.autograd.Function):
    """[S] classification losses of S output sets as ONE autograd node (the launches are still one pass per set)."""

    @staticmethod
    def forward(ctx, kind, maps, scales, alpha, gamma, weight, *logits):
        S = len(logits)
        checked = [_checked(x, m) for x, m in zip(logits, maps)]
        device = checked[0][0].device
        out = torch.empty(S, 2, dtype=torch.float32, device=device)
        grads = [torch.empty_like(x) if ctx.needs_input_grad[6 + i] else None for i, (x, _) in enumerate(checked)]
        if weight is not None:
            weight = weight.detach().contiguous()
        with torch.cuda.device(device):
            partials, status, stream = _cls_buffers(device)
            for i, ((x, cls), g) in enumerate(zip(checked, grads)):
                g_ptr = g.data_ptr() if g is not None else None
                if kind == "ce":
                    rc = lib.dm_cross_entropy_f32(x.data_ptr(), cls.data_ptr(), weight.data_ptr() if weight is not None else None,
                                                  cls.numel(), x.shape[-1], -100, out.data_ptr() + 8 * i, g_ptr,
                                                  partials.data_ptr(), status.data_ptr(), stream)
                else:
                    rc = lib.dm_focal_loss_f32(x.data_ptr(), cls.data_ptr(), cls.numel(), x.shape[-1], alpha, gamma, scales[i],
                                               0, 0, out.data_ptr() + 8 * i, g_ptr, partials.data_ptr(), status.data_ptr(),
                                               stream)
                _lib.check(rc, "dm_cross_entropy_f32" if kind == "ce" else "dm_focal_loss_f32")
        ctx.kind, ctx.present = kind, [g is not None for g in grads]
        ctx.save_for_backward(out, *[g for g in grads if g is not None])
        return out[:, 0]

    @staticmethod
    def backward(ctx, grad_losses):
        out, *bufs = ctx.saved_tensors
        factors = grad_losses * out[:, 1] if ctx.kind == "ce" else grad_losses
        picked = [f for f, present in zip(factors.unbind(0), ctx.present) if present]
        scaled = iter(torch._foreach_mul(bufs, picked) if bufs else [])
        return (None,) * 6 + tuple(next(scaled) if present else None for present in ctx.present)


def set_loss_total(matcher, outputs_list, targets, n_cls: int, cls_loss: str = "focal", l1_weight: float = 1.0,
                   giou_weight: float = 1.0, cls_weight: float = 1.0, focal_alpha: float = 0.25, focal_gamma: float = 2.0,
                   class_weight: Optional[torch.Tensor] = None, reduction: str = "mean") -> torch.Tensor:
    """`l1_weight * l1 + giou_weight * giou + cls_weight * cls` over the Hungarian-matched output sets of ONE output group
    as one scalar -- with reduction='mean' what the reference's `calc_total_loss` makes of the group's per-layer losses
    (/root/reference/models/detr/loss.py:135-158: the mean of each kind over the layers, weighted and summed) -- with
    three autograd nodes for the whole group instead of some twenty per set: the values of `set_losses`, and a step that
    is no longer bound by the host's walk through tiny tensor operations."""
    if reduction not in ("mean", "sum"):
        raise ValueError("reduction must be 'mean' or 'sum'")
    outputs_list = list(outputs_list)
    flats = matcher.match_flat(outputs_list, targets)
    if not flats:
        raise ValueError("no output sets")
    S = len(flats)
    share = 1.0 / S if reduction == "mean" else 1.0
    box = matched_box_loss_total(outputs_list, flats, l1_weight, giou_weight, reduction=reduction)
    maps = target_class_maps(flats, n_cls)
    scales = [1.0 / float(max(f.num_matched, 1)) for f in flats]
    cls = _ClsLossOfSets.apply("ce" if cls_loss == "ce" else "focal", maps, scales, float(focal_alpha), float(focal_gamma),
                               class_weight, *[o["pred_logits"] for o in outputs_list])
    return box + cls.sum() * (share * cls_weight)
