"""The matching path as PyTorch operators: ``torch.ops.detr_amd.match / cost / lap``.

Thin wrappers of the C ABI (include/detr_match.h) registered with ``torch.library``: tensors in,
one fresh tensor out, nothing mutated, a shape function for fake tensors -- so that
``torch.compile``, ``torch.export`` and fake-tensor tracing see the matcher
(/root/reference/models/matcher.py:38-95) as ordinary graph nodes instead of an opaque ctypes call.
The implementations exist for the ``CUDA`` dispatch key only (on ROCm that is the HIP device):
there is no CPU kernel behind these operators.

Layouts (the C ABI's):
  tgt_meta   int32 [2B + 1]   exclusive prefix sums of the targets per image (B + 1), then the
                              image processing order (B, largest number of targets first)
  match ->   int64 [2 * S * M + 1 (+ S * M)]   src (S*M) | tgt (S*M) | status word (| flat (S*M))
  cost  ->   float32 [sum_s M * stride_s]      set s: rows = packed targets, row stride
                                              stride_s = Q_s rounded up to 32
  lap   ->   int64 [2 * S * M + 1]             as match, from a cost buffer in that layout
``HungarianMatcher`` builds the reference's per-image tuples from these buffers.

Precondition on the two host integers that describe the device-side ``tgt_meta``: ``num_targets`` is its
last offset (the total number of targets, M) and ``max_targets`` is at least the largest difference of
consecutive offsets.  The kernels size their LDS arrays by ``max_targets``: an image with more targets than
that is skipped and reported through the status word (bit 16, ``RuntimeError`` from the matcher) instead of
being solved out of bounds; ``num_targets`` smaller than the real total is NOT detected (the output regions
are laid out with it) -- ``HungarianMatcher`` always passes consistent values.
"""
import ctypes
from typing import List, Sequence

import torch

from . import _lib
from ._lib import DmParams, DmSet, lib

_DEF = torch.library.Library("detr_amd", "DEF")
_COMMON = ("Tensor[] pred_logits, Tensor[] pred_boxes, Tensor tgt_labels, Tensor tgt_boxes, Tensor tgt_meta, "
           "int num_targets, int max_targets, int cls_mode, float cost_class, float cost_bbox, float cost_giou, "
           "float focal_alpha, float focal_gamma")
_DEF.define(f"match({_COMMON}, bool flat=False) -> Tensor")
_DEF.define(f"cost({_COMMON}) -> Tensor")
_DEF.define("lap(Tensor cost, int[] queries, Tensor tgt_meta, int num_targets, int max_targets, bool flat=False) -> Tensor")


def row_stride(queries: int) -> int:
    """Row stride of a set's cost region in floats: every row starts on a 128-byte line."""
    return (queries + 31) // 32 * 32


def check_predictions(pred_logits: Sequence[torch.Tensor], pred_boxes: Sequence[torch.Tensor], batch: int):
    """Validates the output sets of one call; returns (device, C, contiguous tensors, queries per set)."""
    if len(pred_logits) == 0 or len(pred_logits) != len(pred_boxes):
        raise ValueError("one pred_logits and one pred_boxes tensor per output set")
    logits0 = pred_logits[0]
    device, C = logits0.device, int(logits0.shape[-1])
    if device.type != "cuda":
        raise RuntimeError("detr_amd.HungarianMatcher runs on the GPU only (got tensors on %s)" % device)
    keep, queries = [], []
    for logits, boxes in zip(pred_logits, pred_boxes):
        if logits.dim() != 3 or boxes.dim() != 3 or boxes.shape[-1] != 4:
            raise ValueError("pred_logits must be [B, Q, C] and pred_boxes [B, Q, 4]")
        if logits.shape[0] != batch or boxes.shape[0] != batch or logits.shape[1] != boxes.shape[1]:
            raise ValueError("pred_logits / pred_boxes / targets disagree on batch size or query count")
        if int(logits.shape[-1]) != C:
            raise ValueError("all output sets of one call must share the class dimension")
        if logits.dtype != torch.float32 or boxes.dtype != torch.float32:
            raise RuntimeError(f"predictions must be float32, got {logits.dtype} / {boxes.dtype}")
        if logits.device != device or boxes.device != device:
            raise ValueError("all output sets must be on the same device")
        if logits.shape[1] == 0:
            raise ValueError("an output set has no queries")
        # dense rows per image are enough (the tail of every image's queries that the denoising models hand over,
        # pred_logits.split(sizes, dim=1)[1], is read in place); anything else is copied
        if image_stride(logits) is None:
            logits = logits.contiguous()
        if image_stride(boxes) is None or boxes.data_ptr() % 16 or image_stride(boxes) % 4:
            boxes = boxes.contiguous()
            if boxes.data_ptr() % 16:
                boxes = boxes.clone()
        keep.append((logits, boxes))
        queries.append(int(logits.shape[1]))
    return device, C, keep, queries


def image_stride(t: torch.Tensor):
    """Floats from one image to the next of a [B, Q, K] tensor whose rows are dense inside every image: 0 when the
    tensor is contiguous, None when the kernels cannot read it in place."""
    if t.is_contiguous():
        return 0
    B, Q, K = t.shape
    if t.stride(2) != 1 or t.stride(1) != K:
        return None
    st = t.stride(0)
    if B > 1 and (st < Q * K or st >= 2 ** 31):
        return None
    return int(st) if B > 1 else 0


def set_array(keep, queries: Sequence[int], num_targets: int):
    """DmSet descriptors of the output sets, their cost regions laid out one after the other.
    Returns (ctypes array, floats of cost buffer needed, per set (offset, row stride))."""
    S = len(queries)
    sets = (DmSet * S)()
    cost_off, regions = 0, []
    for s in range(S):
        q = queries[s]
        stride = row_stride(q)
        d = sets[s]
        if keep is not None:
            d.logits, d.boxes = keep[s][0].data_ptr(), keep[s][1].data_ptr()
            d.logits_stride, d.boxes_stride = image_stride(keep[s][0]), image_stride(keep[s][1])
        d.cost_off, d.out_off = cost_off, s * num_targets
        d.Q, d.q_stride = q, stride
        regions.append((cost_off, stride))
        cost_off += num_targets * stride
    return sets, cost_off, regions


def _check_targets(tgt_labels, tgt_boxes, tgt_meta, device, num_targets):
    if tgt_labels.dtype != torch.int64 or tgt_boxes.dtype != torch.float32 or tgt_meta.dtype != torch.int32:
        raise RuntimeError("targets must be int64 labels, float32 boxes and int32 offsets")
    if tgt_labels.device != device or tgt_boxes.device != device or tgt_meta.device != device:
        raise ValueError("targets and predictions must be on the same device")
    if tgt_meta.dim() != 1 or tgt_meta.shape[0] % 2 != 1:
        raise ValueError("tgt_meta must hold B + 1 offsets followed by B image indices")
    if tgt_labels.shape[0] < num_targets or tgt_boxes.shape[0] < num_targets:
        raise ValueError("fewer packed targets than num_targets")
    return (int(tgt_meta.shape[0]) - 1) // 2


def _on(device):
    return torch.cuda.device(device) if device.index not in (None, torch.cuda.current_device()) else _NO_CONTEXT


class _NoContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


def _match(pred_logits: List[torch.Tensor], pred_boxes: List[torch.Tensor], tgt_labels, tgt_boxes, tgt_meta,
           num_targets: int, max_targets: int, cls_mode: int, cost_class: float, cost_bbox: float,
           cost_giou: float, focal_alpha: float, focal_gamma: float, flat: bool = False):
    device = pred_logits[0].device
    batch = _check_targets(tgt_labels, tgt_boxes, tgt_meta, device, num_targets)
    device, C, keep, queries = check_predictions(pred_logits, pred_boxes, batch)
    S, M = len(queries), num_targets
    sets, n_cost, _ = set_array(keep, queries, M)
    workspace = torch.empty(max(n_cost, 1), dtype=torch.float32, device=device)
    out = torch.zeros((3 if flat else 2) * S * M + 1, dtype=torch.int64, device=device)
    src = out.data_ptr()
    status = src + 16 * S * M
    params = DmParams(cls_mode, cost_class, cost_bbox, cost_giou, focal_alpha, focal_gamma)
    with _on(device):
        rc = lib.dm_match_fused_f32(sets, S, batch, C, tgt_labels.data_ptr(), tgt_boxes.data_ptr(), tgt_meta.data_ptr(),
                                    tgt_meta.data_ptr() + 4 * (batch + 1), max_targets, ctypes.byref(params),
                                    workspace.data_ptr(), src, src + 8 * S * M, status + 8 if flat else None, status,
                                    torch.cuda.current_stream(device).cuda_stream)
    _lib.check(rc, "dm_match_fused_f32")
    return out


def _cost(pred_logits: List[torch.Tensor], pred_boxes: List[torch.Tensor], tgt_labels, tgt_boxes, tgt_meta,
          num_targets: int, max_targets: int, cls_mode: int, cost_class: float, cost_bbox: float,
          cost_giou: float, focal_alpha: float, focal_gamma: float):
    device = pred_logits[0].device
    batch = _check_targets(tgt_labels, tgt_boxes, tgt_meta, device, num_targets)
    device, C, keep, queries = check_predictions(pred_logits, pred_boxes, batch)
    S, M = len(queries), num_targets
    sets, n_cost, _ = set_array(keep, queries, M)
    # the padding behind column Q of a row is never written: start from zeros so that the result is defined
    cost = torch.zeros(max(n_cost, 1), dtype=torch.float32, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    params = DmParams(cls_mode, cost_class, cost_bbox, cost_giou, focal_alpha, focal_gamma)
    with _on(device):
        rc = lib.dm_cost_f32(sets, S, batch, C, tgt_labels.data_ptr(), tgt_boxes.data_ptr(), tgt_meta.data_ptr(),
                             max_targets, ctypes.byref(params), cost.data_ptr(), status.data_ptr(),
                             torch.cuda.current_stream(device).cuda_stream)
    _lib.check(rc, "dm_cost_f32")
    return cost


def _lap(cost, queries: List[int], tgt_meta, num_targets: int, max_targets: int, flat: bool = False):
    device = cost.device
    if device.type != "cuda" or cost.dtype != torch.float32 or tgt_meta.dtype != torch.int32 or tgt_meta.device != device:
        raise RuntimeError("lap needs a float32 cost buffer and int32 offsets on one GPU")
    batch = (int(tgt_meta.shape[0]) - 1) // 2
    S, M = len(queries), num_targets
    if any(q < 1 for q in queries):
        raise ValueError("an output set has no queries")
    sets, n_cost, _ = set_array(None, queries, M)
    cost = cost.contiguous()
    if cost.numel() < n_cost:
        raise ValueError("cost buffer smaller than the sets' regions")
    out = torch.zeros((3 if flat else 2) * S * M + 1, dtype=torch.int64, device=device)
    src = out.data_ptr()
    status = src + 16 * S * M
    with _on(device):
        rc = lib.dm_lap_flat_f32(sets, S, batch, cost.data_ptr(), tgt_meta.data_ptr(),
                                 tgt_meta.data_ptr() + 4 * (batch + 1), max_targets, src, src + 8 * S * M,
                                 status + 8 if flat else None, status, torch.cuda.current_stream(device).cuda_stream)
    _lib.check(rc, "dm_lap_flat_f32")
    return out


_DEF.impl("match", _match, "CUDA")
_DEF.impl("cost", _cost, "CUDA")
_DEF.impl("lap", _lap, "CUDA")


@torch.library.register_fake("detr_amd::match")
def _match_fake(pred_logits, pred_boxes, tgt_labels, tgt_boxes, tgt_meta, num_targets, max_targets, cls_mode,
                cost_class, cost_bbox, cost_giou, focal_alpha, focal_gamma, flat=False):
    S = len(pred_logits)
    return tgt_labels.new_empty(((3 if flat else 2) * S * num_targets + 1,), dtype=torch.int64)


@torch.library.register_fake("detr_amd::cost")
def _cost_fake(pred_logits, pred_boxes, tgt_labels, tgt_boxes, tgt_meta, num_targets, max_targets, cls_mode,
               cost_class, cost_bbox, cost_giou, focal_alpha, focal_gamma):
    n = 0
    for logits in pred_logits:
        n = n + num_targets * ((logits.shape[1] + 31) // 32 * 32)
    return tgt_boxes.new_empty((torch.sym_max(n, 1) if not isinstance(n, int) else max(n, 1),), dtype=torch.float32)


@torch.library.register_fake("detr_amd::lap")
def _lap_fake(cost, queries, tgt_meta, num_targets, max_targets, flat=False):
    return tgt_meta.new_empty(((3 if flat else 2) * len(queries) * num_targets + 1,), dtype=torch.int64)


match = torch.ops.detr_amd.match.default
cost = torch.ops.detr_amd.cost.default
lap = torch.ops.detr_amd.lap.default
