"""Co-DETR auxiliary label assigners on the GPU: drop-ins for `ATSS` and `FasterRCNN` of the
reference (/root/reference/models/co_detr/auxiliary_head.py:17-60, 63-96).

Same constructors, same call `head(outputs, targets)`, same return values (CPU int64 index
tensors per image).  `assign_many(list_of_outputs, targets)` handles all output sets of a step
(one per decoder layer, co_detr/loss.py:225-237) with one launch per kernel and one
device-to-host copy.  Two things the reference leaves to implementation detail are fixed here:
ATSS candidates at exactly equal centre distance are ordered by prediction index, and the
Faster-RCNN positives of one target are listed by ascending prediction index.
There is no CPU fallback.
"""
import ctypes
from typing import Dict, List, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib, ops
from ._lib import DmIndexSet, lib
from .matcher import PackedTargets

Pairs = List[Tuple[torch.Tensor, torch.Tensor]]


def _prepare(outputs_list, targets):
    if not outputs_list:
        raise ValueError("no output sets")
    device = outputs_list[0]["pred_boxes"].device
    if device.type != "cuda":
        raise RuntimeError("detr_amd auxiliary assigners run on the GPU only (got tensors on %s)" % device)
    if isinstance(targets, PackedTargets):
        packed = targets
    else:
        # the reference moves the target boxes to the predictions' device itself (auxiliary_head.py:31)
        if any(t["boxes"].device != device for t in targets):
            packed = PackedTargets.from_host([{k: v.cpu() for k, v in t.items()} for t in targets], device)
        else:
            packed = PackedTargets(targets, device)
    sets = (DmIndexSet * len(outputs_list))()
    keep = []
    for i, out in enumerate(outputs_list):
        boxes = out["pred_boxes"]
        if boxes.dim() != 3 or boxes.shape[-1] != 4 or boxes.shape[0] != packed.batch:
            raise ValueError("pred_boxes must be [B, Q, 4] with B = len(targets)")
        if boxes.dtype != torch.float32:
            raise RuntimeError(f"pred_boxes must be float32, got {boxes.dtype}")
        boxes = boxes.detach()
        stride = ops.image_stride(boxes)          # rows dense inside every image are read in place; else a copy
        if stride is None or stride % 4 or boxes.data_ptr() % 16:
            boxes = boxes.contiguous()
            if boxes.data_ptr() % 16:
                boxes = boxes.clone()
        keep.append(boxes)
        sets[i].pred_boxes, sets[i].Q = boxes.data_ptr(), int(boxes.shape[1])
        sets[i].boxes_stride = ops.image_stride(boxes)
    return device, packed, sets, keep


def _split_per_image(src, tgt, s_idx, t_idx, packed, S) -> List[Pairs]:
    """(src, tgt) pairs sorted by (set, target slot) -> list over sets of lists over images of tuples."""
    B = packed.batch
    counts = np.bincount(s_idx * B + packed.image_index[t_idx], minlength=S * B).tolist()
    pairs = list(zip(src.split(counts), tgt.split(counts)))
    return [pairs[s * B:(s + 1) * B] for s in range(S)]


def _to_host(dev: torch.Tensor) -> np.ndarray:
    host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
    host.copy_(dev, non_blocking=True)
    torch.cuda.current_stream(dev.device).synchronize()
    return host.numpy()


class ATSS(torch.nn.Module):
    """Adaptive training sample selection with single-scale outputs (auxiliary_head.py:17-60)."""

    def __init__(self, k):
        super().__init__()
        self.k = k

    @torch.no_grad()
    def forward(self, outputs: Dict[str, torch.Tensor], targets) -> Pairs:
        return self.assign_many([outputs], targets)[0]

    @torch.no_grad()
    def assign_many(self, outputs_list: Sequence[Dict[str, torch.Tensor]],
                    targets: Union[PackedTargets, Sequence[Dict[str, torch.Tensor]]]) -> List[Pairs]:
        device, packed, sets, keep = _prepare(outputs_list, targets)
        S, B, M, k = len(outputs_list), packed.batch, packed.total, int(self.k)
        empty = torch.zeros(0, dtype=torch.int64)
        if M == 0:
            return [[(empty, empty) for _ in range(B)] for _ in range(S)]
        if any(k > b.shape[1] for b in keep):
            raise RuntimeError("selected index k out of range")      # what torch.topk raises (:38)
        # one int64 buffer: [candidates int32 S*M*k (padded to 8 bytes) | positive masks u64 S*M]
        n_cand = (S * M * k + 1) // 2
        buf = torch.empty(n_cand + S * M, dtype=torch.int64, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            rc = lib.dm_atss_f32(sets, S, B, M, k, packed.boxes.data_ptr(), packed.offsets_ptr, buf.data_ptr(),
                                 buf.data_ptr() + 8 * n_cand, stream)
        _lib.check(rc, "dm_atss_f32")
        host = _to_host(buf)
        cand = host[:n_cand].view(np.int32)[:S * M * k].reshape(S, M, k)
        mask = host[n_cand:].view(np.uint64).reshape(S, M)
        bits = ((mask[:, :, None] >> np.arange(k, dtype=np.uint64)) & np.uint64(1)).astype(bool)     # [S, M, k]
        s_idx, t_idx, j_idx = np.nonzero(bits)           # set-major, then target, then candidate rank
        src = torch.from_numpy(cand[s_idx, t_idx, j_idx].astype(np.int64))
        tgt = torch.from_numpy(packed.local_index[t_idx])
        return _split_per_image(src, tgt, s_idx, t_idx, packed, S)


class FasterRCNN(torch.nn.Module):
    """IoU-threshold assignment (auxiliary_head.py:63-96): returns (pos_indices, neg_indices)."""

    def __init__(self, pos_threshold=0.7, neg_threshold=0.3):
        super().__init__()
        self.pos_threshold = pos_threshold
        self.neg_threshold = neg_threshold

    @torch.no_grad()
    def forward(self, outputs: Dict[str, torch.Tensor], targets):
        return self.assign_many([outputs], targets)[0]

    @torch.no_grad()
    def assign_many(self, outputs_list, targets):
        device, packed, sets, keep = _prepare(outputs_list, targets)
        S, B, M = len(outputs_list), packed.batch, packed.total
        queries = [int(b.shape[1]) for b in keep]
        W = (max(queries) + 63) // 64
        if B == 0:
            return [([], []) for _ in range(S)]
        # one int64 buffer: [pos_bits u64 S*M*W | neg_bits u64 S*B*W | best int32 S*M]
        n_pos, n_neg = S * M * W, S * B * W
        buf = torch.empty(n_pos + n_neg + (S * M + 1) // 2, dtype=torch.int64, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            rc = lib.dm_frcnn_f32(sets, S, B, M, W, float(self.pos_threshold), float(self.neg_threshold),
                                  packed.boxes.data_ptr(), packed.offsets_ptr, buf.data_ptr(),
                                  buf.data_ptr() + 8 * (n_pos + n_neg), buf.data_ptr() + 8 * n_pos, stream)
        _lib.check(rc, "dm_frcnn_f32")
        host = _to_host(buf)
        words = host[:n_pos].view(np.uint64).reshape(S, M, W)
        neg = np.unpackbits(host[n_pos:n_pos + n_neg].view(np.uint8), bitorder="little").view(np.bool_).reshape(S, B, W * 64)
        best = host[n_pos + n_neg:].view(np.int32)[:S * M].reshape(S, M)
        # positives are sparse: expand only the non-zero words, add the best prediction of every
        # target (:90), then sort by (set, target, prediction) and drop the duplicates
        s_i, t_i, w_i = np.nonzero(words)
        r, bit = np.nonzero((words[s_i, t_i, w_i][:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1))
        slot = np.concatenate([s_i[r] * M + t_i[r], np.arange(S * M, dtype=np.int64)])
        q = np.concatenate([w_i[r] * 64 + bit, best.reshape(-1).astype(np.int64)])
        key = np.unique(slot * (W * 64) + q)
        slot, q = key // (W * 64), key % (W * 64)
        s_idx, t_idx = slot // max(M, 1), slot % max(M, 1)
        pos = _split_per_image(torch.from_numpy(q), torch.from_numpy(packed.local_index[t_idx]), s_idx, t_idx,
                               packed, S)
        result = []
        for s in range(S):
            # one scan of the set's [B, Q] table (row-major: ascending prediction index inside every image), split by image
            b_idx, q_idx = np.nonzero(neg[s, :, :queries[s]])
            neg_indices = list(torch.from_numpy(q_idx).split(np.bincount(b_idx, minlength=B).tolist()))
            result.append((pos[s], neg_indices))
        return result

    @torch.no_grad()
    def dense_class_targets(self, outputs_list, targets, n_cls: int) -> Tuple[List[torch.Tensor], torch.Tensor]:
        """`(maps, num_pos)`: the `[B, Q]` int64 class targets of every output set as
        `calc_cls_loss_with_ignore` builds them from `forward`'s lists (co_detr/loss.py:153-159) --
        `n_cls` for negatives, the label of the last positive target otherwise, -1 for ignored
        predictions -- and the int32 `[S]` number of positive pairs per set (loss.py:151).
        Everything stays on the device: no index lists, no host copy, no synchronisation."""
        device, packed, sets, keep = _prepare(outputs_list, targets)
        S, B, M = len(outputs_list), packed.batch, packed.total
        queries = [int(b.shape[1]) for b in keep]
        W = (max(queries) + 63) // 64
        maps = [torch.empty(B, q, dtype=torch.int64, device=device) for q in queries]
        for i, m_ in enumerate(maps):
            sets[i].target_cls = m_.data_ptr()
        # one int64 buffer: [pos_bits u64 S*M*W | neg_bits u64 S*B*W | best int32 S*M (padded) | num_pos int32 S]
        n_pos, n_neg, n_best = S * M * W, S * B * W, (S * M + 1) // 2
        buf = torch.empty(n_pos + n_neg + n_best + (S + 1) // 2, dtype=torch.int64, device=device)
        best_ptr = buf.data_ptr() + 8 * (n_pos + n_neg)
        num_pos = buf[n_pos + n_neg + n_best:].view(torch.int32)[:S]
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            rc = lib.dm_frcnn_f32(sets, S, B, M, W, float(self.pos_threshold), float(self.neg_threshold),
                                  packed.boxes.data_ptr(), packed.offsets_ptr, buf.data_ptr(), best_ptr,
                                  buf.data_ptr() + 8 * n_pos, stream)
            _lib.check(rc, "dm_frcnn_f32")
            rc = lib.dm_frcnn_dense_i64(sets, S, B, M, float(self.pos_threshold), float(self.neg_threshold),
                                        packed.boxes.data_ptr(), packed.labels.data_ptr(), packed.offsets_ptr,
                                        best_ptr, int(n_cls), num_pos.data_ptr(), stream)
        _lib.check(rc, "dm_frcnn_dense_i64")
        return maps, num_pos


def get_aux_heads(args):
    """As the reference's factory (auxiliary_head.py:7-15)."""
    heads = {}
    for key in args.aux_heads:
        if key == "atss":
            heads[key] = ATSS(args.atss_k)
        if key == "faster_rcnn":
            heads[key] = FasterRCNN()
    return heads
