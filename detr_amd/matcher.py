"""Drop-in HungarianMatcher backed by the gfx950 kernels in libdetr_match.so.

Same constructor, call convention and return value as the reference module
(/root/reference/models/matcher.py:13-95): ``matcher(outputs, targets)`` returns a
list of ``(index_i, index_j)`` CPU int64 tensors, one per image, with
``len == min(num_queries, num_targets)``.

On top of that, ``match_many`` matches every Hungarian-matched output set of a
training step (the decoder layers the criteria loop over, e.g.
/root/reference/models/dino_detr/loss.py:165-199, plus the two-stage encoder head)
in one call: with up to four logits per query ONE kernel launch (every assignment
workgroup computes its own cost block in LDS and stores its indices straight into the
pinned host tensor the returned tuples are views of), with more classes one cost launch
and one assignment launch; completion is one event either way.  ``prefetch`` does that
ahead of a loop of per-set calls (see detr_amd/criterion.py).

There is no CPU path: inputs must live on a GPU and the HIP library must be built.
"""
from itertools import accumulate
from typing import Dict, List, Optional, Sequence, Tuple, Union

import ctypes
import functools

import numpy as np
import torch

from . import _lib, ops
from ._lib import DmParams, DmSet, lib

Indices = List[Tuple[torch.Tensor, torch.Tensor]]

_split_with_sizes = torch.ops.aten.unsafe_split_with_sizes.default
_TARGETS_ENTRY = ctypes.cast(lib.dm_match_targets_f32, ctypes.c_void_p).value
# (device index) -> the current stream's handle, without building a Stream object where torch offers that
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda index: torch.cuda.current_stream(index).cuda_stream)

try:        # bulk creation of the per-image index views (csrc/index_views.cpp, built by build_ext.py)
    from . import _index_views
except ImportError:      # not built: the views are made one by one in Python, same result
    _index_views = None
if _index_views is not None and getattr(_index_views, "built_for_torch", None) != torch.__version__:
    # the helper re-points live tensor objects through torch internals that are no stable interface: it is only
    # used under the torch version it was compiled against (build_ext.py rebuilds it); the Python paths give the
    # same results
    import warnings

    warnings.warn("detr_amd._index_views was built for torch %s, this is torch %s: not used (run build_ext.py)"
                  % (getattr(_index_views, "built_for_torch", "?"), torch.__version__))
    _index_views = None


def _helper():
    """The host helper module -- None while torch.compile traces the caller: Dynamo cannot look into a C
    function and would break the graph (with a warning) at each of its calls, so tracing takes the tensor-only
    Python paths, which give the same results."""
    return None if torch.compiler.is_compiling() else _index_views


class PackedTargets:
    """The targets of one batch, packed once and shared by every output set of the step.

    labels [M] int64, boxes [M, 4] float32 (cxcywh), both on the device; per-image
    offsets and the largest-first processing order travel in one small upload.
    """

    def __init__(self, targets: Sequence[Dict[str, torch.Tensor]], device: torch.device, defer: bool = False):
        """defer=True (what the matcher uses for targets it packs itself): the packed arrays are only filled when
        somebody needs them -- a step that runs as one fused kernel reads the targets where they are."""
        self.device = device
        self._source = self._scanned = None
        helper = _helper()
        if helper is not None and device.type == "cuda":
            index = device.index if device.index is not None else torch.cuda.current_device()
            scan = helper.scan_targets(targets, index)
            if scan is not None:          # the usual case: int64 labels, float32 [T, 4] boxes, contiguous, on `device`
                self._pack_scanned(scan, device, index, defer)
                return
        sizes, label_list, box_list = [], [], []
        for t in targets:                       # one pass over the dicts
            b = t["boxes"]
            sizes.append(b.shape[0])
            label_list.append(t["labels"])
            box_list.append(b)
        # what `same_targets` compares a later call's targets with: the tensor objects themselves and
        # their version counters (every in-place change of a tensor bumps its counter)
        self._source = (label_list, box_list, [x._version for x in label_list], [x._version for x in box_list])
        self.sizes = sizes
        self.batch = B = len(sizes)
        self.total = sum(sizes)
        self.max_t = max(sizes) if sizes else 0
        if self.total > 0:
            labels = torch.cat(label_list)
            boxes = torch.cat(box_list)
            if boxes.device != device or labels.device != device:
                raise ValueError("targets and predictions must be on the same device")
            if boxes.dtype != torch.float32:
                raise RuntimeError(f"target boxes must be float32, got {boxes.dtype}")
            if labels.shape[0] != self.total:
                raise ValueError("a target's labels and boxes differ in length")
            self.labels = labels if labels.dtype == torch.int64 else labels.to(torch.int64)
            self.boxes = boxes if boxes.dim() == 2 and boxes.shape[1] == 4 else boxes.reshape(-1, 4)
        else:
            if any(int(t.shape[0]) for t in label_list):
                raise ValueError("a target's labels and boxes differ in length")
            self.labels = torch.zeros(1, dtype=torch.int64, device=device)
            self.boxes = torch.zeros(1, 4, dtype=torch.float32, device=device)
        # [offsets (B+1) | order (B)]: heavy images first, the assignment kernel's tail is set by the largest T
        offsets = [0]
        offsets.extend(accumulate(sizes))
        self.offsets_list = offsets
        order = sorted(range(B), key=sizes.__getitem__, reverse=True) if self.max_t > 0 else list(range(B))
        staged = torch.tensor(offsets + order, dtype=torch.int32, pin_memory=True)
        self.meta = staged.to(device, non_blocking=True)

    def _pack_scanned(self, scan, device, index, defer=False):
        """Packing through the library's pack kernel (dm_pack_targets): one launch reads every image's tensors
        through the pointers the scan collected; offsets and order travel as kernel arguments (no upload)."""
        self.sizes, self.total, self.max_t, pointers, tensors, versions, self.offsets_list, meta = scan
        self._scanned = (tensors, versions)       # keeps the tensors (and so the pointers) alive
        self.batch = B = len(self.sizes)
        self._unpacked = (pointers, meta, index)      # what dm_pack_targets needs; None once the buffer is filled
        self._pointers = None
        if not defer:
            self._ensure_packed()

    def _pack_buffer(self):
        """One device buffer [boxes f32 4M | labels int64 M | meta int32 2B+1], every part aligned for its type;
        allocated when the packed form is first needed.  Returns the addresses (labels, boxes, offsets, order)."""
        if self._pointers is None:
            B, M = self.batch, max(self.total, 1)
            off_labels = 16 * M
            off_meta = off_labels + 8 * M
            buf = torch.empty(off_meta + 4 * (2 * B + 1), dtype=torch.uint8, device=self.device)
            # (the tensors `labels`, `boxes`, `meta` are views of this buffer, made when somebody asks for them:
            # the matcher itself only needs the addresses)
            self._buf, self._layout = buf, (M, off_labels, off_meta)
            base = buf.data_ptr()
            self._pointers = (base + off_labels, base, base + off_meta, base + off_meta + 4 * (B + 1))
        return self._pointers

    def _ensure_packed(self):
        pending = self.__dict__.get("_unpacked")
        if pending is None:
            return
        pointers, meta, index = pending
        labels_ptr, boxes_ptr, meta_ptr, _ = self._pack_buffer()
        self._unpacked = None
        M, off_labels, off_meta = self._layout
        if self.total == 0:
            self._buf[:off_meta].zero_()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if index == torch.cuda.current_device():
            rc = lib.dm_pack_targets(pointers, meta, self.batch, labels_ptr, boxes_ptr, meta_ptr, stream)
        else:
            with torch.cuda.device(self.device):
                rc = lib.dm_pack_targets(pointers, meta, self.batch, labels_ptr, boxes_ptr, meta_ptr, stream)
        _lib.check(rc, "dm_pack_targets")

    def __getattr__(self, name):
        # only reached for attributes that are not set: the lazily made views of the pack kernel's buffer
        if name in ("labels", "boxes", "meta") and "_scanned" in self.__dict__ and self._scanned is not None:
            self._ensure_packed()
            M, off_labels, off_meta = self._layout
            buf = self._buf
            self.boxes = buf[:off_labels].view(torch.float32).view(M, 4)
            self.labels = buf[off_labels:off_meta].view(torch.int64)
            self.meta = buf[off_meta:].view(torch.int32)
            return self.__dict__[name]
        raise AttributeError(name)

    def device_pointers(self):
        """Addresses of (labels, boxes, offsets, order) on the device."""
        self._ensure_packed()
        ptrs = self.__dict__.get("_pointers")
        if ptrs is None:
            meta = self.meta.data_ptr()
            ptrs = self._pointers = (self.labels.data_ptr(), self.boxes.data_ptr(), meta, meta + 4 * (self.batch + 1))
        return ptrs

    def same_targets(self, targets) -> bool:
        """True when `targets` are the very tensors this object was packed from, unchanged since."""
        scanned = getattr(self, "_scanned", None)
        if scanned is not None:
            helper = _helper()
            return helper is not None and helper.same_targets(targets, scanned[0], scanned[1])
        src = getattr(self, "_source", None)
        if src is None or len(targets) != self.batch:
            return False
        labels, boxes, label_versions, box_versions = src
        i = 0
        for t in targets:
            l, b = t["labels"], t["boxes"]
            if l is not labels[i] or b is not boxes[i] or l._version != label_versions[i] or b._version != box_versions[i]:
                return False
            i += 1
        return True

    @property
    def offsets_host(self) -> np.ndarray:
        """host int32 [B + 1]: exclusive prefix sums of the targets per image"""
        if getattr(self, "_offsets_host", None) is None:
            self._offsets_host = np.asarray(self.offsets_list, dtype=np.int32)
        return self._offsets_host

    @classmethod
    def from_host(cls, targets: Sequence[Dict[str, torch.Tensor]], device: torch.device) -> "PackedTargets":
        """Pack targets that are still on the CPU (what the reference's collate function yields,
        /root/reference/datasets/transforms.py:104-108) and move them with ONE upload instead
        of the 2B per-tensor copies of /root/reference/train.py:71."""
        self = cls.__new__(cls)
        self.sizes = [int(t["boxes"].shape[0]) for t in targets]
        self.batch = len(self.sizes)
        self.total = sum(self.sizes)
        self.max_t = max(self.sizes) if self.sizes else 0
        self.device = device
        B, M = self.batch, max(self.total, 1)
        n_meta = 2 * B + 1
        # one pinned byte buffer: [boxes f32 4M | labels int64 M | meta int32 2B+1]; every part starts
        # on the alignment its element type (and the kernels' 16-byte box loads) needs
        off_labels = 16 * M
        off_meta = off_labels + 8 * M
        staged = torch.empty(off_meta + 4 * n_meta, dtype=torch.uint8, pin_memory=True)
        host_boxes = staged[:off_labels].view(torch.float32).view(M, 4)
        host_labels = staged[off_labels:off_meta].view(torch.int64)
        host_meta = staged[off_meta:off_meta + 4 * n_meta].view(torch.int32)
        if self.total > 0:
            for t, n in zip(targets, self.sizes):
                if int(t["labels"].shape[0]) != n:
                    raise ValueError("a target's labels and boxes differ in length")
                if t["boxes"].dtype != torch.float32:
                    raise RuntimeError(f"target boxes must be float32, got {t['boxes'].dtype}")
            torch.cat([t["labels"].to(torch.int64) for t in targets], out=host_labels)
            torch.cat([t["boxes"].reshape(-1, 4) for t in targets], out=host_boxes)
        else:
            host_labels.zero_()
            host_boxes.zero_()
        meta = host_meta.numpy()
        meta[0] = 0
        np.cumsum(self.sizes, out=meta[1:B + 1])
        meta[B + 1:] = np.argsort(-np.asarray(self.sizes, dtype=np.int64), kind="stable")
        self.offsets_list = meta[:B + 1].tolist()
        dev = staged.to(device, non_blocking=True)
        self.boxes = dev[:off_labels].view(torch.float32).view(M, 4)
        self.labels = dev[off_labels:off_meta].view(torch.int64)
        self.meta = dev[off_meta:off_meta + 4 * n_meta].view(torch.int32)
        return self

    def _pair_maps(self):
        """Device int64 [M] maps for consumers of flat indices: image of every packed target and the
        packed position of that image's first target."""
        if getattr(self, "_maps", None) is None:
            sizes = np.asarray(self.sizes, dtype=np.int64)
            both = np.empty((2, max(self.total, 1)), dtype=np.int64)
            both[:, :] = 0
            both[0, :self.total] = np.repeat(np.arange(self.batch, dtype=np.int64), sizes)
            both[1, :self.total] = np.repeat(self.offsets_host[:-1].astype(np.int64), sizes)
            staged = torch.empty(both.shape, dtype=torch.int64, pin_memory=True)
            staged.numpy()[:] = both
            dev = staged.to(self.device, non_blocking=True)
            self._maps = (dev[0, :self.total], dev[1, :self.total])
        return self._maps

    def _filled_slots(self, queries: int) -> torch.Tensor:
        """Device int64 index of the packed slots a set of `queries` queries fills: the first min(Q, T_b) of every
        image (an image with more targets than queries leaves the rest of its slots untouched).  Made on the host
        from the sizes, uploaded from pinned memory without synchronising, kept per query count."""
        cache = self.__dict__.setdefault("_filled", {})
        dev = cache.get(queries)
        if dev is None:
            keep = np.flatnonzero(self.local_index < queries)
            staged = torch.empty(keep.shape[0], dtype=torch.int64, pin_memory=True)
            staged.numpy()[:] = keep
            dev = cache[queries] = staged.to(self.device, non_blocking=True)
        return dev

    @property
    def image_index(self) -> np.ndarray:
        """host int64 [M]: image of every packed target"""
        if getattr(self, "_image_index", None) is None:
            self._image_index = np.repeat(np.arange(self.batch, dtype=np.int64), self.sizes)
        return self._image_index

    @property
    def local_index(self) -> np.ndarray:
        """host int64 [M]: position of every packed target within its image"""
        if getattr(self, "_local_index", None) is None:
            self._local_index = np.arange(self.total, dtype=np.int64) - \
                np.repeat(self.offsets_host[:-1].astype(np.int64), self.sizes)
        return self._local_index

    @property
    def batch_index(self) -> torch.Tensor:
        return self._pair_maps()[0]

    @property
    def target_base(self) -> torch.Tensor:
        return self._pair_maps()[1]

    @property
    def offsets_ptr(self) -> int:
        return self.device_pointers()[2]

    @property
    def order_ptr(self) -> int:
        return self.device_pointers()[3]


class FlatMatches:
    """The matches of ONE output set as flat device tensors -- what the reference's losses build
    from the per-image tuples anyway (/root/reference/models/detr/loss.py:71-75, 102-103, 124-125):

        batch_idx, src_idx   -> outputs['pred_boxes'][batch_idx, src_idx], target_cls[batch_idx, src_idx] = ...
        tgt_idx              per-image target index (the reference's index_j, concatenated)
        tgt_flat             index into the packed targets: packed.boxes[tgt_flat], packed.labels[tgt_flat]

    Nothing here synchronises with the host."""

    def __init__(self, batch_idx, src_idx, tgt_idx, tgt_flat, packed, queries, raw=None, set_index=0):
        # src_idx / tgt_idx / tgt_flat None: rows `set_index` of the launch's index buffer, sliced on first use
        # (the fused consumers of detr_amd.losses read the buffer itself and never ask for them)
        self.batch_idx, self._src, self._tgt, self._flat = batch_idx, src_idx, tgt_idx, tgt_flat
        self.packed, self.queries = packed, queries
        # (index buffer [src (S*M) | tgt (S*M) | status (| flat (S*M))], S) of the launch this set came from and the
        # set's position in it: what detr_amd.losses hands to the fused consumer kernels
        self.raw, self.set_index = raw, set_index

    def _row(self, part: int) -> torch.Tensor:
        out, S = self.raw
        M = self.packed.total
        lo = (part * S + self.set_index) * M + (1 if part == 2 else 0)      # (the status word sits before the flat part)
        return out[lo:lo + M]

    @property
    def src_idx(self) -> torch.Tensor:
        if self._src is None:
            self._src = self._row(0)
        return self._src

    @property
    def tgt_idx(self) -> torch.Tensor:
        if self._tgt is None:
            self._tgt = self._row(1)
        return self._tgt

    @property
    def tgt_flat(self) -> torch.Tensor:
        if self._flat is None:
            self._flat = self._row(2)
        return self._flat

    @property
    def num_matched(self) -> int:
        return self.packed.total if self._src is None else int(self._src.shape[0])

    def src_permutation_idx(self):
        """(batch_idx, src_idx), the reference's `_get_src_permutation_idx` (detr/loss.py:71-75)."""
        return self.batch_idx, self.src_idx

    def target_boxes(self) -> torch.Tensor:
        """cat([t['boxes'][i] ...]) of detr/loss.py:103."""
        return self.packed.boxes[self.tgt_flat]

    def target_labels(self) -> torch.Tensor:
        """cat([t['labels'][i] ...]) of detr/loss.py:124."""
        return self.packed.labels[self.tgt_flat]

    def target_classes(self, no_object_class: int) -> torch.Tensor:
        """Dense [B, Q] class map of detr/loss.py:121-125."""
        dense = torch.full((self.packed.batch, self.queries), no_object_class, dtype=torch.int64,
                           device=self.src_idx.device)
        dense[self.batch_idx, self.src_idx] = self.target_labels()
        return dense


def _raise_for_status(bits: int):
    if bits & _lib.ST_BAD_SIZES:
        # not one of the reference's errors: host-side sizes that contradict the device-side target offsets
        raise RuntimeError("max_targets is smaller than an image's number of targets (device-side offsets)")
    # same precedence as the reference: the class-cost gather (IndexError) runs before the
    # box asserts (AssertionError), which run before scipy (ValueError)
    if bits & _lib.ST_LABEL_RANGE:
        raise IndexError("target label out of range for the logits' class dimension")
    if bits & _lib.ST_DEGENERATE_BOX:
        raise AssertionError("degenerate box (x1 < x0, y1 < y0 or NaN) in predictions or targets")
    if bits & _lib.ST_INVALID_COST:
        raise ValueError("matrix contains invalid numeric entries")
    if bits & _lib.ST_INFEASIBLE:
        raise ValueError("cost matrix is infeasible")


class HungarianMatcher(torch.nn.Module):
    """Assignment between targets and predictions (reference: models/matcher.py:13-36)."""

    def __init__(self, cls_loss: str = "ce", cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1,
                 focal_alpha: float = 0.25, focal_gamma: float = 2.0):
        super().__init__()
        self.cost_class = cost_class
        self.cost_bbox = cost_bbox
        self.cost_giou = cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        self.cls_loss = cls_loss
        self.focal_alpha = focal_alpha
        self.focal_gamma = focal_gamma
        self.last_status = None
        # True: call the kernels through the registered operators torch.ops.detr_amd.* also in eager mode
        # (they are always used while torch.compile traces); False: the same C entry points directly,
        # which saves the operator dispatch and a fresh index buffer per call
        self.through_ops = False
        # True: the kernels store the indices of a CPU-tuple call straight into the pinned host tensor the
        # tuples are views of (the stores cross the bus while the kernel runs); False: into a device buffer
        # that a copy engine transfers afterwards
        self.indices_to_host = True
        self._events = {}
        self._buffers = {}
        self._packed = None
        self._params_cache = None
        self._prefetched = None

    def __getstate__(self):
        # per-device events and buffers, the packed targets and the last status view belong to this
        # process and its streams
        state = self.__dict__.copy()
        state["_events"] = {}
        state["_buffers"] = {}
        state["_packed"] = None
        state["_params_cache"] = None
        state["_prefetched"] = None
        state["last_status"] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        for name, value in (("_events", {}), ("_buffers", {}), ("_packed", None), ("_params_cache", None), ("_prefetched", None),
                            ("through_ops", False), ("indices_to_host", True)):
            self.__dict__.setdefault(name, value)

    def _done_event(self, device) -> torch.cuda.Event:
        key = device.index if device.index is not None else torch.cuda.current_device()
        ev = self._events.get(key)
        if ev is None:
            with torch.cuda.device(device):
                ev = self._events[key] = torch.cuda.Event()
        return ev

    def _params(self) -> DmParams:
        return DmParams(0 if self.cls_loss == "ce" else 1, float(self.cost_class), float(self.cost_bbox),
                        float(self.cost_giou), float(self.focal_alpha), float(self.focal_gamma))

    def _params_ref(self):
        """ctypes reference to the hyper-parameters, rebuilt only when an attribute has changed."""
        key = (self.cls_loss, self.cost_class, self.cost_bbox, self.cost_giou, self.focal_alpha, self.focal_gamma)
        cached = self._params_cache
        if cached is None or cached[0] != key:
            params = self._params()
            cached = self._params_cache = (key, params, ctypes.byref(params), ctypes.addressof(params))
        return cached[2]

    def forward(self, outputs: Dict[str, torch.Tensor], targets) -> Indices:
        """One output set; same contract as the reference's forward (matcher.py:38-95).  (No gradient is
        recorded: match_many runs under torch.no_grad.)"""
        ahead = self.__dict__.get("_prefetched")
        if ahead:
            hit = ahead.get(id(outputs))
            if hit is not None:
                found = self._prefetch_hit(hit, outputs, targets)
                if found is not None:
                    return found
        return self.match_many([outputs], targets)[0]

    def prefetch(self, outputs_list: Sequence[Dict[str, torch.Tensor]], targets, device_indices: bool = False,
                 flat: bool = False) -> None:
        """Match all of `outputs_list` now, in one `match_many` call, and hand the results out when `forward` is
        called with one of these very output dicts and these very targets (same objects, tensors unchanged since):
        what lets a criterion that calls the matcher once per decoder layer
        (/root/reference/models/detr/loss.py:164-169) run at the speed of one batched call without a change to
        its loop.  A call with anything else is matched on the spot as usual.  `clear_prefetched` drops the
        results (and the references to the predictions they are keyed by).
        `device_indices`: the results are views of a device buffer instead of CPU tensors (no host synchronisation;
        input errors are then raised by `check_status()`); every expression the reference's losses apply to the
        indices (detr/loss.py:71-75, 102-103, 124-125) takes them, and gathers with them need no index upload.
        `flat` (implies device indices): the launch's `FlatMatches` are kept beside the tuples, for consumers that
        recognise a served result (`served_flat`) and hand it to the fused kernels of detr_amd.losses."""
        state = self.__dict__
        state["_prefetched"] = None
        outputs_list = list(outputs_list)
        if not outputs_list:
            return
        flats = None
        if flat:
            flats, results = self.match_flat(outputs_list, targets, with_tuples=True)
        else:
            results = self.match_many(outputs_list, targets, device_indices=device_indices)
        packed = self._packed          # (what the call has just made of `targets`: knows whether it is handed them again)
        table = {}
        for k, (o, r) in enumerate(zip(outputs_list, results)):
            l, x = o["pred_logits"], o["pred_boxes"]
            table[id(o)] = (o, l, x, l._version, x._version, targets, packed, r, flats[k] if flats else None)
            if flats:
                table[("indices", id(r))] = (r, flats[k])
        state["_prefetched"] = table

    def clear_prefetched(self) -> None:
        self.__dict__["_prefetched"] = None

    def served_flat_of(self, indices) -> Optional["FlatMatches"]:
        """The `FlatMatches` behind `indices` when that is the very list `prefetch(..., flat=True)` has served for one
        of its output sets; None for indices that came from elsewhere."""
        ahead = self.__dict__.get("_prefetched")
        hit = ahead.get(("indices", id(indices))) if ahead else None
        return hit[1] if hit is not None and hit[0] is indices else None

    def register_flat_view(self, indices_like, flat: "FlatMatches") -> None:
        """Make `served_flat_of(indices_like)` answer `flat` until the prefetched results are dropped: for a consumer
        that hands a criterion an equivalent, cheaper stand-in for a served result (detr_amd.criterion)."""
        ahead = self.__dict__.get("_prefetched")
        if ahead is None:
            ahead = self.__dict__["_prefetched"] = {}
        ahead[("indices", id(indices_like))] = (indices_like, flat)

    def served_flat(self, outputs, indices) -> Optional["FlatMatches"]:
        """The `FlatMatches` of `outputs` when `indices` is the very result `prefetch(..., flat=True)` has ready for that
        output dict (and its prediction boxes are unchanged since); None for anything else -- indices that came from
        elsewhere (the denoising groups bring their own) or a matcher call that was not served from the prefetch."""
        ahead = self.__dict__.get("_prefetched")
        hit = ahead.get(id(outputs)) if ahead else None
        if hit is None or hit[0] is not outputs or hit[7] is not indices or hit[8] is None:
            return None
        x = outputs["pred_boxes"]
        if x is not hit[2] or x._version != hit[4] or x.dtype != torch.float32:
            return None
        return hit[8]

    def _prefetch_hit(self, hit, outputs, targets):
        o, l, x, lv, xv, tg, packed, result, _ = hit
        if o is not outputs or outputs["pred_logits"] is not l or outputs["pred_boxes"] is not x \
                or l._version != lv or x._version != xv:
            return None
        if targets is not tg or packed is None or not packed.same_targets(targets):
            return None
        return result

    @staticmethod
    def _prepare_sets(outputs_list, batch: int):
        return ops.check_predictions([o["pred_logits"] for o in outputs_list],
                                     [o["pred_boxes"] for o in outputs_list], batch)

    def _packed_for(self, targets, device) -> PackedTargets:
        """The packed form of `targets`; the previous call's when these are the same unchanged tensors
        (the reference's criteria call the matcher once per decoder layer with the same targets,
        /root/reference/models/detr/loss.py:164-169)."""
        if isinstance(targets, PackedTargets):
            return targets
        cached = self._packed
        if cached is not None and cached.device == device and cached.same_targets(targets):
            return cached
        # (plain dictionary stores: nn.Module.__setattr__ costs a microsecond per assignment)
        state = self.__dict__
        state["_packed"] = None          # (a failed packing must not leave a stale entry)
        packed = PackedTargets(targets, device, defer=True)
        state["_packed"] = packed
        return packed

    def _workspace(self, device, n_out: int, n_cost: int):
        """Device buffers that live across calls, per device: the cost blocks (between the two kernels)
        and, for the calls that hand CPU tensors back, the index buffer [status | src | tgt] -- its
        status word sits in front so that it stays where it is when the buffer grows, and it is
        cleared only after a call that set it."""
        key = device.index if device.index is not None else torch.cuda.current_device()
        ws = self._buffers.get(key)
        if ws is None:
            ws = self._buffers[key] = {"out": None, "cost": None}
        if n_out and (ws["out"] is None or ws["out"].shape[0] < n_out):
            ws["out"] = torch.zeros(max(n_out, 4096), dtype=torch.int64, device=device)
        if n_cost and (ws["cost"] is None or ws["cost"].shape[0] < n_cost):
            ws["cost"] = torch.empty(n_cost, dtype=torch.float32, device=device)
        return ws

    def _launch_to_host(self, outputs_list, packed: PackedTargets, device):
        """The usual step in one call of the host helper: output sets checked, pinned result tensor
        [status | src | tgt] made, dm_match_targets_f32 called on the targets where they are.  Returns
        (host tensor, queries per set), or None when the step is not of that kind (the general path then decides)."""
        pending = packed.__dict__.get("_unpacked")
        if pending is None or _index_views is None:
            return None
        index = torch.cuda.current_device()
        ws = self._buffers.get(index)
        cost = ws["cost"] if ws is not None else None
        self._params_ref()
        while True:
            # (room for the packed form of the targets is only allocated when a call says that it has to pack)
            room = packed._pointers[:3] if packed._pointers is not None else (0, 0, 0)
            r = _index_views.launch_targets(outputs_list, packed.batch, packed.total, packed.max_t, pending[0], pending[1],
                                            _TARGETS_ENTRY, self._params_cache[3], cost.data_ptr() if cost is not None else 0,
                                            cost.shape[0] if cost is not None else 0, _raw_stream(index), index, *room)
            if r is None:
                return None
            if type(r) is int:          # a larger workspace is needed: nothing was launched
                cost = self._workspace(device, 0, r)["cost"]
                continue
            if r[2] == _lib.NEED_BUFFERS and packed._pointers is None:      # nothing was launched either
                packed._pack_buffer()
                continue
            break
        host, queries, rc, did_pack = r
        if rc:
            _lib.check(rc, "dm_match_targets_f32")
        if did_pack:
            if packed.total == 0:
                packed._buf[:packed._layout[2]].zero_()
            packed._unpacked = None
        self.__dict__["last_status"] = None
        return host, queries

    def _launch(self, outputs_list, packed: PackedTargets, flat: bool = False, own_buffers: bool = False,
                want_cost: bool = False, host_out: Optional[torch.Tensor] = None):
        """Enqueue the cost and assignment kernels for all output sets on the current stream.
        Returns the device index buffer, the queries per set, the cost buffer and, per set, (offset, row
        stride) of its cost region.  Index buffer: a fresh [src (S*M) | tgt (S*M) | status word (| flat
        (S*M))] that the caller may keep, or with `own_buffers` this matcher's own [status | src | tgt],
        valid until its next call -- or `host_out`, a pinned host tensor of that layout whose status word is
        zero, which the kernels then write through the bus.  Unless `want_cost`, the assignment kernels compute their cost blocks
        themselves where the shapes allow it (dm_match_fused_f32) and the cost buffer is only a workspace."""
        S, batch, M = len(outputs_list), packed.batch, packed.total
        if self.through_ops or torch.compiler.is_compiling():
            # the same entry points as registered operators (detr_amd/ops.py): what tracing sees
            logits_list = [o["pred_logits"] for o in outputs_list]
            boxes_list = [o["pred_boxes"] for o in outputs_list]
            prm = (0 if self.cls_loss == "ce" else 1, float(self.cost_class), float(self.cost_bbox),
                   float(self.cost_giou), float(self.focal_alpha), float(self.focal_gamma))
            common = (logits_list, boxes_list, packed.labels, packed.boxes, packed.meta, M, packed.max_t) + prm
            out = ops.match(*common, flat)
            queries = [int(l.shape[1]) for l in logits_list]
            cost = ops.cost(*common) if want_cost else None
            self.last_status = out[2 * S * M:2 * S * M + 1]
            return out, queries, cost, ops.set_array(None, queries, M)[2]
        helper = _helper()
        scan = helper.scan_outputs(outputs_list, batch, M) if helper is not None else None
        if scan is not None:      # the usual case: float32, rows dense within every image, one device, one class count
            index, C, queries, sets, cost_off = scan
            device = outputs_list[0]["pred_logits"].device
            keep = None
        else:
            device, C, keep, queries = self._prepare_sets(outputs_list, batch)
            sets, cost_off, _ = ops.set_array(keep, queries, M)
        if packed.device != device:
            raise ValueError("targets and predictions must be on the same device")
        strides = ops.set_array(None, queries, M)[2] if want_cost else None

        SM = S * M
        if own_buffers:
            ws = self._workspace(device, 0 if host_out is not None else 2 * SM + 1, max(cost_off, 1))
            out, cost = (host_out if host_out is not None else ws["out"]), ws["cost"]
            status_ptr = out.data_ptr()
            src_ptr = status_ptr + 8
            flat_ptr = None
            self.last_status = None
        else:
            cost = torch.empty(max(cost_off, 1), dtype=torch.float32, device=device)
            # flat: a third index array behind the status word (position among the packed targets)
            out = torch.zeros((3 if flat else 2) * SM + 1, dtype=torch.int64, device=device)
            src_ptr = out.data_ptr()
            status_ptr = src_ptr + 16 * SM
            flat_ptr = status_ptr + 8 if flat else None
            self.last_status = out[2 * SM:2 * SM + 1]
        current = torch.cuda.current_device()
        on_current = device.index is None or device.index == current
        stream = _raw_stream(current if device.index is None else device.index)
        entry = lib.dm_match_flat_f32 if want_cost else lib.dm_match_fused_f32
        pending = packed.__dict__.get("_unpacked")
        if pending is not None and not want_cost:
            # targets not packed yet: one call that packs and matches -- or, when the step is one fused kernel,
            # only matches, the kernel reading the targets where they are
            pointers, meta, _ = pending
            did_pack = ctypes.c_int(0)
            while True:
                # (room for the packed form is only allocated when the call says it has to pack)
                labels_ptr, boxes_ptr, meta_ptr = packed._pointers[:3] if packed._pointers is not None else (None, None, None)
                args = (sets, S, batch, C, pointers, meta, packed.max_t, self._params_ref(), labels_ptr, boxes_ptr, meta_ptr,
                        cost.data_ptr(), src_ptr, src_ptr + 8 * SM, flat_ptr, status_ptr, stream, ctypes.byref(did_pack))
                if on_current:
                    rc = lib.dm_match_targets_f32(*args)
                else:
                    with torch.cuda.device(device):
                        rc = lib.dm_match_targets_f32(*args)
                if rc != _lib.NEED_BUFFERS or packed._pointers is not None:
                    break
                packed._pack_buffer()
            _lib.check(rc, "dm_match_targets_f32")
            if did_pack.value:
                if packed.total == 0:
                    packed._buf[:packed._layout[2]].zero_()
                packed._unpacked = None
            return out, queries, cost, strides
        labels_ptr, boxes_ptr, offsets_ptr, order_ptr = packed.device_pointers()
        if on_current:
            rc = entry(sets, S, batch, C, labels_ptr, boxes_ptr, offsets_ptr, order_ptr, packed.max_t,
                       self._params_ref(), cost.data_ptr(), src_ptr, src_ptr + 8 * SM, flat_ptr, status_ptr, stream)
        else:
            with torch.cuda.device(device):
                rc = entry(sets, S, batch, C, labels_ptr, boxes_ptr, offsets_ptr, order_ptr, packed.max_t,
                           self._params_ref(), cost.data_ptr(), src_ptr, src_ptr + 8 * SM, flat_ptr, status_ptr, stream)
        _lib.check(rc, "dm_match_flat_f32" if want_cost else "dm_match_fused_f32")
        return out, queries, cost, strides

    @staticmethod
    def _float32_inputs(outputs_list, targets):
        """The reference computes in whatever dtype it is handed (matcher.py:59-90): float64 predictions and targets
        work there, float64 logits beside float32 boxes too; prediction boxes and target boxes of different dtypes
        raise RuntimeError (torch.cdist), half precision raises NotImplementedError.  The kernels compute in float32:
        float64 inputs are rounded to float32 first (the cost then differs from the reference's float64 cost by
        float32 rounding, ~1e-7 relative; the assignment only where two optima are that close)."""
        first = outputs_list[0]
        if first["pred_logits"].dtype == torch.float32 and first["pred_boxes"].dtype == torch.float32:
            return outputs_list, targets           # the usual case: nothing to look at
        allowed = (torch.float32, torch.float64)
        packed = isinstance(targets, PackedTargets)
        box_dtypes = {o["pred_boxes"].dtype for o in outputs_list}
        box_dtypes |= {torch.float32} if packed else {t["boxes"].dtype for t in targets}
        every = box_dtypes | {o["pred_logits"].dtype for o in outputs_list}
        if any(d in (torch.float16, torch.bfloat16) for d in every):
            raise NotImplementedError("half-precision matching inputs are not supported (the reference's torch.cdist "
                                      "has no kernel for them either): cast predictions and targets to float32")
        if len(box_dtypes) != 1 or not every <= set(allowed):
            raise RuntimeError("prediction boxes and target boxes must have one floating dtype (float32 or float64), "
                               "got %s" % sorted(str(d) for d in every))
        outputs_list = [dict(o, pred_logits=o["pred_logits"].float(), pred_boxes=o["pred_boxes"].float())
                        for o in outputs_list]
        if not packed:
            targets = [dict(t, boxes=t["boxes"].float()) for t in targets]
        return outputs_list, targets

    @staticmethod
    def _by_class_count(outputs_list):
        """None when all output sets have the same number of logits per query (one pass handles them);
        otherwise the positions of the sets of each class count -- heads with different class counts, e.g. a
        class-agnostic first stage beside the decoder layers, take one pass per class count on the same targets."""
        first = outputs_list[0]["pred_logits"].shape[-1]
        if all(o["pred_logits"].shape[-1] == first for o in outputs_list):
            return None
        groups = {}
        for i, o in enumerate(outputs_list):
            groups.setdefault(int(o["pred_logits"].shape[-1]), []).append(i)
        return list(groups.values())

    def _match_groups(self, groups, outputs_list, targets, device_indices, return_cost, while_waiting=None):
        """match_many for output sets of several class counts: one pass per class count, results in the caller's
        order; the status of a device-resident call is the OR over the passes."""
        S = len(outputs_list)
        result, costs, statuses = [None] * S, [None] * S, []
        for members in groups:
            part = self.match_many([outputs_list[i] for i in members], targets, device_indices, return_cost,
                                   while_waiting)
            while_waiting = None        # (once per call)
            if self.last_status is not None:
                statuses.append(self.last_status)
            for k, i in enumerate(members):
                result[i] = part[0][k] if return_cost else part[k]
                if return_cost:
                    costs[i] = part[1][k]
        if statuses:
            self.last_status = functools.reduce(torch.bitwise_or, statuses)
        return (result, costs) if return_cost else result

    @torch.no_grad()
    def match_many(self, outputs_list: Sequence[Dict[str, torch.Tensor]],
                   targets: Union[PackedTargets, Sequence[Dict[str, torch.Tensor]]],
                   device_indices: bool = False, return_cost: bool = False, while_waiting=None):
        """Match S output sets against the same targets.

        Returns a list (per set) of lists (per image) of ``(src, tgt)`` int64 tensors --
        on the CPU by default, like the reference; views of one device buffer with
        ``device_indices=True`` (no host synchronisation; input errors are then only
        raised by ``check_status()``).  With ``return_cost=True`` also returns, per set,
        the ``[M, Q]`` device cost tensor (rows = packed targets).

        ``while_waiting`` (CPU tuples only): a callable run once after the kernels have been enqueued and before
        the host waits for them -- host work of the step that does not need the indices (issuing the collective
        of the loss normalisation, for instance) then costs no time of its own.

        One matcher object serves one host thread at a time (it keeps device buffers between calls).
        """
        S = len(outputs_list)
        if S == 0:
            return []
        outputs_list, targets = self._float32_inputs(outputs_list, targets)
        device = outputs_list[0]["pred_logits"].device
        packed = self._packed_for(targets, device)
        M, sizes = packed.total, packed.sizes
        if packed.batch == 0:
            return [[] for _ in range(S)]

        if device_indices:
            groups = self._by_class_count(outputs_list)
            if groups is not None:
                return self._match_groups(groups, outputs_list, targets, device_indices, return_cost)
            out, queries, cost, strides = self._launch(outputs_list, packed, want_cost=return_cost)
            result = self._split(out, S, M, sizes, queries, packed.offsets_list)
        else:
            own = not (return_cost or self.through_ops or torch.compiler.is_compiling())
            n = 2 * S * M + 1
            direct = own and self.indices_to_host and device.index in (None, torch.cuda.current_device())
            fast = self._launch_to_host(outputs_list, packed, device) if direct else None
            if fast is None:
                # (the one-call path has checked the class counts itself; anything else is checked here)
                groups = self._by_class_count(outputs_list)
                if groups is not None:
                    return self._match_groups(groups, outputs_list, targets, device_indices, return_cost, while_waiting)
            if fast is not None:
                host, queries = fast
                status_word = ctypes.c_int64.from_address(host.data_ptr())
                done = self._done_event(device)
                done.record()
            elif direct:
                host = torch.empty(n, dtype=torch.int64, pin_memory=True)
                status_word = ctypes.c_int64.from_address(host.data_ptr())
                status_word.value = 0        # the kernels only ever OR bits into it
                out, queries, cost, strides = self._launch(outputs_list, packed, own_buffers=True, host_out=host)
                done = self._done_event(device)
                done.record()                # current device, current stream: where the kernels went
            else:
                host = torch.empty(n, dtype=torch.int64, pin_memory=True)
                out, queries, cost, strides = self._launch(outputs_list, packed, own_buffers=own, want_cost=return_cost)
                host.copy_(out[:n], non_blocking=True)
                done = self._done_event(device)
                done.record(torch.cuda.current_stream(device))     # the stream the kernels and the copy went to
            try:
                if while_waiting is not None:
                    while_waiting()
                # the per-image views depend on shapes only: build them while the GPU is working
                if own:
                    result, status_at = self._split(host[1:], S, M, sizes, queries, packed.offsets_list), 0
                else:
                    result, status_at = self._split(host, S, M, sizes, queries, packed.offsets_list), n - 1
            except BaseException:
                # the kernels (or the copy) are still writing into `host`: it must not go back to the pinned
                # allocator, which knows nothing of that use, before they have finished
                done.synchronize()
                raise
            # by now the copy has normally landed: one event query is several times cheaper than a
            # stream synchronisation, which is only the fallback
            if not done.query():
                done.synchronize()
            bits = (status_word.value if direct else int(host[status_at])) & 0xFFFFFFFF
            if bits:
                if own and not direct:
                    out[:1].zero_()          # the kept buffer's status word, for the next call
                _raise_for_status(bits)
        if return_cost:
            costs = [cost[o:o + M * st].view(M, st)[:, :q] for (o, st), q in zip(strides, queries)]
            return result, costs
        return result

    @staticmethod
    def _split(flat, S, M, sizes, queries, offsets):
        """Views of the flat [src | tgt | status] buffer as list (sets) of lists (images) of tuples."""
        if M == 0:
            empty = flat[:0]
            return [[(empty, empty) for _ in sizes] for _ in range(S)]
        helper = _helper()
        if helper is not None and flat.dtype == torch.int64:
            return helper.index_pairs(flat, M, sizes, queries, offsets)
        src_all, tgt_all = flat[:S * M], flat[S * M:2 * S * M]
        B = len(sizes)
        if min(queries) >= max(sizes):
            # every image's region is completely filled: one split call per side for everything
            # (index tensors carry no gradient, so the untracked split is safe and cheaper)
            src_parts = _split_with_sizes(src_all, sizes * S)
            tgt_parts = _split_with_sizes(tgt_all, sizes * S)
            pairs = list(zip(src_parts, tgt_parts))
            return [pairs[s * B:(s + 1) * B] for s in range(S)]
        result = []
        for s, q in enumerate(queries):
            per_image = []
            for b, t in enumerate(sizes):
                lo = s * M + int(offsets[b])
                n = min(q, t)
                per_image.append((src_all[lo:lo + n], tgt_all[lo:lo + n]))
            result.append(per_image)
        return result

    @torch.no_grad()
    def match_flat(self, outputs_list, targets, with_tuples: bool = False):
        """`match_many` for consumers that want flat device-resident indices: one `FlatMatches`
        per output set, no device-to-host copy, no synchronisation (input errors surface through
        `check_status()`).  `with_tuples`: returns (flat matches, the per-image index tuples of
        `match_many(device_indices=True)`) -- two views of the same launch's index buffer."""
        S = len(outputs_list)
        if S == 0:
            return ([], []) if with_tuples else []
        outputs_list, targets = self._float32_inputs(outputs_list, targets)
        groups = self._by_class_count(outputs_list)
        if groups is not None:
            result, tuples, statuses = [None] * S, [None] * S, []
            for members in groups:
                part = self.match_flat([outputs_list[i] for i in members], targets, with_tuples=True)
                statuses.append(self.last_status)
                for k, i in enumerate(members):
                    result[i] = part[0][k] if part[0] else None
                    tuples[i] = part[1][k] if part[1] else []
            if all(r is None for r in result):
                return ([], tuples) if with_tuples else []
            self.last_status = functools.reduce(torch.bitwise_or, [st for st in statuses if st is not None])
            return (result, tuples) if with_tuples else result
        device = outputs_list[0]["pred_logits"].device
        packed = self._packed_for(targets, device)
        if packed.batch == 0:
            return ([], [[] for _ in range(S)]) if with_tuples else []
        out, queries, _, _ = self._launch(outputs_list, packed, flat=True)
        M, sizes = packed.total, packed.sizes
        batch_idx = packed.batch_index
        raw = (out, S)
        result = []
        both = flat_all = None
        for s, q in enumerate(queries):
            if M == 0 or q >= packed.max_t:
                # every target of every image is matched: the set's rows of the buffer, sliced when asked for
                result.append(FlatMatches(batch_idx, None, None, None, packed, q, raw, s))
                continue
            if both is None:
                both = out[:2 * S * M].view(2, S, M)
                flat_all = out[2 * S * M + 1:].view(S, M)             # written by the assignment kernel itself
            src, tgt, flat = both[0, s], both[1, s], flat_all[s]
            # images with more targets than queries leave holes: keep the first min(Q, T_b) slots
            keep_dev = packed._filled_slots(q)
            result.append(FlatMatches(batch_idx[keep_dev], src[keep_dev], tgt[keep_dev], flat[keep_dev], packed, q,
                                      (out, S), s))
        if with_tuples:
            return result, self._split(out, S, M, sizes, queries, packed.offsets_list)
        return result

    def check_status(self):
        """Raises the exception the reference would have raised for the last
        ``device_indices=True`` call (synchronises)."""
        if self.last_status is not None:
            _raise_for_status(int(self.last_status.item()) & 0xFFFFFFFF)

    @torch.no_grad()
    def cost_blocks(self, outputs: Dict[str, torch.Tensor], targets) -> List[torch.Tensor]:
        """Per-image ``[Q, T_b]`` cost blocks of one output set (device tensors), laid out
        like the blocks the reference hands to scipy (matcher.py:91-94)."""
        packed = targets if isinstance(targets, PackedTargets) else \
            PackedTargets(targets, outputs["pred_logits"].device)
        _, costs = self.match_many([outputs], packed, device_indices=True, return_cost=True)
        torch.cuda.current_stream(outputs["pred_logits"].device).synchronize()
        bits = int(self.last_status.item()) & 0xFFFFFFFF
        _raise_for_status(bits & (_lib.ST_LABEL_RANGE | _lib.ST_DEGENERATE_BOX))
        off = packed.offsets_host
        return [costs[0][int(off[b]):int(off[b + 1])].t() for b in range(packed.batch)]
