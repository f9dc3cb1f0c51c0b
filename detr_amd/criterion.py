"""One line that gives an unmodified criterion the batched matching call.

The reference's criteria walk ``outputs['model']`` (one dict per decoder layer), for the two-stage
models also ``outputs['first_stage']``, and call ``self.matcher(output, targets)`` once per dict
(/root/reference/models/detr/loss.py:160-183, dino_detr/loss.py:162-203, co_detr/loss.py:195-246; the
denoising groups 'cdn' / 'noised' bring their indices with them and never reach the matcher).  Each of
those calls is a kernel that takes as long as the heaviest image of the batch, plus a completion round
trip -- six or seven times per step.

``attach(criterion)`` swaps ``criterion.matcher`` for a :class:`detr_amd.HungarianMatcher` with the same
hyper-parameters (and Co-DETR's auxiliary assigners in ``criterion.aux_heads``,
/root/reference/models/co_detr/loss.py:64,223-241, for their GPU counterparts in :mod:`detr_amd.aux_assign`) and registers a forward pre-hook on the criterion: when the criterion is called, the hook
sees the whole ``outputs`` dict, matches all the sets the loop is going to ask for in ONE
``match_many`` call (``HungarianMatcher.prefetch``), and the loop's per-set calls find their result
waiting.  The criterion's code stays as it is; a set the hook did not foresee is matched on the spot.
"""
from typing import Iterable, Sequence

import numpy as np
import torch

from .matcher import HungarianMatcher

#: the keys of `outputs` whose sets the reference's criteria hand to the Hungarian matcher
MATCHED_KEYS = ("model", "first_stage")


def matcher_like(other) -> HungarianMatcher:
    """A detr_amd matcher with the hyper-parameters of `other` (the reference's matcher object, or anything
    with the same attributes: models/matcher.py:29-36)."""
    if isinstance(other, HungarianMatcher):
        return other
    return HungarianMatcher(cls_loss=other.cls_loss, cost_class=other.cost_class, cost_bbox=other.cost_bbox,
                            cost_giou=other.cost_giou, focal_alpha=other.focal_alpha, focal_gamma=other.focal_gamma)


def sets_the_loop_will_match(outputs: dict, keys: Iterable[str] = MATCHED_KEYS, last_only: bool = False):
    """The output dicts a criterion's loop hands to the matcher, in its order: every dict of ``outputs[key]``
    (only the last one with `last_only`, the reference's ``aux_flag = False``: `_outputs[-1:]`)."""
    found = []
    for key in keys:
        group = outputs.get(key) if isinstance(outputs, dict) else None
        if not group:
            continue
        if isinstance(group, dict):
            group = [group]
        if last_only:
            group = group[-1:]
        for o in group:
            if isinstance(o, dict) and "pred_logits" in o and "pred_boxes" in o:
                found.append(o)
    return found


def aux_head_like(other):
    """The detr_amd counterpart of one of the reference's auxiliary assigners
    (/root/reference/models/co_detr/auxiliary_head.py:17-21, 63-67), told apart by the parameters they keep;
    anything else is returned as it is."""
    from . import aux_assign

    if isinstance(other, (aux_assign.ATSS, aux_assign.FasterRCNN)):
        return other
    if hasattr(other, "pos_threshold") and hasattr(other, "neg_threshold"):
        return aux_assign.FasterRCNN(other.pos_threshold, other.neg_threshold)
    if hasattr(other, "k") and type(other).__name__.lower().startswith("atss"):
        return aux_assign.ATSS(other.k)
    return other


def fuse_cls_criterion(own, seen_devices=None) -> bool:
    """Put the one-pass kernels of :mod:`detr_amd.losses` in front of `own.forward` when `own` is one of the reference's
    classification criteria -- ``FocalLoss(alpha, gamma, n_cls)`` (/root/reference/models/detr/loss.py:198-213) or
    ``CrossEntropy(weight)`` around an ``nn.CrossEntropyLoss`` (:185-195) -- told by their class name and the attributes
    they keep.  The module object, its buffers and its ``state_dict`` stay as they are: the replacement is an instance
    attribute ``forward`` (removed by `unfuse_cls_criterion`), which reads the parameters from `own` at every call and
    hands anything the kernels do not take (CPU or non-float32 tensors, a [B, Q, C] / [B, C, Q] layout that cannot be told
    because Q == C) to the module's own forward.  Returns whether `own` was recognised."""
    from . import losses

    if "forward" in own.__dict__:
        return True
    name = type(own).__name__
    own_forward = own.forward
    if name == "FocalLoss" and all(hasattr(own, a) for a in ("alpha", "gamma", "n_cls")):
        def forward(inputs, targets, num_boxes):
            x = losses._class_last(inputs, targets, own.n_cls, None)
            if x is None or x.shape[-1] != own.n_cls or not losses._takes(x, targets):
                return own_forward(inputs, targets, num_boxes)
            if seen_devices is not None:
                seen_devices.add(x.device)
            return losses.focal_loss(x, targets, num_boxes, own.alpha, own.gamma)
    elif name == "CrossEntropy" and isinstance(getattr(own, "criterion", None), torch.nn.CrossEntropyLoss) \
            and own.criterion.reduction == "mean" and own.criterion.label_smoothing == 0.0:
        def forward(inputs, targets, num_boxes=None):
            inner = own.criterion
            w = inner.weight
            C = w.shape[0] if w is not None else None
            x = losses._class_last(inputs, targets, C, None) if C is not None else None
            if x is None or not losses._takes(x, targets) or w.device != x.device or w.dtype != torch.float32:
                return own_forward(inputs, targets, num_boxes)
            if seen_devices is not None:
                seen_devices.add(x.device)
            return losses.cross_entropy(x, targets, w, inner.ignore_index)
    else:
        return False
    own.forward = forward          # (an instance attribute in front of the class's method)
    return True


def unfuse_cls_criterion(own) -> None:
    if isinstance(own, torch.nn.Module) and "forward" in own.__dict__:
        del own.__dict__["forward"]


def flat_of_foreign_indices(matcher: HungarianMatcher, outputs, targets, indices):
    """A `FlatMatches` for index tuples that did NOT come from the matcher -- the denoising groups of DN-DETR / DINO /
    Co-DETR bring their own `(query indices, target indices)` per image together with their own targets
    (/root/reference/models/dino_detr/loss.py:170-172, built at dino_func.py:117-129) -- so that the consumers of
    :mod:`detr_amd.losses` can work on them like on a served result: image, query and packed-target position of every
    pair as three device tensors, the targets packed once.  Built once per `indices` object and kept until the matcher's
    prefetched results are dropped (the same indices serve every decoder layer of a step).  None when these are not
    per-image pairs of int64 index tensors for GPU predictions -- the caller then uses its own code."""
    from . import losses, matcher as matcher_module

    FlatMatches, PackedTargets = matcher_module.FlatMatches, matcher_module.PackedTargets
    flat = matcher.served_flat_of(indices)
    if flat is not None:
        return flat if getattr(flat, "foreign_targets", None) is targets else None
    pred = outputs.get("pred_boxes") if isinstance(outputs, dict) else None
    if pred is None or not losses._takes_boxes(pred) or not isinstance(indices, (list, tuple)):
        return None
    B = int(pred.shape[0])
    if len(indices) != B or not isinstance(targets, (list, tuple)) or len(targets) != B:
        return None
    try:
        srcs, tgts = [s for s, _ in indices], [t for _, t in indices]
        for s_, t_ in zip(srcs, tgts):
            if s_.dtype != torch.int64 or t_.dtype != torch.int64 or s_.dim() != 1 or s_.shape != t_.shape \
                    or s_.device != srcs[0].device or t_.device != srcs[0].device:
                return None
        packed = matcher.__dict__.get("_packed")          # (the step's own targets: packed by the matching call already)
        if packed is None or packed.device != pred.device or not packed.same_targets(targets):
            packed = PackedTargets(targets, pred.device)
    except (TypeError, ValueError, RuntimeError, KeyError, AttributeError):
        return None
    lens = np.asarray([int(s_.shape[0]) for s_ in srcs], dtype=np.int64)
    n = int(lens.sum())
    if n == 0:
        return None
    # (numpy: torch's CPU repeat_interleave wakes its thread pool, a millisecond for 64 numbers)
    image = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int64), lens))
    first = torch.from_numpy(np.repeat(np.asarray(packed.offsets_list[:-1], dtype=np.int64), lens))
    if srcs[0].device.type == "cpu":            # (the reference builds them with torch.arange on the host)
        query, local = torch.cat(srcs), torch.cat(tgts)
        # torch's indexing also takes negative positions and raises for those beyond the end: such lists stay with
        # the criterion's own code (checked here where it costs nothing; indices on the device are checked by the
        # kernel against the batch-wide bounds only)
        limit = torch.from_numpy(np.repeat(np.asarray(packed.sizes, dtype=np.int64), lens))
        if bool((local < 0).any()) or bool((local >= limit).any()) or bool((query < 0).any()) or bool((query >= pred.shape[1]).any()):
            return None
        trio = torch.stack([image, query, local + first]).to(pred.device)
        image, query, slot = trio[0], trio[1], trio[2]
    else:
        image, query, slot = image.to(pred.device), torch.cat(srcs), torch.cat(tgts) + first.to(pred.device)
    flat = FlatMatches(image, query, None, slot, packed, int(pred.shape[1]))
    flat.foreign_targets = targets
    matcher.register_flat_view(indices, flat)
    return flat


def attach(criterion: torch.nn.Module, keys: Sequence[str] = MATCHED_KEYS, aux_heads: bool = True,
           device_indices: bool = False, fused_consumers: bool = False) -> HungarianMatcher:
    """Make `criterion` (an instance of one of the reference's Criterion classes, or anything that keeps its
    matcher in ``.matcher`` and is called as ``criterion(outputs, targets)``) use the gfx950 matcher, all sets
    of a step matched in one call.  `aux_heads`: also swap the ATSS / Faster-RCNN assigners a Co-DETR criterion keeps
    in its ``aux_heads`` dict.  `device_indices`: the loop is handed device-resident index tensors instead of the
    reference's CPU tensors -- its gathers then need no index upload (the reference's loss expressions on a
    64-image, 6-set step: 12 ms instead of 24 ms) and the host never waits for the matcher; the reference's
    exceptions for bad inputs are raised when the criterion returns (one synchronisation per step, where the
    reference's criteria synchronise anyway to log ``total_loss.item()``).  `fused_consumers` (implies device
    indices): three methods every one of the six criteria has work from the launch's flat index buffer
    whenever `indices` is a result this matcher served -- ``calc_box_loss(outputs, targets, indices)``
    (/root/reference/models/detr/loss.py:97-114) by the fused pair-loss kernel of :mod:`detr_amd.losses` (same two
    values, same gradient, O(pairs) instead of the [N, N] GIoU matrix and 2B gathers) and
    ``_get_src_permutation_idx(indices)`` (:71-75) by two ready tensors instead of B ``full_like`` calls and two
    concatenations.  Indices from elsewhere -- the denoising groups bring their own, several groups of queries per
    target, with their own targets -- are flattened once per step (`flat_of_foreign_indices`) and take the same
    routes: the box loss by the pair-list kernel (the reference's [N, N] matrix has N = groups x targets of the whole
    batch there), the class loss on the pseudo-image; anything that is not a list of per-image int64 index pairs for
    GPU predictions goes to the criterion's own methods.
    ``calc_cls_loss(outputs, targets, indices, ...)`` (:116-133) keeps the criterion's own body -- its class map, its
    classification loss, its accuracy log -- but is handed the batch as ONE pseudo-image: the packed labels as the
    only target and one (src, flat target index) pair as the only result, so that its per-image label gathers
    ``cat([t['labels'][i] for t, (_, i) in zip(targets, indices)])`` are one gather instead of B (this relies on the
    method touching `targets` and `indices` through that expression and `_get_src_permutation_idx` only, which holds
    for the six criteria of the reference).  The criterion's ``cls_criterion`` -- the reference's ``FocalLoss`` or
    ``CrossEntropy`` module (:185-213) -- gets the one-pass loss-and-gradient kernel of :mod:`detr_amd.losses` in front of
    its forward (`fuse_cls_criterion`; also for the denoising groups, which bring their own indices); a class value the
    reference's ``one_hot`` / ``cross_entropy`` would refuse is raised as RuntimeError when the criterion returns.
    A Co-DETR criterion's ``calc_cls_loss_with_ignore`` (/root/reference/models/co_detr/loss.py:143-167) computes its
    loss with the same kernel on the dense map with the ignored rows marked, in place of the boolean-mask gather (a
    host synchronisation), the one-hot tensor and the elementwise chain.
    The wrappers are instance attributes holding local functions:
    `detach` the criterion before pickling the object itself (its ``state_dict`` is not affected).  Returns the
    matcher.  Calling it twice on one criterion is harmless."""
    matcher = matcher_like(criterion.matcher)
    criterion.matcher = matcher
    heads = getattr(criterion, "aux_heads", None)
    if aux_heads and isinstance(heads, dict):
        for key in list(heads):
            heads[key] = aux_head_like(heads[key])
    if getattr(criterion, "_detr_amd_hooks", None) is not None:
        return matcher
    device_indices = device_indices or fused_consumers
    cls_devices = set()
    if fused_consumers and isinstance(getattr(criterion, "cls_criterion", None), torch.nn.Module):
        fuse_cls_criterion(criterion.cls_criterion, cls_devices)
    if fused_consumers and callable(getattr(criterion, "calc_box_loss", None)):
        from . import losses

        own_box_loss = criterion.calc_box_loss           # the criterion's bound method

        def calc_box_loss(outputs, targets, indices):
            m = criterion.matcher
            if not isinstance(m, HungarianMatcher):
                return own_box_loss(outputs, targets, indices)
            flat = m.served_flat(outputs, indices)
            if flat is not None:
                return losses.matched_box_losses([outputs], [flat])[0]
            # indices from elsewhere (the denoising groups'): the same two values from the list of pairs, O(pairs)
            pred = outputs.get("pred_boxes") if isinstance(outputs, dict) else None
            flat = flat_of_foreign_indices(m, outputs, targets, indices) if pred is not None and losses._takes_boxes(pred) else None
            if flat is None:
                return own_box_loss(outputs, targets, indices)
            cls_devices.add(pred.device)
            sums = losses.pair_list_box_loss_sums(pred, flat.packed.boxes, flat.batch_idx, flat.src_idx, flat.tgt_flat)
            n = float(max(flat.num_matched, 1))
            return sums[0] / n, sums[1] / n

        criterion.calc_box_loss = calc_box_loss       # (an instance attribute in front of the class's method)
    if fused_consumers and callable(getattr(criterion, "_get_src_permutation_idx", None)):
        own_permutation = criterion._get_src_permutation_idx

        def _get_src_permutation_idx(indices):
            m = criterion.matcher
            flat = m.served_flat_of(indices) if isinstance(m, HungarianMatcher) else None
            return own_permutation(indices) if flat is None else flat.src_permutation_idx()

        criterion._get_src_permutation_idx = _get_src_permutation_idx
        if callable(getattr(criterion, "calc_cls_loss", None)):
            own_cls_loss = criterion.calc_cls_loss

            def calc_cls_loss(outputs, targets, indices, *args, **kwargs):
                m = criterion.matcher
                flat = m.served_flat(outputs, indices) if isinstance(m, HungarianMatcher) else None
                if flat is None and isinstance(m, HungarianMatcher):
                    flat = flat_of_foreign_indices(m, outputs, targets, indices)
                if flat is None:
                    return own_cls_loss(outputs, targets, indices, *args, **kwargs)
                one_pair = [(flat.src_idx, flat.tgt_flat)]
                m.register_flat_view(one_pair, flat)        # (its permutation index is the whole batch's, see above)
                one_image = [{"labels": flat.packed.labels, "boxes": flat.packed.boxes}]
                return own_cls_loss(outputs, one_image, one_pair, *args, **kwargs)

            criterion.calc_cls_loss = calc_cls_loss

    if fused_consumers and callable(getattr(criterion, "calc_cls_loss_with_ignore", None)) and hasattr(criterion, "n_cls"):
        from . import losses

        own_with_ignore = criterion.calc_cls_loss_with_ignore

        def calc_cls_loss_with_ignore(outputs, targets, pos_indices, neg_indices):
            """Co-DETR's classification loss for the Faster-RCNN style assigner
            (/root/reference/models/co_detr/loss.py:143-167): the same dense class map -- -1 everywhere, the positives'
            labels, `n_cls` on the negatives (:153-159) -- but the loss straight on it with the ignored rows skipped by
            the kernel, instead of a boolean-mask gather of the foreground rows (a host synchronisation), a one-hot
            tensor and the elementwise chain of `sigmoid_focal_loss` (:161-166, with that function's default alpha and
            gamma, as the reference calls it)."""
            logits = outputs["pred_logits"]
            n_cls = criterion.n_cls
            device = logits.device
            if logits.dim() != 3 or logits.shape[-1] != n_cls \
                    or not losses._takes(logits, torch.empty(0, dtype=torch.int64, device=device)):
                return own_with_ignore(outputs, targets, pos_indices, neg_indices)
            n_pos = sum(int(src.shape[0]) for src, _ in pos_indices)
            target_cls = torch.full(logits.shape[:2], -1, dtype=torch.int64, device=device)
            if n_pos:
                # the positives as one (image, query, packed target) list -- the one calc_box_loss has just used for
                # the same indices -- instead of a label gather and an index upload per image
                m = criterion.matcher
                flat = flat_of_foreign_indices(m, outputs, targets, pos_indices) if isinstance(m, HungarianMatcher) else None
                if flat is not None:
                    target_cls[flat.batch_idx, flat.src_idx] = flat.packed.labels[flat.tgt_flat]
                else:
                    pos_b = torch.cat([torch.full_like(src, b) for b, (src, _) in enumerate(pos_indices)])
                    labels = torch.cat([t["labels"][i] for t, (_, i) in zip(targets, pos_indices)])
                    target_cls[pos_b.to(device), torch.cat([src for src, _ in pos_indices]).to(device)] = labels.to(device)
            neg_s = list(neg_indices)
            lens = np.asarray([int(src.shape[0]) for src in neg_s], dtype=np.int64)
            if int(lens.sum()):
                neg_q = torch.cat(neg_s)
                neg_b = torch.from_numpy(np.repeat(np.arange(len(neg_s), dtype=np.int64), lens))
                if neg_q.device.type == "cpu" and neg_q.dtype == torch.int64:
                    both = torch.stack([neg_b, neg_q]).to(device)
                    neg_b, neg_q = both[0], both[1]
                else:
                    neg_b, neg_q = neg_b.to(device), neg_q.to(device)
                target_cls[neg_b, neg_q] = n_cls
            cls_devices.add(device)
            return losses.focal_loss(logits, target_cls, max(n_pos, 1), 0.25, 2.0, ignore_index=-1)

        criterion.calc_cls_loss_with_ignore = calc_cls_loss_with_ignore

    def before(module, args, kwargs):
        outputs = args[0] if args else kwargs.get("outputs")
        targets = args[1] if len(args) > 1 else kwargs.get("targets")
        m = module.matcher
        if not isinstance(m, HungarianMatcher) or outputs is None or targets is None:
            return None
        m.prefetch(sets_the_loop_will_match(outputs, keys, last_only=not getattr(module, "aux_flag", True)), targets,
                   device_indices=device_indices, flat=fused_consumers)
        return None

    def after(module, args, kwargs, result):
        m = module.matcher
        if isinstance(m, HungarianMatcher):
            served = m._prefetched is not None
            m.clear_prefetched()
            if device_indices and served:
                m.check_status()          # the reference's exceptions for this step's inputs (synchronises)
        if cls_devices:
            from . import losses

            devices = list(cls_devices)
            cls_devices.clear()
            for d in devices:
                losses.check_cls_status(d)
        return None

    criterion._detr_amd_hooks = (criterion.register_forward_pre_hook(before, with_kwargs=True),
                                 criterion.register_forward_hook(after, with_kwargs=True, always_call=True))
    return matcher


def detach(criterion: torch.nn.Module) -> None:
    """Remove the hooks `attach` registered (the matcher stays)."""
    hooks = getattr(criterion, "_detr_amd_hooks", None)
    if hooks is not None:
        for h in hooks:
            h.remove()
        criterion._detr_amd_hooks = None
        for name in ("calc_box_loss", "_get_src_permutation_idx", "calc_cls_loss", "calc_cls_loss_with_ignore"):
            if name in criterion.__dict__:
                del criterion.__dict__[name]       # the class's own method shows again
        unfuse_cls_criterion(getattr(criterion, "cls_criterion", None))
