"""Data-parallel glue for the matching path: one process per GPU, images sharded by rank.

Matching needs no exchange between ranks -- every (output set, image) problem depends on
that image alone (/root/reference/models/matcher.py:94 uses block i only).  The one
collective of the step is the normalisation constant of the losses: the reference
divides by the local ``max(len(idx[0]), 1)`` (/root/reference/models/detr/loss.py:100,119);
under data parallelism the matched-box counts of all output sets are summed over ranks in
a single all-reduce (RCCL over xGMI when the backend is "nccl") and divided by the world
size, so that averaging gradients over ranks normalises by the global count.
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


def matched_box_counts(target_sizes: Sequence[int], set_queries: Sequence[int]) -> List[int]:
    """Pairs the matcher returns per output set on this rank: sum_b min(Q_s, T_b).

    Known from tensor shapes alone, i.e. before any kernel has run."""
    return [sum(min(q, t) for t in target_sizes) for q in set_queries]


def shard_range(n_images: int, rank: int, world_size: int):
    """Contiguous image range [lo, hi) owned by ``rank``."""
    base, extra = divmod(n_images, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def balanced_shards(target_sizes: Sequence[int], world_size: int) -> List[List[int]]:
    """Image indices per rank when the caller controls placement: images sorted by target count,
    dealt to the ranks in boustrophedon order (0..G-1, G-1..0, ...), so every rank gets the same
    number of images (+-1) and nearly the same number of targets -- the assignment kernel's time on
    a rank is set by its heaviest images."""
    order = sorted(range(len(target_sizes)), key=lambda i: (-int(target_sizes[i]), i))
    shards = [[] for _ in range(world_size)]
    for pos, img in enumerate(order):
        lap, k = divmod(pos, world_size)
        shards[k if lap % 2 == 0 else world_size - 1 - k].append(img)
    return [sorted(s) for s in shards]


_side_streams = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    """One stream per device for the step's collective: nothing on it is ordered against the matching kernels."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    stream = _side_streams.get(key)
    if stream is None:
        stream = _side_streams[key] = torch.cuda.Stream(device=device)
    return stream


class PendingNumBoxes:
    """The matched-box counts of a step on their way through the all-reduce.

    ``result()`` makes the caller's current stream (or, on the CPU, the caller) wait for the
    collective and returns the [S] float32 tensor clamp(sum over ranks / world_size, min=1)."""

    def __init__(self, counts: torch.Tensor, work=None, side=None):
        self._counts, self._work, self._side = counts, work, side
        self._done = None

    def result(self) -> torch.Tensor:
        if self._done is None:
            if self._side is not None:
                # the collective and the clamp were enqueued on the side stream: the caller's stream waits for
                # their event here (no host wait), and the tensor is marked as used on it
                slot, generation, current = self._side
                if slot.generation == generation:
                    current.wait_event(slot.event)
                # (else the slot has been handed out again since, which only happens after the host has seen this
                # step's event complete: there is nothing left to wait for)
                self._counts.record_stream(current)
                self._done = self._counts
            else:
                if self._work is not None:
                    self._work.wait()
                self._done = self._counts.clamp_(min=1.0)
        return self._done


class _Slot:
    """One step's pinned staging buffer and completion event.  A slot is handed out again only after the host
    has seen its event complete -- the upload out of the buffer has run by then, however late a peer rank made
    the collective behind it -- so a caller that never consumes `result()` (device-resident loops that do not
    synchronise) can run any number of steps ahead without a buffer being overwritten under a pending copy."""

    __slots__ = ("event", "staged", "generation", "recorded")

    def __init__(self):
        self.event = torch.cuda.Event()
        self.staged = None
        self.generation = 0
        self.recorded = False


_RING = 8
_slot_rings = {}


def _next_slot(device: torch.device, n: int) -> _Slot:
    key = device.index if device.index is not None else torch.cuda.current_device()
    ring = _slot_rings.get(key)
    if ring is None:
        with torch.cuda.device(device):
            ring = _slot_rings[key] = [[_Slot() for _ in range(_RING)], 0]
    ring[1] = (ring[1] + 1) % _RING
    slot = ring[0][ring[1]]
    if slot.recorded and not slot.event.query():
        slot.event.synchronize()          # the step that used this slot eight steps ago is still in flight
    slot.generation += 1
    if slot.staged is None or slot.staged.shape[0] != n:
        slot.staged = torch.empty(n, dtype=torch.float32, pin_memory=True)
    return slot


def start_num_boxes_all_reduce(local_counts: Sequence[int], device, group=None, _always_reduce: bool = False) -> PendingNumBoxes:
    """Issue the step's one collective -- all-reduce(SUM) of the S matched-box counts -- without
    waiting for it.

    On a GPU everything (the upload of the counts, the all-reduce -- RCCL with the nccl backend --, the clamp)
    is enqueued on a stream of its own, so that none of it is ordered against the matching kernels on the
    caller's stream: issued right after the targets are known, or from
    ``HungarianMatcher.match_many(..., while_waiting=...)`` where also its host cost disappears behind the
    kernels, it runs beside them.  ``result()`` is called where the loss needs the value; only then does the
    caller's stream wait (cross-stream dependencies cost 5-10 us of GPU time each: on the caller's stream the
    same collective stretches a 75 us step to 150 us).
    The division by the world size is applied to the local counts on the host (one device operation less
    per step; exact for world sizes that are powers of two, within one rounding otherwise)."""
    device = torch.device(device)
    world = 1
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
    reduce = world > 1 or (_always_reduce and dist.is_available() and dist.is_initialized())
    scaled = [c / world for c in local_counts] if world > 1 else list(local_counts)
    if device.type != "cuda":
        counts = torch.tensor(scaled, dtype=torch.float32, device=device)
        work = dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group, async_op=True) if reduce else None
        return PendingNumBoxes(counts, work)
    side = _side_stream(device)
    slot = _next_slot(device, len(scaled))
    slot.staged.numpy()[:] = scaled
    # (set_stream twice instead of the `torch.cuda.stream` context manager: a quarter of its cost)
    current = torch.cuda.current_stream(device)
    torch.cuda.set_stream(side)
    try:
        counts = slot.staged.to(device, non_blocking=True)        # pinned -> device, asynchronous
        if reduce:
            # The plain form: with the nccl backend it enqueues the collective on RCCL's stream and makes the
            # current stream -- the side stream here -- wait for it; the host does not block.  (async_op=True
            # followed by Work.wait() does the same for 35 us of host time instead of 8.)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        counts.clamp_(min=1.0)
        slot.event.record(side)
        slot.recorded = True
    finally:
        torch.cuda.set_stream(current)
    return PendingNumBoxes(counts, None, (slot, slot.generation, current))


def all_reduce_num_boxes(local_counts: Sequence[int], device, group=None) -> torch.Tensor:
    """[S] float32 tensor on ``device``: clamp(sum over ranks / world_size, min=1).
    `start_num_boxes_all_reduce(...).result()` in one call."""
    return start_num_boxes_all_reduce(local_counts, device, group).result()
