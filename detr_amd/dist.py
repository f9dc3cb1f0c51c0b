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


class PendingNumBoxes:
    """The matched-box counts of a step on their way through the all-reduce.

    ``result()`` makes the caller's current stream (or, on the CPU, the caller) wait for the
    collective and returns the [S] float32 tensor clamp(sum over ranks / world_size, min=1)."""

    def __init__(self, counts: torch.Tensor, work, world: int):
        self._counts, self._work, self._world = counts, work, world
        self._done = None

    def result(self) -> torch.Tensor:
        if self._done is None:
            if self._work is not None:
                self._work.wait()       # nccl: a stream dependency, the host does not block
            if self._world > 1:
                self._counts.div_(self._world)
            self._done = self._counts.clamp_(min=1.0)
        return self._done


def start_num_boxes_all_reduce(local_counts: Sequence[int], device, group=None) -> PendingNumBoxes:
    """Issue the step's one collective -- all-reduce(SUM) of the S matched-box counts -- without
    waiting for it (``async_op=True``: with the nccl backend RCCL runs it on its own stream).

    The counts follow from tensor shapes, so this can be called right after the targets are known,
    before the matching kernels are enqueued; ``result()`` is called where the loss needs the value."""
    device = torch.device(device)
    if device.type == "cuda":
        # pinned staging + asynchronous copy: the host does not wait for the stream
        staged = torch.tensor(list(local_counts), dtype=torch.float32, pin_memory=True)
        counts = staged.to(device, non_blocking=True)
    else:
        counts = torch.tensor(list(local_counts), dtype=torch.float32, device=device)
    work, world = None, 1
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        if world > 1:
            work = dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return PendingNumBoxes(counts, work, world)


def all_reduce_num_boxes(local_counts: Sequence[int], device, group=None) -> torch.Tensor:
    """[S] float32 tensor on ``device``: clamp(sum over ranks / world_size, min=1).
    `start_num_boxes_all_reduce(...).result()` in one call."""
    return start_num_boxes_all_reduce(local_counts, device, group).result()
