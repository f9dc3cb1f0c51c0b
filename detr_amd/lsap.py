"""Batched linear_sum_assignment on the GPU for caller-supplied cost matrices.

``batched_linear_sum_assignment([c_0, c_1, ...])`` returns, for each ``[Q_i, T_i]``
float32 device matrix, what ``scipy.optimize.linear_sum_assignment(c_i)`` returns
(the call the reference makes per image at /root/reference/models/matcher.py:94),
including scipy's choice among tied optima, using the assignment kernel alone.
"""
import ctypes
from collections import defaultdict
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import DmSet, lib
from .matcher import _raise_for_status


@torch.no_grad()
def batched_linear_sum_assignment(costs: Sequence[torch.Tensor]) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    if len(costs) == 0:
        return []
    device = costs[0].device
    if device.type != "cuda":
        raise RuntimeError("batched_linear_sum_assignment runs on the GPU only")
    groups = defaultdict(list)          # Q -> indices of the matrices with that many rows
    for i, c in enumerate(costs):
        if c.dim() != 2 or c.dtype != torch.float32 or c.device != device:
            raise ValueError("cost matrices must be 2-D float32 tensors on one device")
        groups[int(c.shape[0])].append(i)
    results = [None] * len(costs)
    empty = torch.zeros(0, dtype=torch.int64)
    stream = torch.cuda.current_stream(device).cuda_stream
    pending = []
    for Q, idxs in groups.items():
        sizes = [int(costs[i].shape[1]) for i in idxs]
        M, B = sum(sizes), len(idxs)
        if Q == 0 or M == 0:
            for i in idxs:
                results[i] = (empty, empty)
            continue
        stride = (Q + 31) & ~31
        block = torch.zeros(M, stride, dtype=torch.float32, device=device)
        # rows are targets: the kernels hold the transpose of scipy's [Q, T] argument
        block[:, :Q] = torch.cat([costs[i].t() for i in idxs if costs[i].shape[1] > 0], dim=0)
        meta = np.zeros(2 * B + 1, dtype=np.int32)
        np.cumsum(sizes, out=meta[1:B + 1])
        meta[B + 1:] = np.argsort(-np.asarray(sizes), kind="stable")
        meta_dev = torch.from_numpy(meta).to(device)
        out = torch.empty(2 * M + 1, dtype=torch.int64, device=device)
        out[-1:].zero_()
        sets = (DmSet * 1)()
        sets[0].logits, sets[0].boxes = None, None
        sets[0].cost_off, sets[0].out_off, sets[0].Q, sets[0].q_stride = 0, 0, Q, stride
        with torch.cuda.device(device):
            rc = lib.dm_lap_f32(sets, 1, B, block.data_ptr(), meta_dev.data_ptr(),
                                meta_dev.data_ptr() + 4 * (B + 1), max(sizes), out.data_ptr(),
                                out.data_ptr() + 8 * M, out.data_ptr() + 16 * M, stream)
        _lib.check(rc, "dm_lap_f32")
        pending.append((Q, idxs, sizes, meta, out, block, meta_dev))
    for Q, idxs, sizes, meta, out, _, _ in pending:
        host = out.cpu()
        _raise_for_status(int(host[-1].item()) & 0xFFFFFFFF)
        M = sum(sizes)
        for k, i in enumerate(idxs):
            n, lo = min(Q, sizes[k]), int(meta[k])
            results[i] = (host[lo:lo + n].clone(), host[M + lo:M + lo + n].clone())
    return results
