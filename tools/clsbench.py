#!/usr/bin/env python3
"""Time of the classification-loss kernels (dm_focal_loss_f32 / dm_cross_entropy_f32, called through
detr_amd.losses) on one output set of a workload's shape, against their HBM traffic (logits read once, gradient
written once, 8 bytes of class map per row) and against the reference's expressions with stock torch ops
(/root/reference/models/detr/loss.py:185-242), forward + backward.

    python tools/clsbench.py [B Q C]      (default: the cfg2 and cfg4 shapes, focal and cross entropy)
"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from detr_amd import losses


def events(fn, n=100):
    for _ in range(10):
        fn()
    gc.collect()          # (a full collection in the middle of the loop would stall the launches for tens of ms)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def torch_focal(x, cls, n_cls, nb):
    onehot = F.one_hot(cls, n_cls + 1)[..., :-1].float()
    p = x.sigmoid()
    ce = F.binary_cross_entropy_with_logits(x, onehot, reduction="none")
    p_t = p * onehot + (1 - p) * (1 - onehot)
    return ((0.25 * onehot + 0.75 * (1 - onehot)) * ce * (1 - p_t) ** 2.0).mean(1).sum() / nb * x.shape[1]


def run(B, Q, C):
    gen = torch.Generator().manual_seed(1)
    x = (torch.randn(B, Q, C, generator=gen) * 2).cuda().requires_grad_(True)
    cls = torch.full((B, Q), C, dtype=torch.int64)
    pick = torch.rand(B, Q, generator=gen) < 0.03
    cls[pick] = torch.randint(0, C, (int(pick.sum()),), generator=gen)
    cls, nb = cls.cuda(), max(int(pick.sum()), 1)
    n = B * Q * C
    for what, fwd, bytes_fwd in (
            ("focal, loss only", lambda: losses.focal_loss(x.detach(), cls, nb), 4 * n + 8 * B * Q),
            ("focal, loss + gradient", lambda: losses.focal_loss(x, cls, nb), 8 * n + 8 * B * Q)):
        us = events(fwd)
        print(f"[{B},{Q},{C}] {what:28s} {us:8.1f} us  {bytes_fwd / us / 1e3:7.1f} GB/s  ({bytes_fwd / 1e6:.1f} MB)")

    def both(loss_fn):
        x.grad = None
        loss_fn().backward()

    print(f"[{B},{Q},{C}] focal fwd+bwd through autograd: fused {events(lambda: both(lambda: losses.focal_loss(x, cls, nb)), 50):8.1f} us,"
          f" stock torch ops {events(lambda: both(lambda: torch_focal(x, cls, C, nb)), 50):8.1f} us")
    # cross entropy: the same logits read as C - 1 classes + no object
    w = torch.ones(C, device="cuda")
    w[-1] = 0.1
    ycls = cls.clamp(max=C - 1)
    us = events(lambda: losses.cross_entropy(x, ycls, w))
    print(f"[{B},{Q},{C}] cross entropy, loss + gradient {us:8.1f} us  {(8 * n + 8 * B * Q) / us / 1e3:7.1f} GB/s")
    print(f"[{B},{Q},{C}] cross entropy fwd+bwd through autograd: fused {events(lambda: both(lambda: losses.cross_entropy(x, ycls, w)), 50):8.1f} us,"
          f" stock torch ops {events(lambda: both(lambda: F.cross_entropy(x.transpose(1, 2), ycls, weight=w)), 50):8.1f} us")


if len(sys.argv) == 4:
    run(*(int(a) for a in sys.argv[1:]))
else:
    for shape in ((64, 600, 2), (64, 600, 3), (256, 900, 91), (256, 900, 92)):
        run(*shape)
