"""CPU restatement (numpy) of the reference's classification loss on the dense class map -- TEST INFRASTRUCTURE
ONLY: never imported by the product path.

Follows /root/reference/models/detr/loss.py:
  focal_loss      FocalLoss.forward :208-213 with sigmoid_focal_loss :217-242 (float32, the reference's operation
                  order; `loss.mean(1).sum() / num_boxes * Q` is the sum over all logits / num_boxes)
  cross_entropy   CrossEntropy.forward :193-195 = nn.CrossEntropyLoss(weight), reduction 'mean'
and /root/reference/models/co_detr/loss.py:153-166 for `ignore_index` in focal_loss (rows of class -1 are left out
before the loss).  Gradients are the analytic derivatives of those operations (what autograd returns for a unit
upstream gradient).

Pinned by tests/golden/cls/*.npz, produced by running the reference's own modules.
"""
import numpy as np

F = np.float32


def _softplus(z):
    return np.maximum(z, F(0)) + np.log1p(np.exp(-np.abs(z)))


def focal_loss(logits, target_cls, num_boxes, alpha=0.25, gamma=2.0, ignore_index=None):
    """logits [..., C] float32, target_cls [...] int64 (C = no object).  Returns (loss, grad [..., C])."""
    x = np.asarray(logits, dtype=F)
    C = x.shape[-1]
    cls = np.asarray(target_cls, dtype=np.int64)
    keep = np.ones(cls.shape, dtype=bool) if ignore_index is None else cls != ignore_index
    if ((cls[keep] < 0) | (cls[keep] > C)).any():
        raise RuntimeError("class values must lie in [0, C]")
    t = (cls[..., None] == np.arange(C)).astype(F)                 # one_hot(targets, C + 1)[..., :-1]
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        p = (F(1) / (F(1) + np.exp(-x))).astype(F)                  # :233
        ce = _softplus(np.where(t > 0, -x, x)).astype(F)            # :234
        p_t = p * t + (F(1) - p) * (F(1) - t)                       # :235
        base = F(1) - p_t
        m = (base * base if gamma == 2.0 else np.power(base, F(gamma))).astype(F)
        loss = ce * m                                               # :236
        a_t = (F(alpha) * t + (F(1) - F(alpha)) * (F(1) - t)) if alpha >= 0 else np.ones_like(t)      # :238-240
        loss = a_t * loss
        if gamma == 2.0:
            dm = F(2) * base
        elif gamma == 0.0:
            dm = np.zeros_like(base)
        else:
            dm = F(gamma) * np.power(base, F(gamma - 1.0))
        dbase = -(F(2) * t - F(1)) * p * (F(1) - p)
        grad = a_t * ((p - t) * m + ce * dm * dbase)
    keep3 = keep[..., None]
    total = np.where(keep3, loss, F(0)).astype(np.float64).sum()
    return F(total / float(num_boxes)), (np.where(keep3, grad, F(0)) / F(num_boxes)).astype(F)


def cross_entropy(logits, target_cls, weight=None, ignore_index=-100):
    """Returns (loss, grad [..., C])."""
    x = np.asarray(logits, dtype=F)
    C = x.shape[-1]
    rows = x.reshape(-1, C)
    y = np.asarray(target_cls, dtype=np.int64).reshape(-1)
    keep = y != ignore_index
    if ((y[keep] < 0) | (y[keep] >= C)).any():
        raise RuntimeError("class values must lie in [0, C)")
    w = np.ones(C, dtype=F) if weight is None else np.asarray(weight, dtype=F)
    ys = np.where(keep, y, 0)
    wy = np.where(keep, w[ys], F(0)).astype(F)
    mx = rows.max(axis=1, keepdims=True)
    lse = (mx[:, 0] + np.log(np.exp(rows - mx).sum(axis=1, dtype=F))).astype(F)
    nll = lse - rows[np.arange(rows.shape[0]), ys]
    den = wy.astype(np.float64).sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        loss = F((wy.astype(np.float64) * nll).sum() / den)
        soft = np.exp(rows - lse[:, None]).astype(F)
        soft[np.arange(rows.shape[0]), ys] -= F(1)
        grad = (wy[:, None] * soft / F(den)).astype(F)
    return loss, grad.reshape(x.shape)
