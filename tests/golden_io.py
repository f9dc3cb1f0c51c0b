"""Loader for the committed golden vectors (tests/golden/*.npz, made by gen_golden.py)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_EXC = {"ValueError": ValueError, "AssertionError": AssertionError, "IndexError": IndexError}


class GoldenCase:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.name = os.path.splitext(os.path.basename(path))[0]
        meta = json.loads(str(z["meta"]))
        self.cls_loss = meta["cls_loss"]
        self.weights = tuple(meta["weights"])
        self.alpha, self.gamma = meta["alpha"], meta["gamma"]
        self.error = _EXC[meta["error"]] if meta["error"] else None
        self.logits, self.pboxes = z["logits"], z["pboxes"]
        self.sizes = z["sizes"].tolist()
        self.B, self.Q, self.C = self.logits.shape
        off = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.labels = [z["labels"][off[b]:off[b + 1]] for b in range(self.B)]
        self.tboxes = [z["tboxes"][off[b]:off[b + 1]].reshape(-1, 4) for b in range(self.B)]
        self.cost, self.indices = None, None
        if self.error is None:
            # the reference calls scipy for every image, also those with T_b = 0
            cost, pos = [], 0
            for t in self.sizes:
                cost.append(z["cost"][pos:pos + self.Q * t].reshape(self.Q, t))
                pos += self.Q * t
            self.cost = cost
            idx, pos = [], 0
            for t in self.sizes:
                n = min(self.Q, t)
                idx.append((z["src"][pos:pos + n], z["tgt"][pos:pos + n]))
                pos += n
            self.indices = idx

    def matcher_kwargs(self):
        return dict(cls_loss=self.cls_loss, cost_class=self.weights[0], cost_bbox=self.weights[1],
                    cost_giou=self.weights[2], focal_alpha=self.alpha, focal_gamma=self.gamma)

    def torch_inputs(self, device="cpu"):
        import torch

        outputs = {"pred_logits": torch.from_numpy(self.logits).to(device),
                   "pred_boxes": torch.from_numpy(self.pboxes).to(device)}
        targets = [{"labels": torch.from_numpy(np.ascontiguousarray(l)).to(device),
                    "boxes": torch.from_numpy(np.ascontiguousarray(b)).to(device)}
                   for l, b in zip(self.labels, self.tboxes)]
        return outputs, targets


DTYPE_DIR = os.path.join(GOLDEN_DIR, "dtype")


class DtypeCase:
    """A golden case of tests/golden/dtype (made by gen_golden_dtypes.py): inputs in their own dtypes, the
    reference's float64 cost blocks and indices, or the type name of the exception it raised."""

    def __init__(self, name):
        import torch

        z = np.load(os.path.join(DTYPE_DIR, name + ".npz"), allow_pickle=False)
        meta = json.loads(str(z["meta"]))
        self.name, self.cls_loss, self.weights = name, meta["cls_loss"], tuple(meta["weights"])
        self.error = {"RuntimeError": RuntimeError, "NotImplementedError": NotImplementedError}.get(meta["error"])
        dts = [getattr(torch, d.split(".")[1]) for d in meta["dtypes"]]
        self.sizes = z["sizes"].tolist()
        self.logits = torch.from_numpy(z["logits"]).to(dts[0])
        self.pboxes = torch.from_numpy(z["pboxes"]).to(dts[1])
        self.B, self.Q, self.C = self.logits.shape
        off = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.targets = [{"labels": torch.from_numpy(z["labels"][off[b]:off[b + 1]].copy()),
                         "boxes": torch.from_numpy(z["tboxes"][off[b]:off[b + 1]].reshape(-1, 4).copy()).to(dts[2])}
                        for b in range(self.B)]
        self.cost = self.indices = None
        if self.error is None:
            self.cost, self.indices, pos, ipos = [], [], 0, 0
            for t in self.sizes:
                self.cost.append(z["cost"][pos:pos + self.Q * t].reshape(self.Q, t))
                pos += self.Q * t
                n = min(self.Q, t)
                self.indices.append((z["src"][ipos:ipos + n], z["tgt"][ipos:ipos + n]))
                ipos += n

    def matcher_kwargs(self):
        return dict(cls_loss=self.cls_loss, cost_class=self.weights[0], cost_bbox=self.weights[1],
                    cost_giou=self.weights[2])


CLS_DIR = os.path.join(GOLDEN_DIR, "cls")


class ClsLossCase:
    """A golden case of tests/golden/cls (made by gen_golden_cls.py): class-last logits [B, Q, C], the dense class map,
    the parameters of the reference's FocalLoss / CrossEntropy, its loss and the gradient w.r.t. the logits."""

    def __init__(self, name):
        z = np.load(os.path.join(CLS_DIR, name + ".npz"), allow_pickle=False)
        self.name, self.kind, self.n_cls = name, str(z["kind"]), int(z["n_cls"])
        self.loss, self.grad = float(z["loss"]), z["grad"]
        if "from_matcher_case" in z.files:          # inputs live in the matcher golden and its loss golden
            src = str(z["from_matcher_case"])
            self.logits = load(src).logits
            lz = np.load(os.path.join(GOLDEN_DIR, "loss", src + ".npz"), allow_pickle=False)
            self.target_cls, self.num_boxes = lz["target_cls"], int(lz["num_boxes"])
        else:
            self.logits, self.target_cls = z["logits"], z["target_cls"]
            self.num_boxes = int(z["num_boxes"]) if "num_boxes" in z.files else 1
        self.alpha = float(z["alpha"]) if "alpha" in z.files else None
        self.gamma = float(z["gamma"]) if "gamma" in z.files else None
        self.weight = z["weight"] if "weight" in z.files else None
        self.ignore_index = int(z["ignore_index"]) if "ignore_index" in z.files else None


def cls_case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(CLS_DIR, "*.npz")))


PAIRS_DIR = os.path.join(GOLDEN_DIR, "pairs")


class PairListCase:
    """A golden case of tests/golden/pairs (made by gen_golden_pairs.py): prediction boxes, per-image target boxes and
    per-image (query, target) index pairs that did not come from the matcher; the reference's l1 / giou loss and their
    gradients w.r.t. the prediction boxes."""

    def __init__(self, name):
        z = np.load(os.path.join(PAIRS_DIR, name + ".npz"), allow_pickle=False)
        self.name, self.pred_boxes = name, z["pred_boxes"]
        t_off = np.concatenate([[0], np.cumsum(z["tsizes"])])
        i_off = np.concatenate([[0], np.cumsum(z["isizes"])])
        B = len(z["tsizes"])
        self.target_boxes = [z["tboxes"][t_off[b]:t_off[b + 1]].reshape(-1, 4) for b in range(B)]
        self.indices = [(z["src"][i_off[b]:i_off[b + 1]], z["tgt"][i_off[b]:i_off[b + 1]]) for b in range(B)]
        self.l1, self.giou = float(z["l1"]), float(z["giou"])
        self.grad_l1, self.grad_giou = z["grad_l1"], z["grad_giou"]


def pair_case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(PAIRS_DIR, "*.npz")))


def dtype_case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(DTYPE_DIR, "*.npz")))


def case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    return GoldenCase(os.path.join(GOLDEN_DIR, name + ".npz"))


def assignment_cost(cost_qt, src, tgt):
    """Total matched cost in fp64 on a [Q, T] block."""
    return float(np.asarray(cost_qt, dtype=np.float64)[src, tgt].sum())


# --------------------------------------------------------------------------------------
# cost tolerance
# --------------------------------------------------------------------------------------
# 1e-5 relative to max(|cost|, 1): the terms of an entry (5 * L1, 2 * class, 2 * GIoU, each of order one to
# ten) can cancel to a total near zero, and two correct fp32 evaluations then differ by ulps of the terms, not
# of the total (observed between the reference's torch ops and a plain C restatement: 5.7e-6 on a total of 0.06)
COST_RTOL, COST_ATOL = 1e-5, 1e-5


def _focal64_from_p(p, alpha, gamma):
    neg = (1 - alpha) * p ** gamma * (-np.log(1 - p + 1e-8))
    pos = alpha * (1 - p) ** gamma * (-np.log(p + 1e-8))
    return pos - neg


def focal_sensitivity(logits_qc, labels, alpha, gamma, ulps=2.0):
    """[Q, T] change of the focal class term (matcher.py:76-81) when sigmoid(x) moves by
    +-`ulps` fp32 ulps.  The term contains log(1 - p): for p -> 1 one ulp of p is a large
    relative change of 1 - p, so two correct fp32 evaluations of the reference's formula
    (different exp implementations) legitimately differ by this much."""
    x = np.asarray(logits_qc, dtype=np.float64)[:, np.asarray(labels, dtype=np.int64)]
    p = 1.0 / (1.0 + np.exp(-x))
    eps = ulps * 2.0 ** -24
    hi = np.minimum(p * (1 + eps), 1.0)
    lo = p * (1 - eps)
    return np.abs(_focal64_from_p(hi, alpha, gamma) - _focal64_from_p(lo, alpha, gamma))


# |x| beyond which the last bits of sigmoid(x) can move the focal class term past the plain bar: one ulp of p
# (6e-8) changes cost_class * (1 - alpha) * p^2 * log(1 - p + 1e-8) by 2 * 0.75 * 6e-8 * e^x = 4.9e-6 at x = 4
# and 1.8e-6 at x = 3, and two correct fp32 evaluations of sigmoid (torch's vectorised CPU kernel, libm, the
# GPU's exp / reciprocal instructions) differ by up to 4 ulp from one another
SENSITIVE_LOGIT = 3.0


def assert_cost_close(got_qt, ref_qt, logits_qc, labels, cls_loss, w_class, alpha, gamma, stats=None):
    """The north-star bar -- 1e-5 relative in fp32.  Only for focal entries whose logit is saturated
    (|x| > 3) the bar is widened by the class term's own sensitivity to a 2-ulp change of sigmoid(x).
    Returns the share of ALL entries inside the plain bar; `stats` (a dict) receives that share, the
    worst relative error among the unsaturated entries and the number of saturated ones."""
    got = np.asarray(got_qt, dtype=np.float64)
    ref = np.asarray(ref_qt, dtype=np.float64)
    assert got.shape == ref.shape
    plain = COST_RTOL * np.abs(ref) + COST_ATOL
    tol = plain
    saturated = np.zeros(ref.shape, dtype=bool)
    if cls_loss != "ce" and ref.size:
        x = np.asarray(logits_qc, dtype=np.float64)[:, np.asarray(labels, dtype=np.int64)]
        saturated = np.abs(x) > SENSITIVE_LOGIT
        tol = plain + np.where(saturated, abs(w_class) * focal_sensitivity(logits_qc, labels, alpha, gamma), 0.0)
    err = np.abs(got - ref)
    bad = ~(err <= tol)
    assert not bad.any(), "cost mismatch: %d of %d entries, worst |err| %.3e (tol %.3e)" % (
        bad.sum(), bad.size, err[bad].max(), tol[bad][np.argmax(err[bad])])
    strict = err <= plain
    share = float(strict.mean()) if ref.size else 1.0
    if stats is not None:
        rel = err / np.maximum(np.abs(ref), COST_ATOL / COST_RTOL)
        stats["entries"] = stats.get("entries", 0) + int(ref.size)
        stats["inside_plain_bar"] = stats.get("inside_plain_bar", 0) + int(strict.sum())
        stats["saturated_entries"] = stats.get("saturated_entries", 0) + int(saturated.sum())
        unsat = rel[~saturated]
        stats["worst_rel_err_unsaturated"] = max(stats.get("worst_rel_err_unsaturated", 0.0),
                                                 float(unsat.max()) if unsat.size else 0.0)
        stats["worst_rel_err_all"] = max(stats.get("worst_rel_err_all", 0.0), float(rel.max()) if rel.size else 0.0)
    return share
