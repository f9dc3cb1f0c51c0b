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


def case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    return GoldenCase(os.path.join(GOLDEN_DIR, name + ".npz"))


def assignment_cost(cost_qt, src, tgt):
    """Total matched cost in fp64 on a [Q, T] block."""
    return float(np.asarray(cost_qt, dtype=np.float64)[src, tgt].sum())
