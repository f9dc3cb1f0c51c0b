"""Loader and comparison helpers for the auxiliary-assigner golden vectors (tests/golden/aux/)."""
import os

import numpy as np

import golden_io

AUX_DIR = os.path.join(golden_io.GOLDEN_DIR, "aux")
AUX_CASES = sorted(os.path.splitext(f)[0] for f in os.listdir(AUX_DIR) if f.endswith(".npz"))


class AuxCase:
    def __init__(self, name):
        z = np.load(os.path.join(AUX_DIR, name + ".npz"), allow_pickle=False)
        self.pred_boxes = z["pred_boxes"]
        self.sizes = z["sizes"].tolist()
        self.k = int(z["k"])
        self.B, self.Q = self.pred_boxes.shape[:2]
        off = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.tboxes = [z["tboxes"][off[b]:off[b + 1]].reshape(-1, 4) for b in range(self.B)]
        self.atss = self._split_pairs(z["atss_src"], z["atss_tgt"], z["atss_counts"])
        self.pos = self._split_pairs(z["pos_src"], z["pos_tgt"], z["pos_counts"])
        self.neg = np.split(z["neg"], np.cumsum(z["neg_counts"])[:-1])
        doff = np.concatenate([[0], np.cumsum([t * self.Q for t in self.sizes])]).astype(np.int64)
        self.ref_dist = [z["ref_dist"][doff[b]:doff[b + 1]].reshape(self.sizes[b], self.Q) for b in range(self.B)]

    @staticmethod
    def _split_pairs(src, tgt, counts):
        cut = np.cumsum(counts)[:-1]
        return list(zip(np.split(src, cut), np.split(tgt, cut)))

    def torch_inputs(self, device="cpu"):
        import torch

        outputs = {"pred_boxes": torch.from_numpy(self.pred_boxes).to(device)}
        targets = [{"boxes": torch.from_numpy(np.ascontiguousarray(b)).to(device),
                    "labels": torch.zeros(len(b), dtype=torch.int64, device=device)} for b in self.tboxes]
        return outputs, targets


def pairs_by_target(src, tgt, n_targets):
    return [sorted(np.asarray(src)[np.asarray(tgt) == t].tolist()) for t in range(n_targets)]


def assert_atss_equal(case, got, image):
    """Pairs equal to the reference's, in its order (target-major, candidates by distance).  A target
    may differ only if the reference's own distances put two candidates of that target closer
    together than the rounding noise of its matmul-based cdist (1e-6 absolute on distances <= 1.5)."""
    rs, rt = case.atss[image]
    gs, gt = np.asarray(got[0]), np.asarray(got[1])
    if np.array_equal(gs, rs) and np.array_equal(gt, rt):
        return
    for t in range(case.sizes[image]):
        a, b = gs[gt == t].tolist(), rs[rt == t].tolist()
        if a == b:
            continue
        d = np.sort(case.ref_dist[image][t])[:case.k + 1]
        gaps = np.diff(d)
        assert gaps.min() < 1e-6, f"image {image} target {t}: {a} != {b} without a near-tie in the distances"


def assert_rcnn_equal(case, got_pos, got_neg, image):
    rs, rt = case.pos[image]
    T = case.sizes[image]
    assert pairs_by_target(got_pos[0], got_pos[1], T) == pairs_by_target(rs, rt, T)
    assert np.array_equal(np.asarray(got_pos[1]), np.sort(np.asarray(got_pos[1])))      # target-major order
    assert np.array_equal(np.asarray(got_neg), case.neg[image])
