"""Fused consumers of the indices (detr_amd.losses) against the numpy restatement of the
reference's criterion steps (oracle/pair_loss.py, itself pinned by tests/golden/loss/) and
against the golden vectors themselves wherever the GPU picked the same pairs.

Tolerances: the two loss scalars 1e-5 relative (fp32, different summation order), gradients
rtol 1e-5 / atol 1e-6, class maps bit-exact."""
import os

import numpy as np
import pytest
import torch

import golden_io

pytestmark = pytest.mark.gpu

LOSS_DIR = os.path.join(golden_io.GOLDEN_DIR, "loss")
LOSS_CASES = sorted(os.path.splitext(f)[0] for f in os.listdir(LOSS_DIR) if f.endswith(".npz"))


def _close(a, b, rtol=1e-5):
    return abs(float(a) - float(b)) <= rtol * max(1.0, abs(float(b)))


@pytest.mark.parametrize("name", LOSS_CASES)
def test_golden_cases(name):
    import detr_amd
    from detr_amd import losses
    from oracle import pair_loss

    g = golden_io.load(name)
    z = np.load(os.path.join(LOSS_DIR, name + ".npz"), allow_pickle=False)
    outputs, targets = g.torch_inputs("cuda")
    outputs["pred_boxes"].requires_grad_(True)
    matcher = detr_amd.HungarianMatcher(**g.matcher_kwargs())
    flats = matcher.match_flat([outputs], targets)
    tuples = [(i.numpy(), j.numpy()) for i, j in matcher(outputs, targets)]
    (l1, giou), = losses.matched_box_losses([outputs], flats)
    ref_l1, ref_giou, ref_g1, ref_g2 = pair_loss.box_losses(g.pboxes, g.tboxes, tuples)
    assert _close(l1.item(), ref_l1) and _close(giou.item(), ref_giou)
    g1, = torch.autograd.grad(l1, outputs["pred_boxes"], retain_graph=True)
    g2, = torch.autograd.grad(giou, outputs["pred_boxes"])
    np.testing.assert_allclose(g1.cpu().numpy(), ref_g1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g2.cpu().numpy(), ref_g2, rtol=1e-5, atol=1e-6)
    n_cls = int(z["n_cls"])
    dense, = losses.target_class_maps(flats, n_cls)
    np.testing.assert_array_equal(dense.cpu().numpy(), pair_loss.target_classes((g.B, g.Q), g.labels, tuples, n_cls))
    same_pairs = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(tuples, g.indices))
    if same_pairs:      # then the reference's own numbers apply directly
        assert _close(l1.item(), z["l1"]) and _close(giou.item(), z["giou"])
        np.testing.assert_allclose(g1.cpu().numpy(), z["grad_l1"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g2.cpu().numpy(), z["grad_giou"], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(dense.cpu().numpy(), z["target_cls"])
    else:
        assert name in ("dup_gt", "identical_preds", "blocks_of_equal_preds")   # tie cases only


def _torch_box_loss(pred_boxes, targets, indices):
    """The reference's calc_box_loss expression (loss.py:97-114) with stock torch ops, fp32."""
    bi = torch.cat([torch.full_like(s, b) for b, (s, _) in enumerate(indices)])
    si = torch.cat([s for s, _ in indices])
    n = max(len(bi), 1)
    sel = pred_boxes[bi, si]
    tgt = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)])
    l1 = (sel - tgt).abs().sum() / n

    def xyxy(b):
        cx, cy, w, h = b.unbind(-1)
        return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)

    a, g = xyxy(sel), xyxy(tgt)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_g = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    wh = (torch.min(a[:, 2:], g[:, 2:]) - torch.max(a[:, :2], g[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area_a + area_g - inter
    ewh = (torch.max(a[:, 2:], g[:, 2:]) - torch.min(a[:, :2], g[:, :2])).clamp(min=0)
    earea = ewh[:, 0] * ewh[:, 1]
    giou = inter / union - (earea - union) / earea
    return l1, (1 - giou).sum() / n


@pytest.mark.parametrize("name,batch", [("cfg2", 16), ("cfg4", 8), ("cfg3_hw", 4), ("cfg5", 6)])
def test_all_sets_of_a_step_against_torch_autograd(name, batch):
    import detr_amd
    from detr_amd import losses, synth

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    for o in sets:
        o["pred_boxes"].requires_grad_(True)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    flats = matcher.match_flat(sets, targets)
    tuples = matcher.match_many(sets, targets)
    got = losses.matched_box_losses(sets, flats)
    total = sum(5.0 * l1 + 2.0 * giou for l1, giou in got)
    grads = torch.autograd.grad(total, [o["pred_boxes"] for o in sets])
    for s, o in enumerate(sets):
        idx = [(i.cuda(), j.cuda()) for i, j in tuples[s]]
        ref_boxes = o["pred_boxes"].detach().clone().requires_grad_(True)
        l1, giou = _torch_box_loss(ref_boxes, targets, idx)
        assert _close(got[s][0].item(), l1.item()) and _close(got[s][1].item(), giou.item())
        ref_grad, = torch.autograd.grad(5.0 * l1 + 2.0 * giou, ref_boxes)
        torch.testing.assert_close(grads[s], ref_grad, rtol=1e-5, atol=1e-6)
    # the weighted total in one scalar gives the same value and gradients
    for o in sets:
        o["pred_boxes"].grad = None
    mean = losses.matched_box_loss_total(sets, flats, 5.0, 2.0)      # the reference's mean over the layers
    assert _close(mean.item(), total.item() / len(sets))
    one = losses.matched_box_loss_total(sets, flats, 5.0, 2.0, reduction="sum")
    assert _close(one.item(), total.item())
    one.backward()
    for s, o in enumerate(sets):
        torch.testing.assert_close(o["pred_boxes"].grad, grads[s], rtol=1e-6, atol=1e-7)
    # class maps of all sets in one launch == per-set torch scatter
    no_object = wl.n_logits - 1
    maps = losses.target_class_maps(flats, no_object)
    for s, f in enumerate(flats):
        assert torch.equal(maps[s], f.target_classes(no_object))


def test_bitwise_reproducible_and_rejects_foreign_sets():
    import detr_amd
    from detr_amd import losses, synth

    wl, sets, targets = synth.make_workload("cfg2", device="cuda", batch=8)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    flats = matcher.match_flat(sets, targets)
    a = losses.matched_box_losses(sets, flats)
    b = losses.matched_box_losses(sets, flats)
    for (a0, a1), (b0, b1) in zip(a, b):
        assert torch.equal(a0, b0) and torch.equal(a1, b1)
    other = matcher.match_flat(sets, targets)
    with pytest.raises(ValueError):
        losses.matched_box_losses(sets[:2], [flats[0], other[1]])
    with pytest.raises(ValueError):
        losses.matched_box_losses([{"pred_boxes": sets[0]["pred_boxes"].double()}], flats[:1])


def test_tail_views_of_wider_tensors_losses_and_aux_assigners():
    """The matched queries of the denoising models are views with the images further apart than Q rows
    (pred_boxes.split([n_dn, Q], dim=1)[1]): the pair-loss kernel and the auxiliary assigners read them in place;
    losses, gradients (w.r.t. the WIDE tensor, through the view) and assignments equal those of contiguous copies."""
    import detr_amd
    from detr_amd import aux_assign, losses, synth

    wl, sets, targets = synth.make_workload("cfg3", device="cuda", batch=6)
    sets = sets[:3]
    n_dn = 7
    wide, views = [], []
    for o in sets:
        wb = torch.cat([torch.rand_like(o["pred_boxes"][:, :n_dn]), o["pred_boxes"]], dim=1).requires_grad_(True)
        wl_ = torch.cat([torch.zeros_like(o["pred_logits"][:, :n_dn]), o["pred_logits"]], dim=1)
        wide.append(wb)
        views.append({"pred_logits": wl_.split([n_dn, 300], dim=1)[1], "pred_boxes": wb.split([n_dn, 300], dim=1)[1]})
        assert not views[-1]["pred_boxes"].is_contiguous()
    dense = [{"pred_logits": o["pred_logits"], "pred_boxes": o["pred_boxes"].clone().requires_grad_(True)} for o in sets]
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    flats_v, flats_d = matcher.match_flat(views, targets), matcher.match_flat(dense, targets)
    total_v = losses.matched_box_loss_total(views, flats_v, 5.0, 2.0)
    total_d = losses.matched_box_loss_total(dense, flats_d, 5.0, 2.0)
    assert torch.equal(total_v, total_d)
    grads_v = torch.autograd.grad(total_v, wide)
    grads_d = torch.autograd.grad(total_d, [o["pred_boxes"] for o in dense])
    for gv, gd in zip(grads_v, grads_d):
        assert torch.equal(gv[:, n_dn:], gd) and not gv[:, :n_dn].any()
    def same(x, y):
        if isinstance(x, torch.Tensor):
            return torch.equal(x, y)
        return len(x) == len(y) and all(same(p, q) for p, q in zip(x, y))

    for head in (aux_assign.ATSS(9), aux_assign.FasterRCNN()):
        assert same(head.assign_many(views, targets), head.assign_many(dense, targets))
    maps_v, n_v = aux_assign.FasterRCNN().dense_class_targets(views, targets, 2)
    maps_d, n_d = aux_assign.FasterRCNN().dense_class_targets(dense, targets, 2)
    assert torch.equal(n_v, n_d) and all(torch.equal(x, y) for x, y in zip(maps_v, maps_d))
