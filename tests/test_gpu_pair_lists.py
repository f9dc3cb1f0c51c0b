"""The box loss on index tuples that do not come from the matcher (dm_pair_list_loss_f32 through
detr_amd.losses.pair_list_box_loss_sums and detr_amd.criterion.flat_of_foreign_indices): against the reference's
calc_box_loss + autograd on denoising-group-shaped and many-to-many pairs (tests/golden/pairs), and the CPU oracle at
the size of a real step."""
import numpy as np
import pytest
import torch

import golden_io

pytestmark = pytest.mark.gpu


def _flat(case, device="cuda", index_device="cpu"):
    import detr_amd
    from detr_amd.criterion import flat_of_foreign_indices

    pred = torch.from_numpy(case.pred_boxes).to(device).requires_grad_(True)
    targets = [{"boxes": torch.from_numpy(np.ascontiguousarray(b)).to(device),
                "labels": torch.zeros(len(b), dtype=torch.int64, device=device)} for b in case.target_boxes]
    indices = [(torch.from_numpy(np.ascontiguousarray(s)).to(index_device), torch.from_numpy(np.ascontiguousarray(t)).to(index_device))
               for s, t in case.indices]
    matcher = detr_amd.HungarianMatcher("focal", 2.0, 5.0, 2.0)
    flat = flat_of_foreign_indices(matcher, {"pred_boxes": pred}, targets, indices)
    return matcher, pred, targets, indices, flat


@pytest.mark.parametrize("index_device", ["cpu", "cuda"])
@pytest.mark.parametrize("name", golden_io.pair_case_names())
def test_pair_list_losses_match_the_reference(name, index_device):
    from detr_amd import losses
    from detr_amd.criterion import flat_of_foreign_indices

    case = golden_io.PairListCase(name)
    matcher, pred, targets, indices, flat = _flat(case, index_device=index_device)
    n = sum(len(s) for s, _ in case.indices)
    assert flat is not None and flat.num_matched == n
    # built once per indices object; not for other targets
    assert flat_of_foreign_indices(matcher, {"pred_boxes": pred}, targets, indices) is flat
    assert flat_of_foreign_indices(matcher, {"pred_boxes": pred}, list(targets), indices) is None
    sums = losses.pair_list_box_loss_sums(pred, flat.packed.boxes, flat.batch_idx, flat.src_idx, flat.tgt_flat)
    l1, giou = sums[0] / n, sums[1] / n
    assert abs(l1.item() - case.l1) <= 1e-5 * max(1.0, abs(case.l1))
    assert abs(giou.item() - case.giou) <= 1e-5 * max(1.0, abs(case.giou))
    g1 = torch.autograd.grad(l1, pred, retain_graph=True)[0].cpu().numpy()
    g2 = torch.autograd.grad(giou, pred)[0].cpu().numpy()
    np.testing.assert_allclose(g1, case.grad_l1, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(g2, case.grad_giou, rtol=1e-5, atol=1e-6)
    losses.check_cls_status()
    matcher.clear_prefetched()


def test_a_step_sized_denoising_group_against_the_oracle():
    """64 images, 5 groups of max_obj queries each (positives and negatives interleaved as
    /root/reference/models/dino_detr/dino_func.py:117-124 lays them out): ~8000 pairs, where the reference builds an
    [N, N] matrix of 64 M entries."""
    from detr_amd import losses, synth
    from detr_amd.criterion import flat_of_foreign_indices
    from oracle import pair_loss
    import detr_amd

    wl, sets, targets = synth.make_workload("cfg2", device="cuda")
    sizes = [len(t["labels"]) for t in targets]
    num_group, max_obj = 5, max(sizes)
    Q = num_group * max_obj * 2
    gen = torch.Generator().manual_seed(7)
    pred = torch.cat([torch.rand(64, Q, 2, generator=gen), torch.rand(64, Q, 2, generator=gen) * 0.3 + 0.01], -1).cuda().requires_grad_(True)
    dn_targets, indices = [], []
    for t, n in zip(targets, sizes):
        boxes = t["boxes"]
        if n:
            pad = boxes[torch.randint(0, n, (max_obj - n,), generator=gen).cuda()] if n < max_obj else boxes[:0]
            boxes = torch.cat([boxes, pad]).repeat(num_group, 1)
        dn_targets.append({"boxes": boxes.contiguous(), "labels": torch.zeros(len(boxes), dtype=torch.int64, device="cuda")})
        num_pos = max_obj if n else 0
        idx = torch.arange(num_pos).unsqueeze(0).repeat(num_group, 1)
        indices.append(((idx + torch.arange(num_group).unsqueeze(1) * num_pos * 2).view(-1),
                        (idx + torch.arange(num_group).unsqueeze(1) * num_pos).view(-1)))
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    flat = flat_of_foreign_indices(matcher, {"pred_boxes": pred}, dn_targets, indices)
    n = flat.num_matched
    assert n == sum(len(s) for s, _ in indices) > 5000
    sums = losses.pair_list_box_loss_sums(pred, flat.packed.boxes, flat.batch_idx, flat.src_idx, flat.tgt_flat)
    want = pair_loss.box_losses(pred.detach().cpu().numpy(), [t["boxes"].cpu().numpy() for t in dn_targets],
                                [(s.numpy(), t.numpy()) for s, t in indices])
    assert abs(sums[0].item() / n - want[0]) <= 1e-5 * max(1.0, abs(want[0]))
    assert abs(sums[1].item() / n - want[1]) <= 1e-5 * max(1.0, abs(want[1]))
    total = (5.0 * sums[0] + 2.0 * sums[1]) / n
    (grad,) = torch.autograd.grad(total, pred)
    np.testing.assert_allclose(grad.cpu().numpy(), 5.0 * want[2] + 2.0 * want[3], rtol=1e-5, atol=1e-6)
    # the tail view the denoising models hand over (rows dense inside every image) is read in place
    wide = torch.zeros(64, Q + 7, 4, device="cuda")
    wide[:, 7:] = pred.detach()
    view = wide[:, 7:]
    again = losses.pair_list_box_loss_sums(view, flat.packed.boxes, flat.batch_idx, flat.src_idx, flat.tgt_flat)
    assert torch.equal(again, sums.detach())
    # bitwise reproducible
    assert torch.equal(losses.pair_list_box_loss_sums(pred, flat.packed.boxes, flat.batch_idx, flat.src_idx, flat.tgt_flat), sums)


def test_indices_out_of_range_are_reported_and_contribute_nothing():
    from detr_amd import losses

    losses.check_cls_status()
    pred = torch.rand(2, 5, 4, device="cuda") * 0.3 + 0.2
    tgt = torch.rand(3, 4, device="cuda") * 0.3 + 0.2
    ok = (torch.tensor([0, 1], device="cuda"), torch.tensor([1, 4], device="cuda"), torch.tensor([0, 2], device="cuda"))
    bad = (torch.tensor([0, 1, 2], device="cuda"), torch.tensor([1, 4, 0], device="cuda"), torch.tensor([0, 2, 1], device="cuda"))
    want = losses.pair_list_box_loss_sums(pred, tgt, *ok)
    losses.check_cls_status()
    got = losses.pair_list_box_loss_sums(pred, tgt, *bad)          # image 2 does not exist
    with pytest.raises(RuntimeError):
        losses.check_cls_status()
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        losses.pair_list_box_loss_sums(pred, tgt, ok[0], ok[1], ok[2][:1])
    with pytest.raises(ValueError):
        losses.pair_list_box_loss_sums(pred.double(), tgt, *ok)
    # nothing to pair
    empty = torch.zeros(0, dtype=torch.int64, device="cuda")
    assert losses.pair_list_box_loss_sums(pred, tgt, empty, empty, empty).tolist() == [0.0, 0.0]


def test_what_is_not_a_list_of_index_pairs_is_left_to_the_caller():
    import detr_amd
    from detr_amd.criterion import flat_of_foreign_indices

    m = detr_amd.HungarianMatcher("focal", 2.0, 5.0, 2.0)
    pred = torch.rand(2, 5, 4, device="cuda")
    targets = [{"boxes": torch.rand(2, 4, device="cuda"), "labels": torch.zeros(2, dtype=torch.int64, device="cuda")} for _ in range(2)]
    pair = (torch.tensor([0, 1]), torch.tensor([1, 0]))
    assert flat_of_foreign_indices(m, {"pred_boxes": pred}, targets, [pair]) is None                  # one image short
    assert flat_of_foreign_indices(m, {"pred_boxes": pred.cpu()}, targets, [pair, pair]) is None       # CPU predictions
    assert flat_of_foreign_indices(m, {"pred_boxes": pred}, targets, [(pair[0].int(), pair[1].int())] * 2) is None
    assert flat_of_foreign_indices(m, {"pred_boxes": pred}, targets, [(pair[0], pair[1][:1])] * 2) is None
    # a target position beyond the image's targets, or a negative (wrap-around) position: torch's own indexing decides
    assert flat_of_foreign_indices(m, {"pred_boxes": pred}, targets, [(pair[0], torch.tensor([1, 2]))] * 2) is None
    assert flat_of_foreign_indices(m, {"pred_boxes": pred}, targets, [(pair[0], torch.tensor([-1, 0]))] * 2) is None
    assert flat_of_foreign_indices(m, {"pred_boxes": pred}, targets, [(torch.tensor([0, 5]), pair[1])] * 2) is None
    none = (torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64))
    assert flat_of_foreign_indices(m, {"pred_boxes": pred}, targets, [none, none]) is None             # no pair at all
    assert flat_of_foreign_indices(m, {"pred_boxes": pred}, targets, [pair, pair]) is not None
    m.clear_prefetched()
