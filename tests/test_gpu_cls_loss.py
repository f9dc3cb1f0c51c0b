"""The classification loss on the dense class map (dm_focal_loss_f32 / dm_cross_entropy_f32 through
detr_amd.losses) against the golden vectors of the reference's FocalLoss / CrossEntropy /
calc_cls_loss_with_ignore (tests/golden/cls), the CPU oracle and stock torch ops on the same inputs."""
import numpy as np
import pytest
import torch

import golden_io

pytestmark = pytest.mark.gpu

# loss relative 1e-5; gradient entry-relative 5e-5 plus a floor of 2e-6 of the largest entry (float32 exp / log1p / pow of
# the device library against torch's CPU ones; saturated logits lose digits in 1 - sigmoid(x))
LOSS_RTOL, GRAD_RTOL, GRAD_FLOOR = 1e-5, 5e-5, 2e-6


def product(case, device="cuda", requires_grad=True):
    from detr_amd import losses

    x = torch.from_numpy(case.logits).to(device).requires_grad_(requires_grad)
    cls = torch.from_numpy(case.target_cls).to(device)
    if case.kind == "focal":
        loss = losses.focal_loss(x, cls, case.num_boxes, case.alpha, case.gamma, ignore_index=case.ignore_index)
    else:
        loss = losses.cross_entropy(x, cls, torch.from_numpy(case.weight).to(device))
    return x, loss


@pytest.mark.parametrize("name", golden_io.cls_case_names())
def test_loss_and_gradient_match_the_reference(name):
    from detr_amd import losses

    case = golden_io.ClsLossCase(name)
    x, loss = product(case)
    assert loss.dim() == 0
    assert abs(loss.item() - case.loss) <= LOSS_RTOL * abs(case.loss) + 1e-7, (loss.item(), case.loss)
    loss.backward()
    floor = GRAD_FLOOR * float(np.abs(case.grad).max())
    np.testing.assert_allclose(x.grad.cpu().numpy(), case.grad, rtol=GRAD_RTOL, atol=floor)
    losses.check_cls_status()          # none of the golden cases holds a class the reference refuses
    # without a gradient to prepare: the same value
    with torch.no_grad():
        _, again = product(case, requires_grad=False)
    assert again.item() == loss.item()


def torch_focal(x, cls, n_cls, num_boxes, alpha, gamma):
    """The reference's expressions (detr/loss.py:208-213, 233-242) with stock torch ops on the device."""
    F = torch.nn.functional
    onehot = F.one_hot(cls, n_cls + 1)[..., :-1].float()
    prob = x.sigmoid()
    ce = F.binary_cross_entropy_with_logits(x, onehot, reduction="none")
    p_t = prob * onehot + (1 - prob) * (1 - onehot)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * onehot + (1 - alpha) * (1 - onehot)) * loss
    return loss.mean(1).sum() / num_boxes * x.shape[1]


@pytest.mark.parametrize("B,Q,C", [(64, 600, 2), (8, 900, 91), (3, 77, 7)])
def test_focal_at_workload_sizes_against_stock_torch_ops(B, Q, C):
    from detr_amd import losses

    gen = torch.Generator().manual_seed(B * Q + C)
    x0 = (torch.randn(B, Q, C, generator=gen) * 2.0).cuda()
    cls = torch.full((B, Q), C, dtype=torch.int64)
    pick = torch.rand(B, Q, generator=gen) < 0.03
    cls[pick] = torch.randint(0, C, (int(pick.sum()),), generator=gen)
    cls, nb = cls.cuda(), max(int(pick.sum()), 1)
    a = x0.clone().requires_grad_(True)
    b = x0.clone().requires_grad_(True)
    want = torch_focal(a, cls, C, nb, 0.25, 2.0)
    got = losses.focal_loss(b, cls, nb, 0.25, 2.0)
    (want * 1.7).backward()
    (got * 1.7).backward()          # (an upstream gradient other than one)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    floor = GRAD_FLOOR * float(a.grad.abs().max())
    torch.testing.assert_close(b.grad, a.grad, rtol=GRAD_RTOL, atol=floor)
    # reproducible bit for bit
    c = x0.clone().requires_grad_(True)
    again = losses.focal_loss(c, cls, nb, 0.25, 2.0)
    (again * 1.7).backward()
    assert again.item() == got.item() and torch.equal(c.grad, b.grad)


# (class counts that take each kernel: a few logits per thread, 2^k lanes per row, the 16-byte variant for C = 0 mod 4 with
# one lane and with a wavefront per row, the cached three-pass walk beyond 1024 classes)
@pytest.mark.parametrize("B,Q,C", [(64, 600, 3), (8, 900, 92), (2, 31, 1204), (5, 33, 4), (3, 50, 512), (2, 31, 1024),
                                   (3, 41, 90), (2, 37, 257)])
def test_cross_entropy_at_workload_sizes_against_stock_torch_ops(B, Q, C):
    from detr_amd import losses

    gen = torch.Generator().manual_seed(B * Q + C)
    x0 = (torch.randn(B, Q, C, generator=gen) * 2.0).cuda()
    cls = torch.full((B, Q), C - 1, dtype=torch.int64)
    pick = torch.rand(B, Q, generator=gen) < 0.05
    cls[pick] = torch.randint(0, C - 1, (int(pick.sum()),), generator=gen)
    cls = cls.cuda()
    w = torch.ones(C)
    w[-1] = 0.1
    w = w.cuda()
    a = x0.clone().requires_grad_(True)
    b = x0.clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(a.transpose(1, 2), cls, weight=w)
    got = losses.cross_entropy(b, cls, w)
    (want * 0.6).backward()
    (got * 0.6).backward()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    floor = GRAD_FLOOR * float(a.grad.abs().max())
    torch.testing.assert_close(b.grad, a.grad, rtol=GRAD_RTOL, atol=floor)
    # no class weights
    torch.testing.assert_close(losses.cross_entropy(x0, cls), torch.nn.functional.cross_entropy(x0.transpose(1, 2), cls),
                               rtol=1e-5, atol=1e-6)


def test_modules_take_the_references_call_convention():
    """FocalLoss / CrossEntropy stand where the criterion's cls_criterion is: called with the transposed view of
    pred_logits (detr/loss.py:127) or with pred_logits itself (co_detr/loss.py:135), the class map and num_boxes."""
    from detr_amd import losses

    case = golden_io.ClsLossCase("focal_c2_q50")
    x = torch.from_numpy(case.logits).cuda().requires_grad_(True)
    cls = torch.from_numpy(case.target_cls).cuda()
    for channels_first in (None, True):
        mod = losses.FocalLoss(case.alpha, case.gamma, case.n_cls, channels_first=channels_first)
        loss = mod(x.transpose(1, 2), cls, case.num_boxes)
        assert abs(loss.item() - case.loss) <= LOSS_RTOL * abs(case.loss)
    for channels_first in (None, False):
        mod = losses.FocalLoss(case.alpha, case.gamma, case.n_cls, channels_first=channels_first)
        assert abs(mod(x, cls, case.num_boxes).item() - case.loss) <= LOSS_RTOL * abs(case.loss)
    x.grad = None
    losses.FocalLoss(case.alpha, case.gamma, case.n_cls)(x.transpose(1, 2), cls, case.num_boxes).backward()
    assert x.grad.shape == x.shape
    np.testing.assert_allclose(x.grad.cpu().numpy(), case.grad, rtol=GRAD_RTOL, atol=GRAD_FLOOR * float(np.abs(case.grad).max()))

    ce = golden_io.ClsLossCase("ce_c3_q600")
    y = torch.from_numpy(ce.logits).cuda()
    ycls = torch.from_numpy(ce.target_cls).cuda()
    mod = losses.CrossEntropy(torch.from_numpy(ce.weight)).cuda()
    assert abs(mod(y.transpose(1, 2), ycls, ce.num_boxes).item() - ce.loss) <= LOSS_RTOL * abs(ce.loss)
    assert abs(mod(y, ycls, ce.num_boxes).item() - ce.loss) <= LOSS_RTOL * abs(ce.loss)


def test_what_the_kernel_does_not_take_goes_to_the_fallback():
    from detr_amd import losses

    seen = []

    def own(inputs, targets, num_boxes):
        seen.append(tuple(inputs.shape))
        return inputs.sum() * 0

    mod = losses.FocalLoss(0.25, 2.0, 3, fallback=own)
    cls = torch.zeros(2, 5, dtype=torch.int64)
    mod(torch.zeros(2, 3, 5), cls, 1)                                   # CPU tensors
    mod(torch.zeros(2, 3, 5, dtype=torch.float64).cuda(), cls.cuda(), 1)      # not float32
    mod(torch.zeros(2, 3, 3).cuda(), torch.zeros(2, 3, dtype=torch.int64).cuda(), 1)    # Q == C: layout cannot be told
    assert seen == [(2, 3, 5), (2, 3, 5), (2, 3, 3)]
    with pytest.raises(RuntimeError):
        losses.FocalLoss(0.25, 2.0, 3)(torch.zeros(2, 3, 5), cls, 1)
    # an explicit layout settles Q == C
    got = losses.FocalLoss(0.25, 2.0, 3, channels_first=True)(torch.zeros(2, 3, 3).cuda(), torch.full((2, 3), 3).cuda(), 1)
    want = torch_focal(torch.zeros(2, 3, 3).cuda(), torch.full((2, 3), 3).cuda(), 3, 1, 0.25, 2.0)
    torch.testing.assert_close(got, want)


def test_class_values_the_reference_refuses_are_reported():
    from detr_amd import losses

    losses.check_cls_status()
    x = torch.zeros(2, 9, 4).cuda()
    cls = torch.full((2, 9), 4, dtype=torch.int64).cuda()
    cls[1, 3] = 5                                  # one_hot(targets, n_cls + 1) raises for it
    bad = losses.focal_loss(x, cls, 1)
    cls[1, 3] = 4
    good = losses.focal_loss(x[:, :8], cls[:, :8], 1)
    with pytest.raises(RuntimeError):
        losses.check_cls_status()
    losses.check_cls_status()                       # reported once
    # the offending row counted as ignored
    torch.testing.assert_close(bad / 17, good / 16)
    cls[0, 0] = -7
    losses.cross_entropy(x, cls)
    with pytest.raises(RuntimeError):
        losses.check_cls_status()
    with pytest.raises(TypeError):
        losses.focal_loss(x.double(), cls, 1)
    with pytest.raises(ValueError):
        losses.focal_loss(x, cls[:, :5], 1)


def test_other_streams_and_unaligned_views():
    from detr_amd import losses

    case = golden_io.ClsLossCase("syn_focal_gamma1p5")
    base = torch.zeros(case.logits.size + 3, device="cuda")
    view = base[3:].view(case.logits.shape)                   # 12 bytes off a 16-byte boundary
    view.copy_(torch.from_numpy(case.logits))
    cls = torch.from_numpy(case.target_cls).cuda()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        loss = losses.focal_loss(view, cls, case.num_boxes, case.alpha, case.gamma)
    side.synchronize()
    assert abs(loss.item() - case.loss) <= LOSS_RTOL * abs(case.loss)


def test_cross_entropy_off_a_16_byte_boundary():
    """Logits that do not start on a 16-byte boundary reach the C entry point as they are only when the Python layer lets
    them: it copies.  Called through the C ABI directly, dm_cross_entropy_f32 takes them with its 4-byte kernels."""
    import ctypes

    from detr_amd import _lib, losses
    from detr_amd._lib import lib

    case = golden_io.ClsLossCase("syn_ce_c92")
    n = case.logits.size
    base = torch.zeros(n + 1, device="cuda")
    view = base[1:].view(case.logits.shape)
    view.copy_(torch.from_numpy(case.logits))
    assert view.data_ptr() % 16 == 4
    cls = torch.from_numpy(case.target_cls).cuda()
    w = torch.from_numpy(case.weight).cuda()
    out = torch.zeros(2, device="cuda")
    grad = torch.zeros(n + 1, device="cuda")[1:]
    partials = torch.empty(_lib.CLS_LOSS_PARTIALS, dtype=torch.float64, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.dm_cross_entropy_f32(view.data_ptr(), cls.data_ptr(), w.data_ptr(), cls.numel(), 92, -100, out.data_ptr(),
                                  grad.data_ptr(), partials.data_ptr(), status.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert abs(out[0].item() - case.loss) <= LOSS_RTOL * abs(case.loss) and int(status.item()) == 0
    got = (grad * out[1]).view(case.logits.shape).cpu().numpy()
    np.testing.assert_allclose(got, case.grad, rtol=GRAD_RTOL, atol=GRAD_FLOOR * float(np.abs(case.grad).max()))
    # and through the Python layer (which hands the kernel an aligned copy)
    assert abs(losses.cross_entropy(view, cls, w).item() - case.loss) <= LOSS_RTOL * abs(case.loss)


def test_random_shapes_against_the_oracle():
    """Seeded sweep over row counts, class counts (1 .. 130: every position of a quad of four logits relative to the
    row boundaries, tails of one to three logits, the one-class and the 16-byte cross-entropy kernels), gamma, alpha,
    ignored rows and class weights: loss and gradient against the CPU restatement."""
    from detr_amd import losses
    from oracle import cls_loss as oracle_cls

    rng = np.random.default_rng(20240607)
    for trial in range(80):
        B, Q = int(rng.integers(1, 5)), int(rng.integers(1, 70))
        C = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 31, 64, 91, 92, 130]))
        x = (rng.standard_normal((B, Q, C)) * 3).astype(np.float32)
        if trial % 2 == 0:
            gamma = float(rng.choice([2.0, 2.0, 0.0, 1.0, 1.5, 3.0]))
            alpha = float(rng.choice([0.25, 0.6, -1.0]))
            cls = rng.integers(0, C + 1, (B, Q))
            ignore = None
            if trial % 4 == 0:
                ignore = -1
                cls[rng.random((B, Q)) < 0.3] = -1
            nb = int(rng.integers(1, 20))
            want, want_grad = oracle_cls.focal_loss(x, cls, nb, alpha, gamma, ignore_index=ignore)
            xt = torch.from_numpy(x).cuda().requires_grad_(True)
            got = losses.focal_loss(xt, torch.from_numpy(cls).cuda(), nb, alpha, gamma, ignore_index=ignore)
        else:
            cls = rng.integers(0, C, (B, Q))
            cls[rng.random((B, Q)) < 0.2] = -100
            if (cls == -100).all():
                cls[0, 0] = 0
            w = (rng.random(C) + 0.1).astype(np.float32) if trial % 4 == 1 else None
            want, want_grad = oracle_cls.cross_entropy(x, cls, w)
            xt = torch.from_numpy(x).cuda().requires_grad_(True)
            got = losses.cross_entropy(xt, torch.from_numpy(cls).cuda(), None if w is None else torch.from_numpy(w).cuda())
        got.backward()
        what = (trial, B, Q, C)
        assert abs(got.item() - float(want)) <= LOSS_RTOL * abs(float(want)) + 1e-7, what
        np.testing.assert_allclose(xt.grad.cpu().numpy(), want_grad, rtol=GRAD_RTOL,
                                   atol=GRAD_FLOOR * float(np.abs(want_grad).max()) + 1e-12, err_msg=str(what))
    losses.check_cls_status()
