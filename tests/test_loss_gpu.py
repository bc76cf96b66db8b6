"""GPU parity of the cross-entropy kernels (the criterion of the train step:
torch.nn.CrossEntropyLoss(ignore_index=num_classes), configs/model/semantic/default.yaml:47-49)
against torch in float64: loss within 1e-6 relative, d logits within 1e-6 of the largest entry."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,C", [(1, 13), (255, 13), (256, 13), (70_001, 13), (428_571, 13),
                                    (5000, 2), (5000, 16), (5000, 17), (5000, 32)])
@pytest.mark.parametrize("ignore", [False, True])
def test_cross_entropy_matches_torch(rows, C, ignore, dev):
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    logits = (torch.randn(rows, C, generator=g) * 3).to(dev).requires_grad_()
    target = torch.randint(0, C + (1 if ignore else 0), (rows,), generator=g).to(dev)
    ii = C if ignore else -100
    if ignore and bool((target == C).all()):
        target[0] = 0
    w = torch.tensor(0.37, device=dev)
    loss = ops.cross_entropy(logits, target, ignore_index=ii)
    (loss * w).backward()
    ld = logits.detach().double().requires_grad_()
    ref = torch.nn.functional.cross_entropy(ld, target, ignore_index=ii)
    (ref * w.double()).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6 * max(1.0, abs(float(ref.detach())))
    assert float((logits.grad.double() - ld.grad).abs().max()) <= 1e-6 * float(ld.grad.abs().max()) + 1e-12
    # deterministic
    loss2 = ops.cross_entropy(logits.detach(), target, ignore_index=ii)
    assert float(loss2) == float(loss)


def test_cross_entropy_falls_back_for_wide_or_odd_inputs(dev):
    from superpoint_transformer_amd import ops
    x = torch.randn(100, 40, device=dev)
    t = torch.randint(0, 40, (100,), device=dev)
    assert torch.allclose(ops.cross_entropy(x, t), torch.nn.functional.cross_entropy(x, t))


def test_cross_entropy_poisons_the_loss_on_a_label_out_of_range(dev):
    """torch raises a device assert for a target outside [0, C) that is not ignore_index; the HIP
    loss comes out NaN (never a silent mean over fewer rows), while ignore_index rows are skipped."""
    from superpoint_transformer_amd import ops
    lg = torch.randn(1000, 13, device=dev)
    t = torch.randint(0, 13, (1000,), device=dev)
    t[5] = -100
    ok = ops.cross_entropy(lg, t, ignore_index=-100)
    assert torch.isfinite(ok).item()
    for bad in (13, -1, 1 << 40):
        t2 = t.clone()
        t2[17] = bad
        assert torch.isnan(ops.cross_entropy(lg, t2, ignore_index=-100)).item()
