"""CPU: host logic of the deferred prune (eogs2_amd.optim.retire_rows / alive_rows) — pure tensor bookkeeping, so it runs on a
plain torch.optim.Adam over CPU tensors; the rendering side (retired Gaussians list nothing, survivors equal the compacted model
bit for bit, FusedAdam leaves them retired) is tests/test_gpu_optim.py::test_retired_rows_render_nothing_and_stay_retired."""
import pytest
import torch

from eogs2_amd.optim import RETIRED_LOGIT, alive_rows, retire_rows


def _opt(P):
    g = torch.Generator().manual_seed(0)
    groups = [{"params": [torch.nn.Parameter(torch.randn(P, 3, generator=g))], "lr": 1e-3, "name": "xyz"},
              {"params": [torch.nn.Parameter(torch.randn(P, 1, generator=g))], "lr": 5e-2, "name": "opacity"}]
    return torch.optim.Adam(groups, lr=0.0, eps=1e-15)


def test_retire_marks_rows_without_changing_shapes_or_addresses():
    opt = _opt(100)
    op = opt.param_groups[1]["params"][0]
    ptr, before = op.data_ptr(), op.detach().clone()
    keep = torch.arange(100) % 3 != 0
    retire_rows(opt, keep)
    assert op.data_ptr() == ptr and op.shape == (100, 1)  # in place: a recorded graph keeps reading the same tensor
    assert torch.equal(op.detach()[keep], before[keep]) and bool((op.detach()[~keep] == RETIRED_LOGIT).all())
    assert torch.equal(alive_rows(opt), keep)
    assert float(torch.sigmoid(op.detach()[~keep]).max()) == 0.0  # opacity exactly 0
    # retiring again with a mask that asks to keep retired rows cannot revive them through alive_rows & keep
    retire_rows(opt, torch.ones(100, dtype=torch.bool))
    assert torch.equal(alive_rows(opt), keep)


def test_adam_with_stale_momentum_leaves_retired_rows_retired():
    opt = _opt(64)
    xyz, op = (g["params"][0] for g in opt.param_groups)
    xyz.grad, op.grad = torch.randn_like(xyz), torch.randn_like(op) * 10.0
    opt.step()  # momentum everywhere
    keep = torch.rand(64, generator=torch.Generator().manual_seed(1)) > 0.5
    retire_rows(opt, keep)
    for _ in range(20):  # what the rasterizer hands a retired Gaussian: zero gradients
        xyz.grad, op.grad = torch.zeros_like(xyz), torch.where(keep[:, None], torch.randn_like(op), torch.zeros_like(op))
        opt.step()
    assert torch.equal(alive_rows(opt), keep)
    assert torch.isfinite(xyz).all()


def test_unknown_group_name_is_an_error():
    opt = _opt(4)
    with pytest.raises(KeyError):
        retire_rows(opt, torch.ones(4, dtype=torch.bool), name="opacities")
    with pytest.raises(KeyError):
        alive_rows(opt, name="opacities")


def test_mask_of_another_dtype_or_length():
    opt = _opt(6)
    retire_rows(opt, torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.uint8))  # a 0/1 mask is a mask, not bits to invert
    assert alive_rows(opt).tolist() == [True, False, True, True, False, True]
    with pytest.raises(ValueError):
        retire_rows(opt, torch.ones(5, dtype=torch.bool))
