"""Optimizer step and density-control compaction over the C-ABI of include/eogs_optim.h.

* `FusedAdam` — drop-in for the reference's `torch.optim.Adam(l, lr=0.0, eps=1e-15)` with one single-tensor param group
  per Gaussian attribute (src/gaussiansplatting/scene/gaussian_model.py:228-262). It *is* a `torch.optim.Adam`
  (same `param_groups`, same `state[p]["exp_avg"|"exp_avg_sq"|"step"]`), so the reference's optimizer surgery
  (`replace_tensor_to_optimizer`, `_prune_optimizer`, `cat_tensors_to_optimizer`, gaussian_model.py:451-540) and
  checkpointing keep working; only `step()` is replaced: all groups in ONE HIP launch.
* `compact_rows(mask, tensors)` — every `tensor[mask]` of `prune_points` / `_prune_optimizer`
  (gaussian_model.py:466-505) in one scan + one gather launch and a single host sync.
* `prune_optimizer(optimizer, mask, extra=())` — `_prune_optimizer` + the statistics of `prune_points`, on top of it.

No CPU / eager fallback: arithmetic only in the HIP library.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from ._abi import AdamTensor
from .rasterizer import _Ctx, _ptr

MAX_TENSORS = 16  # EOGS_ADAM_MAX_TENSORS


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        abi = _lib.get()
        # tensors are batched per (device, betas, eps, step): the reference has one such class
        batches = {}
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("weight_decay") or group.get("maximize"):
                raise NotImplementedError("FusedAdam mirrors the reference's configuration: plain Adam")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam: dense fp32 parameters only")
                state = self.state[p]
                if len(state) == 0:  # lazy init, like torch.optim.Adam
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not (p.is_contiguous() and state["exp_avg"].is_contiguous() and state["exp_avg_sq"].is_contiguous()):
                    raise RuntimeError("FusedAdam: parameters and moments must be contiguous")
                state["step"] += 1
                t = int(state["step"])
                key = (p.device, group["betas"], group["eps"], t)
                batches.setdefault(key, []).append((p, p.grad.contiguous(), state, float(group["lr"])))
        for (dev, (b1, b2), eps, t), items in batches.items():
            with _Ctx(abi, dev) as cx:
                for i0 in range(0, len(items), MAX_TENSORS):
                    chunk = items[i0:i0 + MAX_TENSORS]
                    arr = (AdamTensor * len(chunk))()
                    for a, (p, g, st, lr) in zip(arr, chunk):
                        a.param, a.grad = p.data_ptr(), g.data_ptr()
                        a.exp_avg, a.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                        a.numel, a.lr = p.numel(), lr
                    abi.check(abi.adam_step(len(chunk), ctypes.cast(arr, ctypes.c_void_p), b1, b2, eps, t, cx.stream))
        return loss


def compact_rows(mask, tensors):
    """[t[mask] for t in tensors] for row-major tensors sharing dim 0 (keep-mask: bool/uint8 [N]), one launch pair."""
    abi = _lib.get()
    if mask.dtype not in (torch.bool, torch.uint8) or mask.ndim != 1:
        raise RuntimeError("compact_rows: mask must be a 1-D bool / uint8 tensor")
    dev, N = mask.device, mask.shape[0]
    srcs = []
    for t in tensors:
        if t.shape[0] != N or t.device != dev:
            raise RuntimeError("compact_rows: every tensor needs the mask's length and device")
        if t.element_size() != 4:
            raise RuntimeError("compact_rows: 4-byte element types only")
        srcs.append(t.detach().contiguous())
    keep = mask.contiguous().view(torch.uint8)
    with _Ctx(abi, dev) as cx:
        n = ctypes.c_size_t()
        abi.check(abi.compact_bytes(N, ctypes.byref(n)))
        ws = torch.empty((n.value,), dtype=torch.uint8, device=dev)
        n_keep = ctypes.c_int64()
        abi.check(abi.compact_plan(N, _ptr(keep), _ptr(ws), ws.numel(), ctypes.byref(n_keep), cx.stream))
        K = n_keep.value
        outs = [torch.empty((K,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in srcs]
        row_bytes = [(t.numel() // N if N else 0) * 4 for t in srcs]
        if any(rb > 256 for rb in row_bytes):
            raise RuntimeError("compact_rows: rows of at most 64 elements")
        k = len(srcs)
        if k and N:
            S = (ctypes.c_void_p * k)(*[t.data_ptr() if t.numel() else None for t in srcs])
            D = (ctypes.c_void_p * k)(*[o.data_ptr() if o.numel() else None for o in outs])
            RB = (ctypes.c_int * k)(*[rb if K else 0 for rb in row_bytes])
            abi.check(abi.compact_apply(N, _ptr(keep), k, ctypes.cast(S, ctypes.c_void_p), ctypes.cast(D, ctypes.c_void_p),
                                        ctypes.cast(RB, ctypes.c_void_p), _ptr(ws), ws.numel(), cx.stream))
    return outs


def prune_optimizer(optimizer, mask, extra=()):
    """`GaussianModel._prune_optimizer(mask)` (gaussian_model.py:466-486) for every group at once, plus `extra`
    per-Gaussian tensors (xyz_gradient_accum, denom, max_radii2D — prune_points, gaussian_model.py:499-502).
    Returns ({group name: new nn.Parameter}, [compacted extra tensors]). `mask` marks the rows that are KEPT."""
    plan = []
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1
        p = group["params"][0]
        st = optimizer.state.get(p, None)
        plan.append((group, p, st))
    flat = []
    for _, p, st in plan:
        flat.append(p.data)
        if st is not None:
            flat += [st["exp_avg"], st["exp_avg_sq"]]
    flat += list(extra)
    out = compact_rows(mask, flat)
    optimizable, i = {}, 0
    for group, p, st in plan:
        new_p = nn.Parameter(out[i].requires_grad_(True))
        i += 1
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = out[i], out[i + 1]
            i += 2
            del optimizer.state[p]
            optimizer.state[new_p] = st
        group["params"][0] = new_p
        optimizable[group["name"]] = new_p
    return optimizable, out[i:]


def reset_opacity(optimizer, name="opacity", cap=0.01):
    """`GaussianModel.reset_opacity` (gaussian_model.py:347-352) with `replace_tensor_to_optimizer` (:451-464): the
    opacity logits are capped at logit(`cap`), both Adam moments of the group restart at zero, the group gets a new
    Parameter. Runs every `opacity_reset_interval` = 3000 iterations (gs_config/train.yaml:104): two elementwise PyTorch
    ops, not a kernel of this library. Returns {name: new nn.Parameter}."""
    out = {}
    for group in optimizer.param_groups:
        if group["name"] != name:
            continue
        p = group["params"][0]
        with torch.no_grad():
            o = torch.min(torch.sigmoid(p), torch.ones_like(p) * cap)
            new = torch.log(o / (1 - o))  # general_utils.inverse_sigmoid
        st = optimizer.state.get(p, None)
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
            del optimizer.state[p]
        new_p = nn.Parameter(new.requires_grad_(True))
        group["params"][0] = new_p
        if st is not None:
            optimizer.state[new_p] = st
        out[name] = new_p
    return out


__all__ = ["FusedAdam", "compact_rows", "prune_optimizer", "reset_opacity"]
