"""Optimizer step and density-control compaction over the C-ABI of include/eogs_optim.h.

* `FusedAdam` — drop-in for the reference's `torch.optim.Adam(l, lr=0.0, eps=1e-15)` with one single-tensor param group
  per Gaussian attribute (src/gaussiansplatting/scene/gaussian_model.py:228-262). It *is* a `torch.optim.Adam`
  (same `param_groups`, same `state[p]["exp_avg"|"exp_avg_sq"|"step"]`), so the reference's optimizer surgery
  (`replace_tensor_to_optimizer`, `_prune_optimizer`, `cat_tensors_to_optimizer`, gaussian_model.py:451-540) and
  checkpointing keep working; only `step()` is replaced: all groups in ONE HIP launch.
* `compact_rows(mask, tensors)` — every `tensor[mask]` of `prune_points` / `_prune_optimizer`
  (gaussian_model.py:466-505) in one scan + one gather launch and a single host sync.
* `prune_optimizer(optimizer, mask, extra=())` — `_prune_optimizer` + the statistics of `prune_points`, on top of it.
* `retire_rows(optimizer, keep)` / `alive_rows(optimizer)` — the same prune deferred: opacity 0 now (no shapes change, no
  wait, a recorded graph of the iteration stays valid), compaction at a coarser interval.
* `densify_and_clone` / `densify_and_split` / `cat_tensors_to_optimizer` — the clone / split densification of
  gaussian_model.py:507-660 (off by default in the reference, `only_prune: True`): every boolean-mask gather of a step in
  one scan + one gather, the random draw left to `torch.normal` so that it is the reference's own.

No CPU / eager fallback: arithmetic only in the HIP library.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from ._abi import AdamTensor, SumTensor
from .rasterizer import _Ctx, _ptr

MAX_TENSORS = 16  # EOGS_ADAM_MAX_TENSORS
SUM_MAX_TENSORS, SUM_MAX_SOURCES = 8, 4  # EOGS_SUM_MAX_TENSORS / _SOURCES


def sum_into_(dsts, sources):
    """dsts[k] += sources[0][k]; dsts[k] += sources[1][k]; ... for every k, in that order, in ONE launch (eogs_sum_into,
    include/eogs_optim.h): the gradient accumulation of an iteration's renders (autograd's `p.grad += g`, render by render:
    train_pan.py:278-469) without one add kernel per parameter and render. fp32 contiguous tensors on one GPU."""
    dsts, sources = list(dsts), [list(s) for s in sources]
    if not dsts or not sources:
        return
    if any(len(s) != len(dsts) for s in sources):
        raise ValueError("sum_into_: every source list needs one tensor per destination")
    dev = dsts[0].device
    for k, d in enumerate(dsts):
        for t in [d] + [s[k] for s in sources]:
            if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != d.numel():
                raise RuntimeError("sum_into_: contiguous fp32 tensors of equal size on one device")
    abi = _lib.get()
    with _Ctx(abi, dev) as cx, torch.no_grad():
        for s0 in range(0, len(sources), SUM_MAX_SOURCES):
            srcs = sources[s0:s0 + SUM_MAX_SOURCES]
            for k0 in range(0, len(dsts), SUM_MAX_TENSORS):
                chunk = range(k0, min(k0 + SUM_MAX_TENSORS, len(dsts)))
                arr = (SumTensor * len(chunk))()
                for a, k in zip(arr, chunk):
                    a.dst, a.numel = dsts[k].data_ptr(), dsts[k].numel()
                    for j, s in enumerate(srcs):
                        a.src[j] = s[k].data_ptr()
                abi.check(abi.sum_into(len(chunk), ctypes.cast(arr, ctypes.c_void_p), len(srcs), cx.stream))


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


def _groups(optimizer):
    plan = []
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1
        p = group["params"][0]
        plan.append((group, p, optimizer.state.get(p, None)))
    return plan


def _install(optimizer, plan, new_tensors):
    """Replaces every group's parameter (and moments) by the given tensors: `replace_tensor_to_optimizer` /
    `cat_tensors_to_optimizer` / `_prune_optimizer` all end this way (gaussian_model.py:451-540)."""
    out, i = {}, 0
    for group, p, st in plan:
        new_p = nn.Parameter(new_tensors[i].requires_grad_(True))
        i += 1
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = new_tensors[i], new_tensors[i + 1]
            i += 2
            del optimizer.state[p]
            optimizer.state[new_p] = st
        group["params"][0] = new_p
        out[group["name"]] = new_p
    return out


def cat_tensors_to_optimizer(optimizer, tensors_dict):
    """`GaussianModel.cat_tensors_to_optimizer` (gaussian_model.py:507-538): appends rows to every group's parameter and
    zero rows to its Adam moments. Returns {group name: new nn.Parameter}."""
    plan = _groups(optimizer)
    new = []
    for group, p, st in plan:
        ext = tensors_dict[group["name"]]
        new.append(torch.cat((p.data, ext), dim=0))
        if st is not None:
            new.append(torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0))
            new.append(torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0))
    return _install(optimizer, plan, new)


def _selected_rows(optimizer, mask, names, extra=()):
    """{name: rows of that group's parameter where mask} + the same rows of `extra`: ONE scan + ONE gather launch and one
    host sync for all of them (the reference indexes each tensor with the boolean mask: a nonzero + gather + sync each)."""
    by_name = {g["name"]: g["params"][0] for g in optimizer.param_groups}
    rows = compact_rows(mask, [by_name[n].data for n in names] + list(extra))
    return dict(zip(names, rows[:len(names)])), rows[len(names):]


def build_rotation(r):
    """utils/general_utils.py:82-105 (quaternion normalised here, unlike the rasterizer's computeCov3D)."""
    q = r / torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])[:, None]
    rr, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - rr * z)
    R[:, 0, 2] = 2 * (x * z + rr * y)
    R[:, 1, 0] = 2 * (x * y + rr * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - rr * x)
    R[:, 2, 0] = 2 * (x * z - rr * y)
    R[:, 2, 1] = 2 * (y * z + rr * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def densify_and_clone(optimizer, selected_pts_mask, tmp_radii=None):
    """`GaussianModel.densify_and_clone` + `densification_postfix` (gaussian_model.py:625-660, 540-571) on the
    optimizer's groups: the selected Gaussians are appended once more (zero moments for the copies). The caller computes
    `selected_pts_mask` exactly as the reference (gradient norm >= threshold & max scaling <= percent_dense * extent) and
    resets the three statistics to zeros of the new size. Returns ({name: new Parameter}, new tmp_radii or None)."""
    names = [g["name"] for g in optimizer.param_groups]
    rows, extra = _selected_rows(optimizer, selected_pts_mask, names, () if tmp_radii is None else (tmp_radii,))
    params = cat_tensors_to_optimizer(optimizer, rows)
    return params, (torch.cat((tmp_radii, extra[0])) if tmp_radii is not None else None)


def densify_and_split(optimizer, selected_pts_mask, N=2, tmp_radii=None):
    """`GaussianModel.densify_and_split` (gaussian_model.py:573-623) on the optimizer's groups: every selected Gaussian is
    replaced by N samples of itself (positions drawn from it with `torch.normal`, scales divided by 0.8 N) and then
    pruned. The random draw is `torch.normal(mean=zeros, std=stds)` as in the reference, so the same generator state gives
    the same samples. Activations as the reference's model: scaling = exp(_scaling), its inverse log.
    Returns ({name: new Parameter}, new tmp_radii or None, keep mask over the intermediate [old ++ new] rows)."""
    names = [g["name"] for g in optimizer.param_groups]
    sel, extra = _selected_rows(optimizer, selected_pts_mask, names, () if tmp_radii is None else (tmp_radii,))
    scaling = torch.exp(sel["scaling"])
    stds = scaling.repeat(N, 1)
    means = torch.zeros((stds.size(0), 3), device=stds.device)
    samples = torch.normal(mean=means, std=stds)
    rots = build_rotation(sel["rotation"]).repeat(N, 1, 1)
    new = {
        "xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + sel["xyz"].repeat(N, 1),
        "scaling": torch.log(scaling.repeat(N, 1) / (0.8 * N)),
        "rotation": sel["rotation"].repeat(N, 1),
        "opacity": sel["opacity"].repeat(N, 1),
    }
    for n in names:
        if n not in new:  # f_dc [K,1,3], f_rest [K,M,3], anything else a model carries per Gaussian
            new[n] = sel[n].repeat(N, *([1] * (sel[n].ndim - 1)))
    cat_tensors_to_optimizer(optimizer, new)
    radii = torch.cat((tmp_radii, extra[0].repeat(N))) if tmp_radii is not None else None
    # prune the originals (prune_filter = cat(selected, zeros(N K)), gaussian_model.py:616-623): one compaction
    n_new = N * int(sel["xyz"].shape[0])
    keep = torch.cat((~selected_pts_mask, torch.ones(n_new, dtype=torch.bool, device=selected_pts_mask.device)))
    params, _ = prune_optimizer(optimizer, keep)
    return params, radii, keep


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


RETIRED_LOGIT = -1.0e30  # sigmoid() of it is exactly 0 in fp32; Adam's bounded updates cannot move it


def retire_rows(optimizer, keep, name="opacity"):
    """Deferred form of the transparent-Gaussian prune (train_pan.py:673-678 runs `prune_points` every iteration): the rows
    NOT marked in `keep` get the opacity logit RETIRED_LOGIT instead of being removed. A retired Gaussian has opacity 0: the
    rasterizer lists it in no tile (alpha < 1/255 everywhere, forward.cu:374-376), it receives zero gradients, and because
    the stable compaction of `prune_optimizer` would have kept the survivors in the same relative order — the tie-break of
    the depth sort — the renders, the gradients and the Adam updates of the survivors are those of the pruned model, bit
    for bit. No tensor changes shape or address and nothing waits for the device: a recorded HIP graph of the iteration
    (`eogs2_amd.graph.GraphedStep`) keeps replaying, where a prune forces a new recording. Remove the retired rows for good
    with `prune_optimizer(optimizer, alive_rows(optimizer))` at a coarser interval — before anything that reads all rows
    (`reset_opacity`, densification, saving the model)."""
    for group in optimizer.param_groups:
        if group["name"] == name:
            p = group["params"][0]
            if keep.numel() != p.numel():
                raise ValueError(f"retire_rows: mask of {keep.numel()} rows for {p.numel()} Gaussians")
            with torch.no_grad():
                p.view(-1).masked_fill_(~keep.reshape(-1).to(device=p.device, dtype=torch.bool), RETIRED_LOGIT)
            return
    raise KeyError(f"retire_rows: no parameter group named {name!r}")


def alive_rows(optimizer, name="opacity"):
    """Boolean mask of the rows `retire_rows` has not retired (a device tensor: no wait)."""
    for group in optimizer.param_groups:
        if group["name"] == name:
            return group["params"][0].detach().view(-1) > 0.5 * RETIRED_LOGIT
    raise KeyError(f"alive_rows: no parameter group named {name!r}")


__all__ = ["FusedAdam", "compact_rows", "prune_optimizer", "reset_opacity", "cat_tensors_to_optimizer", "densify_and_clone",
           "densify_and_split", "build_rotation", "retire_rows", "alive_rows", "RETIRED_LOGIT"]
