"""View-sharded data parallelism for the rasterizer hot path.

The reference is single-GPU (one view per optimizer step, GS/train_pan.py:252-257). Renders of different
cameras are independent given the same Gaussians, so rank r renders view r with replicated parameters and
the only exchange step is a SUM all-reduce of the Gaussian parameter gradients: 56 B/Gaussian
(xyz 12 + f_dc 12 + opacity 4 + scaling 12 + rotation 16), i.e. one RCCL exchange per step
(`torch.distributed` backend "nccl" is RCCL on ROCm; over xGMI RCCL picks ring/tree/direct itself).
Densification statistics (GS/train_pan.py:683-690) are kept replica-identical with three tiny all-reduces.

Layout of the exchange buffer: ONE flat fp32 allocation holding one contiguous `[P, k]` block per parameter,
`[xyz | f_dc | opacity | scaling | rotation]`, each block 256-byte aligned. A block has exactly the layout of its
parameter's gradient, so

* the rasterizer's backward writes the gradients of full-width parameters straight into their blocks
  (`BackwardPlan.alloc`), `p.grad` IS a view of the buffer (no pack, no unpack, and `FusedAdam` consumes the
  views in place: they are contiguous), and
* the whole buffer is reduced by one collective, or — `chunks` > 1 — the per-Gaussian backward runs over
  ascending Gaussian ranges (`eogs_rast_backward_range`) and the rows of range i are all-reduced
  (5 coalesced pieces, one per block) on RCCL's stream while range i+1 is being computed.

One process per GPU; works unchanged on the gloo backend (CPU tests, world_size 2).
"""
import torch
import torch.distributed as dist

from .rasterizer import BackwardPlan, set_backward_plan

_ALIGN = 64  # floats: every block starts on a 256-byte boundary (the kernels store float4 rows)


def _dist_on(group=None):
    return dist.is_available() and dist.is_initialized()


def _copy_cols(block, g, cols, to_block):
    """block [rows, w] (contiguous) <-> columns `cols` of g [rows, k]: one launch of eogs_pack_columns
    (include/eogs_optim.h) on the GPU; plain tensor indexing for CPU tensors (gloo tests)."""
    lo, hi, _ = cols.indices(g.shape[1])
    if block.is_cuda and g.is_contiguous() and g.dtype == torch.float32 and block.shape[0] > 0:
        import ctypes

        from . import _lib
        from ._abi import PackTensor

        abi = _lib.get()
        arr = (PackTensor * 1)()
        arr[0].data, arr[0].width, arr[0].col0, arr[0].ncols = g.data_ptr(), g.shape[1], lo, hi - lo
        with torch.cuda.device(block.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(block.device).cuda_stream)
            abi.check(abi.pack_columns(block.shape[0], 1, ctypes.cast(arr, ctypes.c_void_p), ctypes.c_void_p(block.data_ptr()),
                                       hi - lo, 0 if to_block else 1, stream))
    elif to_block:
        block.copy_(g[:, cols])
    else:
        g[:, cols].copy_(block)


class GradBucket(BackwardPlan):
    """The exchange buffer of the Gaussian parameter gradients.

    params: leaf tensors with P rows ([P], [P, k], [P, 1, 3], ...).  cols: per parameter, the slice of its columns that are model parameters (default:
    all; `colors_precomp` carries f_dc in columns 0:3, the altitude / constant channels stay rank-local).
    names: per parameter, the name of the rasterizer gradient it receives ("means3D", "colors", "opacities",
    "scales", "rotations") — needed only for the overlapped path (`begin()` ... backward ... `finish()`).
    algo: "all_reduce" (one SUM all-reduce of the buffer) or "rs_ag" (reduce-scatter + all-gather of the same buffer,
    SURVEY.md 5; RCCL only, whole-buffer exchanges only — otherwise "all_reduce" is used). How the collectives are issued
    (one coalesced launch per Gaussian range or one call per block, which algorithm) is fixed HERE, from the backend,
    identically on every rank: nothing falls back in the middle of a step, where ranks could disagree.

    Two ways to use it, both ending with every `p.grad` holding the all-reduced sum:

    * `all_reduce()` after ANY number of backward passes and any other gradient sources (several renders per step,
      regularisers applied to the parameters: what the reference's iteration does, train_pan.py:270-316,469): gradients
      that are not already views of the buffer are copied into their blocks, one exchange, full-width parameters get
      the block back as `.grad` (a view).
    * `begin()` before a step whose parameter gradients come from ONE rasterizer backward and nothing else, `finish()`
      after it: the backward writes into the buffer and the collectives are started from inside it, range by range (see
      the module docstring). Anything else is an error, and is reported as one: a second gradient accumulation into a
      bucketed parameter while the bucket is armed raises at once (it would add into a buffer whose exchange is in
      flight), and `finish()` raises when a parameter's gradient is not what the backward handed over (another
      contribution was summed in by autograd: it would be dropped) — checked by value on the first `verify_steps`
      armed steps, by identity afterwards.
    """

    def __init__(self, params, cols=None, names=None, chunks=1, group=None, algo="all_reduce", verify_steps=1, async_whole=True):
        self.params = list(params)
        P = self.params[0].shape[0]
        assert all(p.ndim >= 1 and p.shape[0] == P for p in self.params)  # [P], [P, k], [P, 1, 3], ...
        self.P = P
        rowlen = [p.numel() // max(P, 1) if P else int(torch.tensor(p.shape[1:]).prod()) for p in self.params]
        self.cols = list(cols) if cols is not None else [slice(0, n) for n in rowlen]
        self.names = list(names) if names is not None else [None] * len(self.params)
        self.chunks = max(1, int(chunks))
        self.group = group
        self.rowlen = rowlen
        self.widths = [len(range(*c.indices(n))) for n, c in zip(rowlen, self.cols)]
        # a column subset is only meaningful for [P, k] parameters (colors_precomp); [P, 1, 3] etc. go in whole
        assert all(w == n or p.ndim == 2 for w, n, p in zip(self.widths, rowlen, self.params))
        # ... and only a LEADING subset can be written by the backward itself (dL_dcolors_lead)
        self._leading = [c.indices(n)[0] == 0 and c.indices(n)[2] == 1 for n, c in zip(rowlen, self.cols)]
        self.offsets, o = [], 0
        for w in self.widths:
            self.offsets.append(o)
            o = (o + P * w + _ALIGN - 1) // _ALIGN * _ALIGN
        on = _dist_on()
        self.world = dist.get_world_size(group) if on else 1
        backend = str(dist.get_backend(group)) if on else ""
        self.used = o
        o = (o + self.world * _ALIGN - 1) // (self.world * _ALIGN) * (self.world * _ALIGN)  # equal shards for rs_ag
        self.flat = torch.zeros(o, dtype=torch.float32, device=self.params[0].device)
        assert algo in ("all_reduce", "rs_ag")
        self.algo = algo if (backend == "nccl" and self.world > 1) else "all_reduce"
        # one grouped launch for the pieces of a Gaussian range: RCCL only (decided once, never per call)
        self._coalesce = backend == "nccl" and hasattr(dist, "_coalescing_manager")
        self._async_whole = bool(async_whole)  # whole-buffer exchange of the armed path: asynchronous handle, or issued in line
        self._shard = torch.empty(o // self.world, dtype=torch.float32, device=self.flat.device) if self.algo == "rs_ag" else None
        self._works = []
        self._armed = False      # begin() called, backward not yet seen
        self._consumed = False   # the armed backward ran through this plan
        self._placed = [False] * len(self.params)   # block i was written directly by the backward
        self._partial = {}       # i -> the backward's gradient tensor whose bucket columns must be copied back
        self._accum = [0] * len(self.params)        # gradient accumulations seen while armed
        self._verify_left = int(verify_steps)
        self._local = None       # first armed steps: copy of the buffer as the backward left it (finish() compares)
        self.exchanges = 0       # collectives issued so far (diagnostics)
        self._hooks = []
        for i, p in enumerate(self.params):
            if hasattr(p, "register_post_accumulate_grad_hook") and p.is_leaf and p.requires_grad:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(_p):
            if self._armed:
                self._accum[i] += 1
                if self._accum[i] > 1:
                    raise RuntimeError(
                        f"GradBucket: parameter {i} ({self.names[i]}) received a second gradient while the overlapped "
                        "exchange of the first was armed; steps with several backward passes use all_reduce()")
        return hook

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    # ---- layout ----
    @property
    def bytes_per_gaussian(self):
        return 4 * sum(self.widths)

    def _full(self, i):
        return self.widths[i] == self.rowlen[i]

    def block(self, i, p0=0, p1=None):
        """Rows [p0, p1) of block i as a contiguous [rows, width] view of the buffer."""
        p1 = self.P if p1 is None else p1
        w, o = self.widths[i], self.offsets[i]
        return self.flat[o + p0 * w:o + p1 * w].view(p1 - p0, w)

    def _is_block(self, g, i):
        b = self.block(i)
        return g is not None and g.data_ptr() == b.data_ptr() and g.numel() == b.numel() and g.is_contiguous()

    # ---- the exchange of the whole buffer ----
    def _exchange_whole(self, async_op):
        """SUM over the ranks of the whole buffer, by the algorithm chosen at construction. Returns the work handles
        (async) or an empty list."""
        works = []
        if self.algo == "rs_ag":
            w1 = dist.reduce_scatter_tensor(self._shard, self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            w2 = dist.all_gather_into_tensor(self.flat, self._shard, group=self.group, async_op=async_op)
            works = [w1, w2] if async_op else []
        else:
            w1 = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            works = [w1] if async_op else []
        self.exchanges += 1
        return works

    # ---- synchronous path ----
    def pack(self):
        """Brings every block up to date with its parameter's .grad. Decided per parameter: a gradient that already IS
        the block (a previous unpack / the overlapped path handed it out, and autograd accumulated into it in place) is
        left alone, everything else is copied (zeros for a missing gradient)."""
        for i, (p, c) in enumerate(zip(self.params, self.cols)):
            g = p.grad
            if self._is_block(g, i):
                continue
            b = self.block(i)
            if g is None:
                b.zero_()
            elif self._full(i):
                b.copy_(g.reshape(b.shape))
            else:
                _copy_cols(b, g, c, True)

    def unpack(self):
        """Full-width parameters get a VIEW of their block as .grad (no copy); partially bucketed ones
        (colors_precomp: only the f_dc columns are parameters) get their columns copied back."""
        for i, (p, c) in enumerate(zip(self.params, self.cols)):
            if self._full(i):
                if not self._is_block(p.grad, i):
                    p.grad = self.block(i).view(p.shape)
            else:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                _copy_cols(self.block(i), p.grad, c, False)

    def zero_(self):
        """Gradients that start at zero INSIDE the buffer: clears it and makes every parameter's .grad the view of its block
        (partially bucketed parameters: their own zeroed tensor). The backward passes of the step then accumulate in place and
        all_reduce() finds nothing to copy. This is what a recorded step needs (eogs2_amd.graph.GraphedStep): the same gradient
        tensors in every replay —

            def fwd_bwd(): bucket.zero_(); ...renders, losses...; loss.backward()
            step = GraphedStep(fwd_bwd)
            step(); bucket.all_reduce(); optimizer.step()
        """
        if self._armed:
            raise RuntimeError("GradBucket.zero_() while armed by begin(): call finish()")
        self.flat.zero_()
        for i, p in enumerate(self.params):
            if self._full(i):
                if not self._is_block(p.grad, i):
                    p.grad = self.block(i).view(p.shape)
            elif p.grad is None:
                p.grad = torch.zeros_like(p)
            else:
                p.grad.zero_()

    def all_reduce(self, average=False):
        if self._armed:
            raise RuntimeError("GradBucket.all_reduce() while armed by begin(): call finish()")
        self.pack()
        if _dist_on():
            self._exchange_whole(async_op=False)
            if average:
                self.flat.div_(self.world)
        self.unpack()

    # ---- overlapped path: BackwardPlan ----
    def begin(self):
        """Arms the bucket for a step whose parameter gradients come from ONE rasterizer backward: gradients are cleared
        (set to None) and the next backward writes into / exchanges from this buffer."""
        for p in self.params:
            p.grad = None
        self._works, self._partial = [], {}
        self._placed = [False] * len(self.params)
        self._accum = [0] * len(self.params)
        self._local = None
        self._armed, self._consumed = True, False
        set_backward_plan(self)

    def alloc(self, name, shape, device):
        self._consumed = True
        lead = name == "colors_lead"
        for i, n in enumerate(self.names):
            if device != self.flat.device:
                continue
            if not lead and n == name and self._full(i) and tuple(shape) == (self.P, self.widths[i]):
                self._placed[i] = True
                return self.block(i)  # a fresh view each time: autograd may adopt it as .grad without a copy
            if lead and n == "colors" and not self._full(i) and self._leading[i] and self.params[i].ndim == 2:
                self._placed[i] = True  # the backward writes the exchanged (leading) columns here itself
                return self.block(i)
        return None

    def on_chunk(self, i, p0, p1, grads):
        self._consumed = True
        views = []
        for k, (n, c) in enumerate(zip(self.names, self.cols)):
            b = self.block(k, p0, p1)
            g = grads.get(n) if n is not None else None
            if not self._placed[k]:
                if g is None:
                    b.zero_()
                elif self._full(k):
                    b.copy_(g[p0:p1].reshape(b.shape))
                else:
                    _copy_cols(b, g[p0:p1], c, True)
            if not self._full(k) and g is not None:
                self._partial[k] = g
            views.append(b.view(-1))
        whole = p0 == 0 and p1 == self.P
        if self._verify_left > 0:  # what the backward produced for these rows, before any other rank's data is added
            if self._local is None:
                self._local = torch.empty(self.used, dtype=torch.float32, device=self.flat.device)
            for k, (w, o) in enumerate(zip(self.widths, self.offsets)):
                self._local[o + p0 * w:o + p1 * w].copy_(self.flat[o + p0 * w:o + p1 * w])
        if not _dist_on():
            return
        if whole:
            self._works += self._exchange_whole(async_op=self._async_whole)
        elif self._coalesce:  # one grouped launch for the pieces of this range
            with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
                for v in views:
                    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            self._works.append(cm)
            self.exchanges += 1
        else:
            for v in views:
                self._works.append(dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.exchanges += 1

    def _check_sources(self):
        """The overlapped path is only right when every bucketed gradient is exactly what the backward wrote into the
        buffer. By identity: a full-width parameter's .grad must be the block view autograd adopted. By value (first
        armed steps, `verify_steps`): whatever .grad is, it must equal the local copy taken before the exchange."""
        for i, p in enumerate(self.params):
            if not self._placed[i] or p.grad is None:
                continue
            if self._full(i) and self._is_block(p.grad, i):
                continue
            if self._local is not None:
                w, o = self.widths[i], self.offsets[i]
                mine = self._local[o:o + self.P * w].view(self.P, w)
                got = p.grad.reshape(self.P, -1)[:, self.cols[i]] if not self._full(i) else p.grad.reshape(self.P, w)
                if torch.equal(got, mine):
                    continue  # autograd copied instead of adopting the view: same numbers, the reduced block replaces it
            elif not self._full(i):
                continue  # partial parameter: its own tensor by construction; verified by value on the first steps
            raise RuntimeError(
                f"GradBucket.finish(): the gradient of parameter {i} ({self.names[i]}) is not the one the rasterizer backward "
                "wrote into the exchange buffer — another loss term contributed to it and would be lost; use all_reduce() "
                "for steps whose parameters get gradients from more than the one rasterizer backward")

    def finish(self, average=False):
        """Waits for the exchanges started by the backward and leaves the all-reduced sums in every .grad. If the
        backward did not run through the plan (no rasterizer backward, or P == 0), falls back to `all_reduce()`."""
        if not self._armed:
            raise RuntimeError("GradBucket.finish() without begin()")
        self._armed = False
        leftover = set_backward_plan(None)
        if leftover is not None and leftover is not self:
            set_backward_plan(leftover)  # someone else's plan: not ours to drop
        if not self._consumed:
            return self.all_reduce(average=average)
        for w in self._works:
            w.wait()
        self._works = []
        self._check_sources()
        if self._local is not None:
            self._verify_left -= 1
            self._local = None
        if average and _dist_on():
            self.flat.div_(self.world)
        for i, (p, c) in enumerate(zip(self.params, self.cols)):
            if self._full(i):
                if not self._is_block(p.grad, i):
                    p.grad = self.block(i).view(p.shape)  # (autograd copied: verified above to hold the same numbers)
            else:
                g = p.grad if p.grad is not None else self._partial.get(i)
                if g is None:
                    g = torch.zeros_like(p)
                _copy_cols(self.block(i), g, c, False)
                p.grad = g
        self._partial = {}


def all_reduce_densification_stats(xyz_gradient_accum, denom, max_radii2D, group=None):
    """Sum / sum / max across replicas so prune, clone and split decisions stay identical on every rank
    (GS/scene/gaussian_model.py:719-723, GS/train_pan.py:679-690)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def shard_views(num_views, rank=None, world=None):
    """Round-robin view indices of this rank (independent views, no data-path collective)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return list(range(rank, num_views, world))


# Iteration-counted knobs of the reference's optimisation config (gs_config/train.yaml:87-130) that fire after a number of VIEWS.
# `densify_from_iter` / `densification_interval` sit one level down (optimization.densification_strategy), and every loss
# term has an `iterstart_*` / `iterend_*` threshold (iterstart_shadowmapping: 1000, iterstart_L_new_resample: 1000, ...).
_PER_VIEW_INTERVALS = ("iterations", "position_lr_max_steps", "densify_from_iter", "densify_until_iter",
                       "densification_interval", "opacity_reset_interval", "color_reset_iterations")
_PER_VIEW_PREFIXES = ("iterstart_", "iterend_")
_LEARNING_RATES = ("position_lr_init", "position_lr_final", "feature_lr", "opacity_lr", "scaling_lr", "rotation_lr")
# what a complete `optimization` section of the reference holds: a settings dict without them is probably not one
_EXPECTED = ("iterations", "position_lr_init", "opacity_reset_interval", "densify_from_iter", "densification_interval")


def view_sharded_schedule(opt, world=None, strict=False):
    """The "equal views seen" protocol for training with `world` views per optimizer step: returns a copy of the
    optimisation settings `opt` (a dict with the reference's key names, nested sections such as `densification_strategy`
    included) in which every interval or threshold counted in iterations — the listed keys and every `iterstart_*` /
    `iterend_*` key — is divided by `world` (at least 1), every learning rate is multiplied by sqrt(world), plus
    `grad_average: True` — the exchanged gradient is the mean over the views (`GradBucket.all_reduce(average=True)` /
    `finish(average=True)`). world == 1 returns the settings unchanged. Values of 0 or below (disabled) are kept, and so
    are the reference's "never" sentinels (>= 9,999,999). A settings dict that lacks the reference's core keys is
    reported (warning; RuntimeError with strict=True) instead of being left silently unscaled."""
    import copy
    import math
    import warnings

    world = (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1) if world is None else int(world)
    out = copy.deepcopy(dict(opt))
    out["grad_average"] = world > 1
    seen = set()

    def is_num(v):
        return isinstance(v, (int, float)) and not isinstance(v, bool)

    def walk(d):
        for k, v in list(d.items()):
            if isinstance(v, dict):
                walk(v)
                continue
            seen.add(k)
            if world <= 1 or not is_num(v):
                continue
            if k in _PER_VIEW_INTERVALS or k.startswith(_PER_VIEW_PREFIXES):
                if 0 < v < 9_999_999:
                    d[k] = max(1, int(round(v / world)))
            elif k in _LEARNING_RATES:
                d[k] = v * math.sqrt(world)

    walk(out)
    missing = [k for k in _EXPECTED if k not in seen] if world > 1 else []  # (world 1 scales nothing: nothing to miss)
    if missing:
        msg = f"view_sharded_schedule: settings lack {missing}: not the reference's `optimization` section? (those stay unscaled)"
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg)
    return out
