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

    params: [P, k] leaf tensors.  cols: per parameter, the slice of its columns that are model parameters (default:
    all; `colors_precomp` carries f_dc in columns 0:3, the altitude / constant channels stay rank-local).
    names: per parameter, the name of the rasterizer gradient it receives ("means3D", "colors", "opacities",
    "scales", "rotations") — needed only for the overlapped path (`begin()` ... backward ... `finish()`).

    Two ways to use it, both ending with every `p.grad` holding the all-reduced sum:

    * `all_reduce()` after any number of backward passes: gradients that are not already views of the buffer are
      copied into their blocks, one collective, full-width parameters get the block back as `.grad` (a view).
    * `begin()` before the step's ONLY rasterizer backward, `finish()` after it: the backward writes into the buffer
      and the collectives are started from inside it, range by range (see the module docstring).
    """

    def __init__(self, params, cols=None, names=None, chunks=1, group=None):
        self.params = list(params)
        P = self.params[0].shape[0]
        assert all(p.ndim >= 2 and p.shape[0] == P for p in self.params)
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
        self.offsets, o = [], 0
        for w in self.widths:
            self.offsets.append(o)
            o = (o + P * w + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = torch.zeros(o, dtype=torch.float32, device=self.params[0].device)
        self._works = []
        self._armed = False      # begin() called, backward not yet seen
        self._consumed = False   # the armed backward ran through this plan
        self._placed = [False] * len(self.params)   # block i was written directly by the backward
        self._partial = {}       # i -> the backward's gradient tensor whose bucket columns must be copied back
        self.exchanges = 0       # collectives issued so far (diagnostics)

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

    def all_reduce(self, average=False):
        self.pack()
        if _dist_on():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.exchanges += 1
            if average:
                self.flat.div_(dist.get_world_size(self.group))
        self.unpack()

    # ---- overlapped path: BackwardPlan ----
    def begin(self):
        """Arms the bucket for the step's only rasterizer backward: gradients are cleared (set to None) and the next
        backward writes into / exchanges from this buffer."""
        for p in self.params:
            p.grad = None
        self._works, self._partial = [], {}
        self._placed = [False] * len(self.params)
        self._armed, self._consumed = True, False
        set_backward_plan(self)

    def alloc(self, name, shape, device):
        self._consumed = True
        for i, n in enumerate(self.names):
            if n == name and self._full(i) and tuple(shape) == (self.P, self.widths[i]) and device == self.flat.device:
                self._placed[i] = True
                return self.block(i)  # a fresh view each time: autograd may adopt it as .grad without a copy
        return None

    def on_chunk(self, i, p0, p1, grads):
        self._consumed = True
        views = []
        for k, (n, c) in enumerate(zip(self.names, self.cols)):
            b = self.block(k, p0, p1)
            if not self._placed[k]:
                g = grads.get(n) if n is not None else None
                if g is None:
                    b.zero_()
                elif self._full(k):
                    b.copy_(g[p0:p1].reshape(b.shape))
                else:
                    _copy_cols(b, g[p0:p1], c, True)
                    self._partial[k] = g
            views.append(b.view(-1))
        if not _dist_on():
            return
        whole = p0 == 0 and p1 == self.P
        if whole:
            self._works.append(dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            try:  # one grouped launch for the five pieces of this range
                with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
                    for v in views:
                        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
                self._works.append(cm)
            except (AttributeError, RuntimeError, ValueError):
                for v in views:
                    self._works.append(dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.exchanges += 1

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
        if average and _dist_on():
            self.flat.div_(dist.get_world_size(self.group))
        for i, (p, c) in enumerate(zip(self.params, self.cols)):
            if self._full(i):
                if not self._is_block(p.grad, i):
                    # autograd kept its own tensor (it copied, or a second contribution arrived): the reduced block wins
                    # only when it holds everything p.grad was built from, which begin()'s contract guarantees
                    p.grad = self.block(i).view(p.shape)
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


# Iteration-counted knobs of the reference's optimisation config (gs_config/train.yaml:87-116) that fire after a number of VIEWS
_PER_VIEW_INTERVALS = ("iterations", "position_lr_max_steps", "densify_from_iter", "densify_until_iter",
                       "densification_interval", "opacity_reset_interval", "iterend_opacity_reset_interval",
                       "color_reset_iterations")
_LEARNING_RATES = ("position_lr_init", "position_lr_final", "feature_lr", "opacity_lr", "scaling_lr", "rotation_lr")


def view_sharded_schedule(opt, world=None):
    """The "equal views seen" protocol for training with `world` views per optimizer step: returns a copy of the
    optimisation settings `opt` (a dict with the reference's key names) in which every interval counted in iterations is
    divided by `world` (at least 1), every learning rate is multiplied by sqrt(world), plus `grad_average: True` — the
    exchanged gradient is the mean over the views (`GradBucket.all_reduce(average=True)` / `finish(average=True)`).
    world == 1 returns the settings unchanged. Keys that are absent stay absent; values of 0 or below (disabled) are kept."""
    import math

    world = (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1) if world is None else int(world)
    out = dict(opt)
    out["grad_average"] = world > 1
    if world <= 1:
        return out
    for k in _PER_VIEW_INTERVALS:
        if k in out and out[k] is not None and out[k] > 0:
            out[k] = max(1, int(round(out[k] / world)))
    for k in _LEARNING_RATES:
        if k in out and out[k] is not None:
            out[k] = out[k] * math.sqrt(world)
    return out
