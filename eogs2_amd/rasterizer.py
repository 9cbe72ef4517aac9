"""Host side of the drop-in rasterizer: the reference's Python interface over the C-ABI.

Mirrors DGR/diff_gaussian_rasterization/__init__.py (DGR/ =
src/gaussiansplatting/submodules/diff-gaussian-rasterization/ in the reference):

* `GaussianRasterizationSettings`  — __init__.py:219-232 (same 13 fields, same order)
* `rasterize_gaussians`            — __init__.py:26-48 (viewmatrix passed as differentiable input)
* `_RasterizeGaussians`            — __init__.py:51-216 (10 inputs, 3 outputs, 10 grads)
* `GaussianRasterizer`             — __init__.py:235-300 (forward + markVisible, same exceptions)

and the torch marshalling of DGR/rasterize_points.cu:35-245 (shape check, output
allocation, the three opaque workspaces), done here with `torch.empty` because the
C-ABI library never allocates device memory.

PyTorch is plumbing only (device memory, streams, autograd wiring): every
arithmetic step runs in the HIP library reached through `_lib.get()`. There is no
CPU fallback; a missing library raises.
"""
import ctypes
import os
import time
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib
from ._abi import (FLAG_ALT_ONLY, FLAG_ANTIALIASING, FLAG_DEBUG, FLAG_DEFER_COUNTS, FLAG_NO_READBACK, FLAG_RAW_PARAMS, MIRROR_BYTES,
                   RastError)

NUM_CHANNELS = 5  # DGR/cuda_rasterizer/config.h:15


def _backend():
    return _lib.get()


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool


def _flags(rs):
    return (FLAG_ANTIALIASING if rs.antialiasing else 0) | (FLAG_DEBUG if rs.debug else 0)


def _f32(t, dev):
    """Contiguous fp32 view on `dev` of a non-empty tensor, else None (-> NULL)."""
    if t is None or t.numel() == 0:
        return None
    if t.device != dev:
        raise RuntimeError(f"rasterizer input on {t.device}, expected {dev}")
    if t.dtype == torch.float32 and t.is_contiguous():
        return t  # the common case: only the data pointer is used, nothing to convert
    return t.detach().to(torch.float32).contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _Ctx:
    """device guard + stream for one call"""

    def __init__(self, abi, dev):
        self.abi = abi
        if dev.type != abi.device_type:
            raise RuntimeError(
                f"rasterizer tensors live on '{dev.type}' but the loaded library ({abi.backend}) works on "
                f"'{abi.device_type}' memory; there is no CPU fallback"
            )
        self.dev = dev
        self.guard = torch.cuda.device(dev) if dev.type == "cuda" else None

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()
            self.stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        else:
            self.stream = None
        return self

    def __exit__(self, *a):
        if self.guard is not None:
            self.guard.__exit__(*a)


_scratch_lock = threading.Lock()
_scratch = {}  # (device, stream) -> uint8 tensor

# Deferred count readback (include/eogs_rast.h EOGS_FLAG_DEFER_COUNTS): the exact token of the last forward per shape
SPECULATION_SLACK = 0.25
_speculate = os.environ.get("EOGS_SPECULATE", "0") == "1"
_spec = {}  # (device, P, H, W, raw) -> exact num_rendered token of the previous forward of that shape
_spec_stats = {"exact": 0, "hit": 0, "redo": 0}
_last_exact = {}  # device -> exact token of the last forward (the token a forward returns may be a capacity)


def set_speculation(on, forget=False):
    """Turns the deferred count readback on or off (default off; EOGS_SPECULATE=1 in the environment turns it on); returns
    the previous setting. Off: every forward waits for its counts before it queues binning and blending, as the
    reference does. (In an eager loop the wait only moves from the middle of the forward to its end — no step got faster on
    MI355X, profiles/r03_host_path.txt — so it is off by default.) `forget` drops the remembered counts, so the next
    forward of every shape waits for its own."""
    global _speculate
    old, _speculate = _speculate, bool(on)
    if forget:
        _spec.clear()
    return old


def speculation_stats(reset=False):
    """{"exact": forwards that waited for their counts, "hit": forwards queued whole, "redo": forwards repeated because the
    workspace guessed from the previous one was too small}"""
    out = dict(_spec_stats)
    if reset:
        for k in _spec_stats:
            _spec_stats[k] = 0
    return out


# Forwards recorded into a HIP graph (torch.cuda.graph): no readback at all inside the capture (EOGS_FLAG_NO_READBACK); the
# workspaces hold GRAPH_SLACK more than the largest counts any eager forward of the shape has had, and whoever replays the
# graph checks afterwards that this was enough (eogs2_amd/graph.py).
GRAPH_SLACK = 0.25
_peak = {}  # (device, P, H, W, raw) -> token with the largest slot and entry counts seen, flags of the latest forward
_recording = None  # the record_captured scope of the capture in progress
_SLOTS, _ENTRIES = 0x7FFFFFFF, 0x07FFFFFF << 32  # csrc/common.h nr_slots / nr_entries


def _merge_counts(peak, exact):
    if peak is None:
        return exact
    return (exact & ~(_SLOTS | _ENTRIES)) | max(peak & _SLOTS, exact & _SLOTS) | max(peak & _ENTRIES, exact & _ENTRIES)


class CapturedForward:
    """One forward inside a captured graph: after a replay `fits()` obtains this forward's counts and tells whether the
    capacity recorded with it held them; a forward that did not fit has rendered the background. With a `mirror` (a slot
    of pinned host memory the graph copies the counts into, a few kernels into the forward) fits() polls that slot and
    returns while the rest of the graph is still running; without, it waits for the stream (eogs_rast_read_counts)."""

    POLL_SECONDS = 0.05  # then wait for the stream instead (after which the copy has landed by definition)

    def __init__(self, abi, key, geom, capacity, have_scratch, mirror=None):
        # have_scratch: bit 0 = a scratch buffer went with the forward, bit 1 = an altitude-only forward (include/eogs_rast.h)
        self.abi, self.key, self.geom, self.capacity, self.have_scratch = abi, key, geom, capacity, int(have_scratch)
        self.mirror = mirror  # ctypes.c_void_p or None

    def arm(self):
        if self.mirror is not None:
            self.abi.check(self.abi.mirror_arm(self.mirror))

    def fits(self):
        dev, P, H, W = self.key[:4]
        R, arrived = ctypes.c_int64(), ctypes.c_int(0)
        if self.mirror is not None:
            deadline = time.perf_counter() + self.POLL_SECONDS
            while True:
                self.abi.check(self.abi.mirror_token(P, H, W, self.mirror, int(self.have_scratch), ctypes.byref(R), ctypes.byref(arrived)))
                if arrived.value:
                    break
                if time.perf_counter() > deadline:
                    torch.cuda.current_stream(dev).synchronize()
                    self.abi.check(self.abi.mirror_token(P, H, W, self.mirror, int(self.have_scratch), ctypes.byref(R), ctypes.byref(arrived)))
                    break
        if not arrived.value:
            with torch.cuda.device(dev):
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                self.abi.check(self.abi.read_counts(P, H, W, _ptr(self.geom), self.geom.numel(), int(self.have_scratch), stream,
                                                    ctypes.byref(R)))
        exact = R.value
        _peak[self.key] = _merge_counts(_peak.get(self.key), exact)
        _spec[self.key] = _last_exact[dev] = exact
        return (exact & _SLOTS) <= (self.capacity & _SLOTS) and (exact & _ENTRIES) <= (self.capacity & _ENTRIES)


class record_captured:
    """Context manager around a graph capture: collects the CapturedForward of every rasterizer forward recorded inside.
    `mirror`: pinned uint8 tensor (allocated before the capture) whose 64-byte slots receive the forwards' counts."""

    def __init__(self, mirror=None):
        self.slots = []
        if mirror is not None:
            if not mirror.is_pinned() or mirror.dtype != torch.uint8:
                raise ValueError("record_captured: mirror must be a pinned uint8 tensor")
            p0 = (mirror.data_ptr() + 63) // 64 * 64
            n = (mirror.data_ptr() + mirror.numel() - p0) // MIRROR_BYTES
            self.slots = [ctypes.c_void_p(p0 + i * MIRROR_BYTES) for i in range(max(0, n))]
        self.forwards = []

    def take_slot(self):
        return self.slots.pop(0) if self.slots else None

    def __enter__(self):
        global _recording
        self.prev, _recording = _recording, self
        return self.forwards

    def __exit__(self, *a):
        global _recording
        _recording = self.prev


def last_exact_token(device):
    """The exact num_rendered token of the last forward on `device` (what bench.py reports as counts)."""
    return _last_exact.get(torch.device(device) if not isinstance(device, torch.device) else device)


def _scratch_for(abi, dev, stream_id, P, H, W):
    """The transient entry-sort buffer of a forward (include/eogs_rast.h `scratch`): one per device and stream, grown on
    demand, never saved for backward (work on a stream is ordered, so the next forward may overwrite it)."""
    n = ctypes.c_size_t()
    abi.check(abi.scratch_bytes(P, H, W, ctypes.byref(n)))
    if n.value == 0:
        return None
    key = (dev, stream_id)
    with _scratch_lock:
        t = _scratch.get(key)
        # grown on demand; given back when the scene shrank to less than half of it (pruning) instead of staying at its peak
        if t is None or t.numel() < n.value or t.numel() > 2 * (n.value + n.value // 4):
            t = _scratch[key] = torch.empty((n.value + n.value // 4,), dtype=torch.uint8, device=dev)
    return t


def clear_scratch(device=None, stream=None):
    """Drops cached entry-sort buffers (192 B per Gaussian + 25 %, one per device and stream that has run a forward): all of
    them, those of one device, or the one of a stream. They are transient — the next forward on a stream allocates its own
    again — so this is safe whenever no forward is being queued concurrently."""
    with _scratch_lock:
        for key in list(_scratch):
            dev, sid = key
            if device is not None and torch.device(device) != dev:
                continue
            if stream is not None and (stream.device != dev or stream.cuda_stream != sid):
                continue
            del _scratch[key]


def rasterize_gaussians(
    means3D,
    means2D,
    sh,
    colors_precomp,
    opacities,
    scales,
    rotations,
    cov3Ds_precomp,
    raster_settings,
):
    return _RasterizeGaussians.apply(
        means3D,
        means2D,
        sh,
        colors_precomp,
        opacities,
        scales,
        rotations,
        cov3Ds_precomp,
        raster_settings.viewmatrix,
        raster_settings,
    )


def _run_forward(rs, viewmat, means3D, colors, opacities, scales, rotations, cov3Ds_precomp, alt_affine=None, raw=False,
                 alt_only=False, want_invdepth=True):
    """Marshalling of DGR/rasterize_points.cu:35-131 over the C-ABI.

    Returns (num_rendered, color, radii, invdepths, geom, binning, img). With `raw` the per-Gaussian tensors are the
    model's raw parameters (EOGS_FLAG_RAW_PARAMS, include/eogs_rast.h) and `colors` is f_dc [P,3]. With `alt_only`
    (EOGS_FLAG_ALT_ONLY) only feature channel 3 is rendered: `color` is [1, H, W] and `invdepths` is None. With
    `want_invdepth=False` the inverse-depth image is neither allocated nor blended (out_invdepth = NULL): `invdepths` is None.
    """
    abi = _backend()
    # DGR/rasterize_points.cu:58-60
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    flags = _flags(rs) | (FLAG_RAW_PARAMS if raw else 0) | (FLAG_ALT_ONLY if alt_only else 0)
    ncol = 3 if raw else NUM_CHANNELS

    with _Ctx(abi, dev) as cx:
        # outputs as DGR/rasterize_points.cu:69-76 (zero images when P == 0: forward is skipped)
        color = torch.empty((1 if alt_only else NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
        # (alt_only: no such output; want_invdepth=False: the caller drops it, as the reference's render() does)
        invdepths = None if (alt_only or not want_invdepth) else torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        empty_u8 = torch.empty((0,), dtype=torch.uint8, device=dev)
        geom = binning = img = empty_u8
        num_rendered = 0
        if P == 0:
            color.zero_()
            if invdepths is not None:
                invdepths.zero_()
        else:
            # DGR/cuda_rasterizer/rasterizer_impl.cu:244-247
            if colors is None or colors.numel() == 0:
                raise RuntimeError("For non-RGB, provide precomputed Gaussian colors!")
            if colors.shape[0] != P or colors.shape[-1] != ncol or colors.numel() != P * ncol:
                raise RuntimeError(f"colors_precomp must have dimensions (num_points, {ncol})")
            m3 = _f32(means3D, dev)
            col = _f32(colors, dev)
            opa = _f32(opacities, dev)
            sc = _f32(scales, dev)
            rot = _f32(rotations, dev)
            cov = _f32(cov3Ds_precomp, dev)
            vm = _f32(viewmat, dev)
            pm = _f32(rs.projmatrix, dev)
            bg = _f32(rs.bg, dev)
            alt = _f32(alt_affine, dev)
            if opa is None or opa.numel() != P:
                raise RuntimeError("opacities must have num_points elements")
            if bg is None or bg.numel() != NUM_CHANNELS:
                raise RuntimeError(f"bg must have {NUM_CHANNELS} elements")
            if raw and (alt is None or alt.numel() != 4 or sc is None or rot is None):
                raise RuntimeError("raw-parameter mode needs log-scales, raw rotations and a 4-element alt_affine")

            nbytes = ctypes.c_size_t()
            abi.check(abi.geom_bytes(P, ctypes.byref(nbytes)))
            geom = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
            abi.check(abi.image_bytes(H, W, ctypes.byref(nbytes)))
            img = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)

            capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
            if capturing:  # a buffer of the graph's own pool (the cached one belongs to eager work on another stream)
                abi.check(abi.scratch_bytes(P, H, W, ctypes.byref(nbytes)))
                scratch = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev) if nbytes.value else None
            else:
                scratch = _scratch_for(abi, dev, cx.stream.value if cx.stream is not None else 0, P, H, W)
            n_scratch = 0 if scratch is None else scratch.numel()
            R = ctypes.c_int64()

            def prepare(extra_flags):
                abi.check(
                    abi.forward_prepare(
                        P, H, W, _ptr(m3), _ptr(sc), _ptr(rot), _ptr(cov), _ptr(opa), _ptr(col),
                        float(rs.scale_modifier), _ptr(vm), _ptr(pm), _ptr(alt), flags | extra_flags,
                        _ptr(radii), _ptr(geom), geom.numel(), _ptr(scratch), n_scratch, ctypes.byref(R), cx.stream,
                    )
                )

            def render(token):
                abi.check(abi.binning_bytes(P, H, W, token, ctypes.byref(nbytes)))
                ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
                abi.check(
                    abi.forward_render(
                        P, H, W, token, _ptr(bg), flags,
                        _ptr(geom), geom.numel(), _ptr(ws), ws.numel(), _ptr(img), img.numel(),
                        _ptr(scratch), n_scratch, _ptr(color), None if invdepths is None else _ptr(invdepths), cx.stream,
                    )
                )
                return ws

            # The reference blocks on the count readback in the middle of every forward (rasterizer_impl.cu:284), and so
            # does the first forward of a shape here. Later forwards of the same shape size the binning workspace from the
            # previous one's counts plus slack and queue the whole forward before asking for the counts
            # (EOGS_FLAG_DEFER_COUNTS): the device builds no lists when the guess does not hold them, and the forward is then
            # repeated with the exact counts. Which way a forward went changes the kernel variants it runs (a capacity token
            # keeps the earlier forward's list granularity), never what it computes, with one guarded exception: a forward
            # that needs the back-to-front backward does not fit a token counted without it and is redone.
            key = (dev, P, H, W, bool(raw), bool(alt_only))
            hs = lambda: int(scratch is not None) | (2 if alt_only else 0)  # `have_scratch` of the token-building calls
            last = _spec.get(key) if (_speculate and abi.backend != "cpu-oracle") else None
            if capturing:
                # Recorded into a graph: nothing may wait. The workspaces hold GRAPH_SLACK more than any eager forward of
                # this shape has needed; the replaying side checks each replay (CapturedForward.fits, eogs2_amd/graph.py).
                peak = _peak.get(key)
                if peak is None:
                    raise RuntimeError("rasterizer forward inside a graph capture: run the same step once outside the "
                                       "capture first (its counts size the captured workspaces)")
                if flags & FLAG_DEBUG:
                    raise RuntimeError("debug=True waits for the stream after every kernel: not inside a graph capture")
                cap = ctypes.c_int64()
                abi.check(abi.capacity_token(P, peak, GRAPH_SLACK, hs(), 0, ctypes.byref(cap), None))
                prepare(FLAG_DEFER_COUNTS | FLAG_NO_READBACK)
                exact = num_rendered = cap.value
                binning = render(num_rendered)
                if _recording is not None:
                    slot = _recording.take_slot()
                    if slot is not None:  # the counts reach the host while the rest of the graph runs
                        abi.check(abi.mirror_counts(P, _ptr(geom), geom.numel(), slot, cx.stream))
                    _recording.forwards.append(CapturedForward(abi, key, geom, num_rendered, hs(), slot))
            elif last is None:
                prepare(0)
                exact = num_rendered = R.value
                binning = render(num_rendered)
                _spec_stats["exact"] += 1
            else:
                prepare(FLAG_DEFER_COUNTS)
                cap, fits = ctypes.c_int64(), ctypes.c_int()
                abi.check(abi.capacity_token(P, last, SPECULATION_SLACK, hs(), 0, ctypes.byref(cap), None))
                binning = render(cap.value)
                abi.check(abi.forward_counts(ctypes.byref(R)))
                exact = R.value
                abi.check(abi.capacity_token(P, last, SPECULATION_SLACK, hs(), exact, ctypes.byref(cap), ctypes.byref(fits)))
                if fits.value:
                    num_rendered = cap.value
                    _spec_stats["hit"] += 1
                else:
                    num_rendered = exact
                    binning = render(exact)
                    _spec_stats["redo"] += 1
            if not capturing:
                _spec[key] = exact
                _peak[key] = _merge_counts(_peak.get(key), exact)
            _last_exact[dev] = exact
    return num_rendered, color, radii, invdepths, geom, binning, img


class BackwardPlan:
    """How the next backward hands over its per-Gaussian gradients (used by eogs2_amd.parallel.GradBucket).

    `alloc(name, shape, device)` may return the tensor a gradient is written into (name in "means3D", "colors",
    "opacities", "scales", "rotations"; None -> torch.empty); for "colors_lead" (activated inputs only) it may return a
    contiguous [P, k] tensor, k <= 5, that receives the first k columns of the colour gradient a second time. With `chunks` > 1 the per-Gaussian pass runs over that
    many ascending Gaussian ranges (eogs_rast_backward_range) and `on_chunk(i, p0, p1, grads)` is called right after
    range i has been queued on the stream — the caller starts its collective on those rows while the next range
    computes. A plan is consumed by ONE backward."""

    chunks = 1

    def alloc(self, name, shape, device):
        return None

    def on_chunk(self, i, p0, p1, grads):
        pass


_plan_lock = threading.Lock()
_plan = None


def set_backward_plan(plan):
    """Installs `plan` for the next rasterizer backward on any thread (autograd runs backward on its own thread);
    None removes it. Returns the previous plan."""
    global _plan
    with _plan_lock:
        old, _plan = _plan, plan
    return old


def _take_plan():
    global _plan
    with _plan_lock:
        plan, _plan = _plan, None
    return plan


def chunk_ranges(P, chunks):
    """Ascending Gaussian ranges covering [0, P) whose inner bounds are multiples of 256 (the per-Gaussian workgroup)."""
    chunks = max(1, min(int(chunks), (P + 255) // 256))
    per = ((P + chunks - 1) // chunks + 255) // 256 * 256
    out, p0 = [], 0
    while p0 < P:
        out.append((p0, min(P, p0 + per)))
        p0 += per
    return out or [(0, 0)]


def _run_backward(rs, num_rendered, grad_out_color, grad_out_depth, means3D, colors, opacities, scales, rotations,
                  cov3Ds_precomp, radii, geom, binning, img, color, invdepths, want_vm, alt_affine=None, raw=False,
                  alt_only=False):
    """Marshalling of DGR/rasterize_points.cu:133-224 over the C-ABI (P > 0).

    Returns (d_means2D, d_colors, d_opacity[P,1], d_means3D, d_cov3D|None, d_scales|None, d_rot|None, grad_viewmatrix|None).
    """
    abi = _backend()
    plan = _take_plan()
    dev = means3D.device
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    flags = _flags(rs) | (FLAG_RAW_PARAMS if raw else 0)
    grad_viewmatrix = torch.zeros_like(rs.viewmatrix) if want_vm else None

    with _Ctx(abi, dev) as cx:
        g_color = _f32(grad_out_color, dev)
        if g_color is None:
            g_color = torch.zeros((1 if alt_only else NUM_CHANNELS, H, W), **f32)
        if g_color.numel() != (1 if alt_only else NUM_CHANNELS) * H * W:
            raise RuntimeError("backward: the image gradient does not have the forward's shape")
        if alt_only and grad_out_depth is not None:
            raise RuntimeError("an altitude-only render has no inverse-depth output to differentiate")
        # the reference always receives a materialised (usually all-zero) invdepth gradient;
        # here an unused invdepth output arrives as None and its work is skipped
        g_depth = _f32(grad_out_depth, dev)
        have_sr = scales is not None and scales.numel() != 0

        def out(name, shape):  # a data-parallel plan may place a gradient inside its exchange buffer
            t = plan.alloc(name, shape, dev) if plan is not None else None
            if t is None:
                return torch.empty(shape, **f32)
            if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
                raise RuntimeError(f"backward plan: bad buffer for {name}")
            return t

        d_means2D = torch.empty((P, 3), **f32)
        d_colors = out("colors", (P, 3 if raw else NUM_CHANNELS))
        # a plan that exchanges only the leading (f_dc) columns of colors_precomp gets them written a second time,
        # contiguous, into its buffer (include/eogs_rast.h dL_dcolors_lead)
        lead = plan.alloc("colors_lead", (P, 3), dev) if (plan is not None and not raw) else None
        if lead is not None and (lead.dtype != torch.float32 or lead.device != dev or not lead.is_contiguous()
                                 or lead.ndim != 2 or lead.shape[0] != P or not 0 < lead.shape[1] <= NUM_CHANNELS):
            raise RuntimeError("backward plan: bad buffer for colors_lead")
        d_opacity = out("opacities", (P, 1))
        d_means3D = out("means3D", (P, 3))
        d_cov3D = None if (raw or have_sr) else torch.empty((P, 6), **f32)  # only returned for cov3D_precomp inputs
        d_scales = out("scales", (P, 3)) if have_sr else None
        d_rot = out("rotations", (P, 4)) if have_sr else None
        dT_sum = torch.empty((6,), **f32) if want_vm else None
        dvm_mean = torch.empty((12,), **f32) if want_vm else None

        grads = {"means3D": d_means3D, "colors": d_colors, "opacities": d_opacity, "scales": d_scales, "rotations": d_rot}
        args = (
            P, H, W, num_rendered,
            _ptr(_f32(rs.bg, dev)), _ptr(_f32(means3D, dev)), _ptr(radii), _ptr(_f32(colors, dev)),
            _ptr(_f32(opacities, dev)), _ptr(_f32(scales, dev)), _ptr(_f32(rotations, dev)),
            float(rs.scale_modifier), _ptr(_f32(cov3Ds_precomp, dev)),
            _ptr(_f32(rs.viewmatrix, dev)), _ptr(_f32(rs.projmatrix, dev)), _ptr(_f32(alt_affine, dev)), flags,
            _ptr(color), _ptr(invdepths), _ptr(g_color), _ptr(g_depth),
            _ptr(geom), geom.numel(), _ptr(binning), binning.numel(), _ptr(img), img.numel(),
            _ptr(d_means2D), _ptr(d_colors), _ptr(d_opacity), _ptr(d_means3D), _ptr(d_cov3D),
            _ptr(d_scales), _ptr(d_rot), _ptr(dT_sum), _ptr(dvm_mean), _ptr(lead), 0 if lead is None else int(lead.shape[1]),
        )
        if plan is None or plan.chunks <= 1:
            abi.check(abi.backward(*args, cx.stream))
            if plan is not None:
                plan.on_chunk(0, 0, P, grads)
        else:
            for i, (p0, p1) in enumerate(chunk_ranges(P, plan.chunks)):
                abi.check(abi.backward_range(*args, p0, p1, cx.stream))
                plan.on_chunk(i, p0, p1, grads)

        if want_vm:
            # DGR/diff_gaussian_rasterization/__init__.py:174-202, on the reduced sums.
            # (NCD2Screen @ dL_dT^T).sum(0): row k of the (3,2) result is scaled by NCD2Screen[k,k].
            with torch.no_grad():
                # (scalars, not a host-built tensor: a host-to-device copy is not allowed while a HIP graph is being recorded)
                dL_dA = dT_sum.view(2, 3).t().clone()
                dL_dA[0] *= W / 2
                dL_dA[1] *= H / 2
                grad_viewmatrix[:3, :2] += dL_dA.to(grad_viewmatrix.dtype)
                grad_viewmatrix[:3, :3] += dvm_mean[:9].view(3, 3).to(grad_viewmatrix.dtype)
                grad_viewmatrix[-1, :3] += dvm_mean[9:].to(grad_viewmatrix.dtype)
    return d_means2D, d_colors, d_opacity, d_means3D, d_cov3D, d_scales, d_rot, grad_viewmatrix


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(
        ctx,
        means3D,
        means2D,
        sh,
        colors_precomp,
        opacities,
        scales,
        rotations,
        cov3Ds_precomp,
        viewmat,
        raster_settings,
    ):
        rs = raster_settings
        num_rendered, color, radii, invdepths, geom, binning, img = _run_forward(
            rs, viewmat, means3D, colors_precomp, opacities, scales, rotations, cov3Ds_precomp
        )
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered  # layout of the workspaces (may be a capacity: see _run_forward)
        ctx.num_rendered_exact = last_exact_token(means3D.device) if num_rendered else 0
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        ctx.save_for_backward(
            colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
            geom, binning, img, color, invdepths,
        )
        return color, radii, invdepths

    @staticmethod
    def backward(ctx, grad_out_color, _, grad_out_depth):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
         geom, binning, img, color, invdepths) = ctx.saved_tensors
        dev = means3D.device
        P = means3D.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        want_vm = ctx.needs_input_grad[8]

        if P == 0:
            z = lambda *s: torch.zeros(s, **f32)
            return (z(0, 3), z(0, 3), None, z(0, NUM_CHANNELS), z(*opacities.shape), z(*scales.shape) if scales.numel() else None,
                    z(*rotations.shape) if rotations.numel() else None, None,
                    torch.zeros_like(rs.viewmatrix) if want_vm else None, None)

        have_sr = scales is not None and scales.numel() != 0
        d_means2D, d_colors, d_opacity, d_means3D, d_cov3D, d_scales, d_rot, grad_viewmatrix = _run_backward(
            rs, ctx.num_rendered, grad_out_color, grad_out_depth, means3D, colors_precomp, opacities, scales, rotations,
            cov3Ds_precomp, radii, geom, binning, img, color, invdepths, want_vm,
        )
        return (
            d_means3D,
            d_means2D,
            None,  # sh: always the empty tensor on this 5-channel build (DGR/cuda_rasterizer/rasterizer_impl.cu:244-247)
            d_colors,
            d_opacity.view(opacities.shape),
            d_scales,
            d_rot,
            d_cov3D if not have_sr else None,
            grad_viewmatrix,
            None,
        )


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # DGR/diff_gaussian_rasterization/__init__.py:240-248 -> DGR/rasterize_points.cu:226-245
        abi = _backend()
        with torch.no_grad():
            rs = self.raster_settings
            P = positions.shape[0]
            present = torch.empty((P,), dtype=torch.bool, device=positions.device)
            if P:
                with _Ctx(abi, positions.device) as cx:
                    abi.check(
                        abi.mark_visible(
                            P, _ptr(_f32(positions, positions.device)), _ptr(_f32(rs.viewmatrix, positions.device)),
                            _ptr(_f32(rs.projmatrix, positions.device)), _ptr(present), cx.stream,
                        )
                    )
        return present

    def forward(
        self,
        means3D,
        means2D,
        opacities,
        shs=None,
        colors_precomp=None,
        scales=None,
        rotations=None,
        cov3D_precomp=None,
    ):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")

        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None
        ):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")

        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])

        return rasterize_gaussians(
            means3D,
            means2D,
            shs,
            colors_precomp,
            opacities,
            scales,
            rotations,
            cov3D_precomp,
            raster_settings,
        )


__all__ = [
    "GaussianRasterizationSettings",
    "GaussianRasterizer",
    "rasterize_gaussians",
    "RastError",
    "NUM_CHANNELS",
    "BackwardPlan",
    "set_backward_plan",
    "chunk_ranges",
]
