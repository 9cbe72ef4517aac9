"""A training / rendering step recorded once into a HIP graph and replayed (torch.cuda.CUDAGraph over hipGraph).

Small scenes are host-bound in an eager loop: at 100 k Gaussians / 512^2 the kernels of forward + backward need 0.17 ms, the
Python + autograd + launch path 0.31 ms (profiles/r03_host_path.txt). The reference cannot be captured at all — its forward
waits for `num_rendered` on the host to size the sort buffers (DGR/cuda_rasterizer/rasterizer_impl.cu:284,
DGR/rasterize_points.cu:35-131 resizes torch tensors from inside the call). Here a forward inside a capture only queues
kernels (EOGS_FLAG_DEFER_COUNTS | EOGS_FLAG_NO_READBACK, include/eogs_rast.h): its workspaces are sized from the counts the
same step had when it ran eagerly, plus slack, the device builds no lists when a replay outgrows them
(csrc/binning.hip block_lists_kernel), and `GraphedStep` reads the counts back after each replay and re-records the graph
with larger workspaces when that happened.

    step = GraphedStep(lambda: fwd_bwd(params, camera_buffers))   # eager warm-up runs, then one capture
    for it in range(n):
        camera_buffers.copy_(next_camera)                         # inputs live in fixed tensors, updated in place
        color, loss, *grads = step()                              # replay (+ check)
        optimizer.step()

What a recorded forward freezes besides its capacities is the list granularity (per-tile lists or 32-px block lists,
DESIGN.md §2.3) of the last eager forward of that shape: an eager loop re-decides it per forward, so near the switch the two
loops take different — equally valid, individually oracle-tested — kernel variants and their gradients differ at the
level of threshold flips (examples/train_synthetic.py at 200 k / 512²: identical loss curves with the switch disabled,
0.00815 against 0.00844 after 200 iterations with it). The choice of the back-to-front backward (image-sized opaque
Gaussians) is frozen the same way, with a guard: a replay whose forward asks for it while the graph was
recorded without it counts as not fitting (`CapturedForward.fits`) and is recorded again.

When the tensors `fn` reads are REPLACED (a prune gives every parameter a new shape and address) the step has to be recorded
again: `step.record_again()` does it inside the step's own memory pool — recording often then neither grows the process nor pays
for allocations — and `eogs2_amd.optim.retire_rows` avoids the replacement altogether.

`fn` follows the rules of torch.cuda.graph: it reads its inputs from tensors that exist before the capture, allocates
everything else itself, and never waits for the device. Gradients: set them to None at the start of `fn`
(`p.grad = None`), so the backward inside the capture writes fresh tensors instead of accumulating; a replay rewrites
exactly those tensors — p.grad keeps pointing at them as long as nothing else (an eager step, zero_grad(set_to_none=True))
replaces it; returning them from `fn` keeps a handle either way.
"""
import torch

from . import rasterizer


class Branches:
    """Independent pieces of one step — the renders of a training iteration (train_pan.py:278,305-316,375-391: the view, the
    sun camera, a random camera, all of the same Gaussians) — queued on streams of their own, forked from and joined back into
    the current stream. Recorded inside `GraphedStep` they become parallel branches of the HIP graph: one render's
    launch-floor kernels (count scan, histogram, column scan: a workgroup or sixteen) and the tail of its blend kernels
    (profiles/r04_wave_trace.txt: a fifth of a render launch runs at half occupancy) run beside another render's large
    ones.

    Gradients of parameters the pieces SHARE. A piece that calls `backward()` itself leaves its gradient in `p.grad` on the
    piece's own stream, and the next piece's `p.grad += g` runs on ANOTHER stream with nothing ordering the two: autograd
    binds a leaf's accumulation to the stream that was current when the leaf entered the piece's autograd graph, and that
    graph — and the binding with it — is gone when the piece returns (in a replayed graph with truly parallel
    branches a small render's `+=` can run before the large render has stored the buffer it adds to). `run` therefore takes
    the shared parameters explicitly: every piece starts from `p.grad = None`, so its backward only ever WRITES a gradient
    tensor of its own, and after the join the pieces' gradients are summed on the current stream in the order the pieces
    were queued — `g_0`, `+= g_1`, `+= g_2`, onto whatever `p.grad` held before: the arithmetic and the order of the serial
    loop, bit for bit, with every dependency an edge of the graph. Pieces that only run forwards (one `backward()` after
    the join: autograd orders that itself) pass `shared=()`.

        br = Branches(3)
        def fn():
            br.run([lambda v=v: render_and_backward(v) for v in views], shared=model_parameters)
    """

    def __init__(self, n, device=None):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n)]

    @staticmethod
    def _sum_fused(shared, before, parts):
        """The sum after the join as ONE launch for all parameters (eogs2_amd.optim.sum_into_: the same additions in the same
        order) when every piece left a dense fp32 gradient for every shared parameter; False -> the caller adds tensor by tensor."""
        if not shared or not parts or any(g is None for part in parts for g in part):
            return False
        every = [g for part in parts for g in part] + [b for b in before if b is not None]
        if any(not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and not g.is_sparse) for g in every):
            return False
        if any((b is None) != (before[0] is None) for b in before):
            return False
        from .optim import sum_into_

        if before[0] is None:
            dsts, srcs = parts[0], parts[1:]
        else:
            dsts, srcs = before, parts
        if any(d.numel() != s[k].numel() for s in srcs for k, d in enumerate(dsts)):
            return False
        if srcs:
            sum_into_(dsts, srcs)
        for p, d in zip(shared, dsts):
            p.grad = d
        return True

    def run(self, fns, *, shared):
        """fns: callables, one per branch, queued in this order. shared: the leaf tensors more than one piece differentiates
        (an empty sequence when no piece calls backward). Returns the pieces' results."""
        fns = list(fns)
        shared = list(shared)
        if len(fns) > len(self.streams):
            raise ValueError(f"Branches({len(self.streams)}) asked to run {len(fns)} pieces")
        cur = torch.cuda.current_stream()
        before = [p.grad for p in shared]
        outs, parts = [], []
        try:
            for s, fn in zip(self.streams, fns):
                s.wait_stream(cur)  # fork: the piece sees everything queued so far
                for p in shared:
                    p.grad = None  # the piece's backward writes a tensor of its own, it never adds into another piece's
                with torch.cuda.stream(s):
                    outs.append(fn())
                parts.append([p.grad for p in shared])
        except BaseException:
            # A piece raised: the gradients are what they were before the call — not the sum of the pieces that happened to finish
            # plus nothing of the one that failed. The streams are still joined: what was queued runs to its end.
            for s in self.streams[:len(fns)]:
                cur.wait_stream(s)
            for p, b in zip(shared, before):
                p.grad = b
            raise
        else:
            for s in self.streams[:len(fns)]:
                cur.wait_stream(s)  # join
            # The sum, after the join, in queue order. (Allocator note: a piece's gradient was allocated on the piece's stream
            # and is read here and by the optimizer on `cur`; its block can only be reused by a later allocation on that same
            # piece stream, and everything queued there comes after a fork that waits for `cur` — so no record_stream.)
            with torch.no_grad():
                fused = self._sum_fused(shared, before, parts)
                for k, p in enumerate(shared):
                    if fused:
                        break
                    acc = before[k]
                    for part in parts:
                        g = part[k]
                        if g is None:
                            continue
                        if acc is None:
                            acc = g
                        else:
                            acc.add_(g)
                    p.grad = acc
        return outs


class CapacityExceeded(RuntimeError):
    """A replay needed larger list workspaces than any recorded graph of this step had, twice in a row."""


class GraphedStep:
    MAX_MIRRORED = 16  # forwards per step whose counts are mirrored (further ones are read back after the replay)

    def __init__(self, fn, warmup=2, idempotent=True):
        """fn() -> anything (tensors it returns are the graph's output buffers, rewritten by every replay).
        warmup: eager runs before the capture (at least 1: their counts size the captured workspaces).
        idempotent: fn may simply be run again when a replay did not fit (true for forward + backward; false as soon as fn
        updates its own inputs, e.g. an optimizer step inside — then __call__ raises CapacityExceeded instead and the
        caller decides)."""
        if warmup < 1:
            raise ValueError("GraphedStep needs at least one eager run before the capture")
        self.fn, self.idempotent = fn, idempotent
        self.replays = self.recaptures = 0
        # pinned host slots the graph copies each forward's counts into (include/eogs_rast.h eogs_rast_mirror_counts):
        # fits() polls them instead of waiting for the whole replay
        self._mirror = torch.empty((64 * (self.MAX_MIRRORED + 1),), dtype=torch.uint8, pin_memory=True)
        # Every recording of this step allocates from ONE private pool: what the previous recording held is reused by the next
        # instead of going back to the device. (It also had to: a step re-created per recording lost 424 MiB of device memory
        # per recording of the 200 k / 512^2 example with parallel branches, the 288 GB card full after ~875 —
        # tools/graph_leak_probe2.py)
        self._pool = torch.cuda.graph_pool_handle()
        # The allocator drops a pool with its last graph and refuses to reopen it: a one-tensor graph that is never replayed
        # keeps this one open between dropping a recording and making the next. (Do not remove it as an optimisation: the
        # control runs of the probe show that this small graph, captured first, is also what makes a DISCARDED step's memory
        # return to the device in full on this stack.)
        self._anchor = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._anchor, pool=self._pool):
            self._anchor_out = torch.zeros(1, device="cuda")
        self._warm_up(warmup)
        self._capture()

    def _warm_up(self, runs):
        # The eager warm-up runs on a side stream (torch.cuda.graph's own recipe) and the capture on another: a leaf's
        # AccumulateGrad node, created by the warm-up's backward, then meets a gradient produced on the capture stream, and
        # PyTorch 2.10 warns about it on every such step ("AccumulateGrad node's stream does not match ..."). It is this
        # class's doing and harmless here — everything the capture queues is ordered by the capture itself, nothing
        # synchronises (a synchronisation would abort the capture) — so the warning is switched off, process-wide, the first
        # time a step is recorded (VERDICT r3-r5 listed it as noise in the suite's and the bench's output).
        setter = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if setter is not None:
            setter(False)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # (torch.cuda.graph's own recipe: warm up on a side stream)
            for _ in range(runs):
                self.fn()
        cur.wait_stream(side)
        rasterizer.clear_scratch(stream=side)  # (the warm-up stream is never used again: do not keep its entry-sort buffer)

    def record_again(self, warmup=1):
        """Records the step anew — after its inputs changed shape or address (a prune: new parameter tensors), which a replay
        cannot follow. `warmup` eager runs first, as at construction (their counts size the new workspaces; like those they
        really run `fn`: a step that updates its own inputs advances by them). Keeps this step's memory pool: prefer it to
        constructing a new GraphedStep."""
        if warmup < 1:
            raise ValueError("GraphedStep needs at least one eager run before the capture")
        torch.cuda.current_stream().synchronize()
        self.graph = self.outputs = None  # (their memory serves the eager runs' successor: the new recording)
        self.forwards = []
        self._warm_up(warmup)
        self._capture()

    def _capture(self):
        torch.cuda.current_stream().synchronize()  # (a replay of the graph being replaced may still be running)
        self.graph = None  # (frees the previous graph's pool before the new capture allocates)
        self.outputs = None
        self.forwards = []  # (each holds its graph's geometry workspace: the old pool is only released without them; the
        #                      counts they read are already merged into rasterizer._peak by fits())
        graph = torch.cuda.CUDAGraph()
        with rasterizer.record_captured(self._mirror) as forwards:
            with torch.cuda.graph(graph, pool=self._pool):
                out = self.fn()
        self.graph, self.outputs, self.forwards = graph, out, list(forwards)

    def replay(self):
        """Queues one replay; returns without waiting. Call fits() before trusting the results."""
        for f in self.forwards:
            f.arm()
        self.graph.replay()
        self.replays += 1

    def fits(self):
        """True when every forward of the last replay had room for its lists. Waits only until the forwards' counts have
        reached the host — each a few kernels into its forward — not for the replay to finish. (Also raises the forward's
        own errors — EOGS_ERR_ALTITUDE — which a captured forward cannot report in line.)"""
        ok = True
        for f in self.forwards:
            ok = f.fits() and ok  # (every forward is read: each updates the counts the next capture is sized from)
        return ok

    def __call__(self):
        self.replay()
        if self.fits():
            return self.outputs
        if not self.idempotent:
            self._capture()
            raise CapacityExceeded("the replayed step outgrew its list workspaces: its results are not valid; the graph "
                                   "has been recorded again with larger ones")
        self.recaptures += 1
        self._capture()  # (the counts just read now size the workspaces; the capture itself does not run the step)
        self.replay()
        if not self.fits():
            raise CapacityExceeded("the step outgrew its list workspaces twice in a row")
        return self.outputs


__all__ = ["GraphedStep", "Branches", "CapacityExceeded"]
