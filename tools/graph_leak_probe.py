"""Round 4: where does device memory go when a step is recorded again and again? `examples/train_synthetic.py --graph
--parallel-renders --prune-every 1` (a new recording nearly every iteration) ran the 288 GB card out of memory after ~875 recordings
while PyTorch's allocator reported 0.8 GB in use. Prints the device's free memory (hipMemGetInfo) every few recordings of
  torch-serial   : a bare torch.cuda.graph that allocates 3 x 64 MB inside the capture (no code of this repo)
  torch-branches : the same with the three allocations on three forked / joined side streams
  step-serial    : GraphedStep over three rasterizer renders + backward, one stream
  step-branches  : the same with eogs2_amd.graph.Branches

    python tools/graph_leak_probe.py [rounds=40] [legs=torch-serial,torch-branches,step-serial,step-branches]
"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
LEGS = (sys.argv[2] if len(sys.argv) > 2 else "torch-serial,torch-branches,step-serial,step-branches").split(",")


def free_mib():
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0] / 2**20


def report(tag, i, base):
    f = free_mib()
    print(f"  {tag:15s} after {i:4d} recordings: device free {f:10.1f} MiB ({base - f:+9.1f} since the leg's start), torch allocated "
          f"{torch.cuda.memory_allocated() / 2**20:7.1f} reserved {torch.cuda.memory_reserved() / 2**20:7.1f}", flush=True)


def torch_leg(branches):
    x = torch.ones(16 << 20, device=dev)
    streams = [torch.cuda.Stream() for _ in range(3)]
    base = free_mib()
    for i in range(1, ROUNDS + 1):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cur = torch.cuda.current_stream()
            outs = []
            for s in streams:
                if branches:
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        outs.append(x * 2.0)
                else:
                    outs.append(x * 2.0)
            if branches:
                for s in streams:
                    cur.wait_stream(s)
            y = outs[0] + outs[1] + outs[2]
        g.replay()
        torch.cuda.synchronize()
        assert float(y[0]) == 6.0
        del g, y, outs
        if i % 10 == 0 or i == ROUNDS:
            report("torch-branches" if branches else "torch-serial", i, base)


def step_leg(branches):
    from eogs2_amd.fused import rasterize_raw
    from eogs2_amd.graph import Branches, GraphedStep
    from eogs2_amd.rasterizer import GaussianRasterizationSettings
    from eogs2_amd.synthetic import make_scene

    P, H, W = 200_000, 512, 512
    sc = make_scene(P, H, W, seed=0, opacity="trained", device=dev)
    xyz = sc["means3D"].clone().requires_grad_(True)
    f_dc = ((sc["colors"][:, :3] - 0.5) / 0.28209479177387814).contiguous().requires_grad_(True)
    logit = torch.logit(sc["opacities"].squeeze(1).clamp(1e-4, 1 - 1e-4)).requires_grad_(True)
    lsc = torch.log(sc["scales"]).requires_grad_(True)
    rot = sc["rotations"].clone().requires_grad_(True)
    vm = sc["viewmatrix"]
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=1.0, tanfovy=1.0, bg=sc["bg"], scale_modifier=1.0,
                                       viewmatrix=vm, projmatrix=vm, sh_degree=0, campos=torch.zeros(3, device=dev),
                                       prefiltered=False, debug=False, antialiasing=False)
    alt = vm[:, 2].detach().contiguous()
    w = torch.randn(5, H, W, device=dev)
    br = Branches(3, device=dev) if branches else None

    def one():
        m2 = torch.zeros_like(xyz, requires_grad=True)
        c, _, _ = rasterize_raw(xyz, m2, f_dc, logit, lsc, rot, alt, rs)
        return (c * w).sum()

    def fn():
        for p in (xyz, f_dc, logit, lsc, rot):
            p.grad = None
        losses = br.run([one, one, one], shared=()) if br is not None else [one(), one(), one()]
        loss = losses[0] + losses[1] + losses[2]
        loss.backward()
        return loss.detach()

    base = free_mib()
    for i in range(1, ROUNDS + 1):
        step = GraphedStep(fn, warmup=1)
        v = float(step())
        assert v == v
        del step
        if i % 10 == 0 or i == ROUNDS:
            report("step-branches" if branches else "step-serial", i, base)


for leg in LEGS:
    {"torch-serial": lambda: torch_leg(False), "torch-branches": lambda: torch_leg(True),
     "step-serial": lambda: step_leg(False), "step-branches": lambda: step_leg(True)}[leg]()
