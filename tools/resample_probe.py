"""Tuning aid (GPU box): one configuration of the resample forward + backward in a loop, for `rocprofv3 --kernel-trace --stats`.
usage: python3 tools/resample_probe.py <scale 1|2> <C> <n_out> [size] [rand|smooth]   (prints ms per fwd+bwd; kernel split from the profiler)"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eogs2_amd.resample import resample
f, C, n_out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
dev = torch.device("cuda:0")
H = W = S
U, V = torch.meshgrid(torch.linspace(-1, 1, W, device=dev), torch.linspace(-1, 1, H, device=dev), indexing="xy")
M = torch.eye(3, device=dev); M[:2, 2] = torch.tensor([0.01, -0.02], device=dev)
vr = torch.rand(C, H * f, W * f, device=dev, requires_grad=True)
mode = sys.argv[5] if len(sys.argv) > 5 else "rand"
if mode == "smooth":  # a terrain-like altitude: low-frequency field (what a rendered altitude map looks like), same range
    alt = torch.nn.functional.interpolate(torch.rand(1, 1, 16, 16, device=dev), size=(H, W), mode="bicubic", align_corners=True)[0, 0].clamp(0, 1).contiguous().requires_grad_(True)
else:         # white noise: neighbouring pixels land up to ~20 cells apart (worst case for the backward's candidate boxes)
    alt = torch.rand(H, W, device=dev, requires_grad=True)
w = torch.randn(n_out, H, W, device=dev)
uva = torch.stack((U, V, alt), dim=-1)
def run():
    vr.grad = alt.grad = None
    s, uv = resample(vr, M, torch.stack((U, V, alt), dim=-1), n_out=n_out, fill_channel=n_out - 1)
    torch.autograd.backward([s, uv], [w, torch.ones_like(uv)])
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): run()
torch.cuda.synchronize()
print(f"scale {f} C {C} n_out {n_out} size {S} altitude {mode}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per fwd+bwd (incl. torch.stack and autograd)")
