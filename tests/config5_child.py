"""BASELINE.json's configuration 5 in miniature (`reproduce_main.sh`: training with opacity reset + prune, TSDF
post-processing, view-sharded data parallelism), as one script that the tests run as 1 rank and as 2 ranks.

Per iteration (src/gaussiansplatting/train_pan.py:252-330,663-732 with every heavy step on the HIP library): this rank's
view of the step is rendered through `eogs2_amd.render.render` (raw-parameter front end), photometric loss
`(1-l) L1 + l (1-SSIM)` against the view's target, backward; with several ranks the raw-parameter gradients are summed by
`GradBucket.all_reduce()` (the synchronous path: what a step with other gradient sources must use); `FusedAdam` step;
transparent-Gaussian prune EVERY iteration (`opt.only_prune`, train_pan.py:673-678: logit < min_opacity = -6) by stream
compaction; opacity reset every `--reset` iterations (:726-732); densification statistics kept replica-identical. After
training, the altitude renders of four views are integrated into a TSDF volume (`src/gaussiansplatting/tsdf.py:459-498`).

Prints one JSON line: losses, Gaussian counts, a digest of every prune decision and of the final parameters (ranks of a
data-parallel run must print the same digests), TSDF statistics.
usage: python tests/config5_child.py --gaussians 300000 --size 800 --iters 300 [--backend gloo]  (RANK / WORLD_SIZE from env)
"""
import argparse
import hashlib
import json
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--reset", type=int, default=100, help="opacity_reset_interval (3000 in gs_config/train.yaml:104)")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--backend", default="gloo")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda:0")  # (a rehearsal: the ranks share the card and exchange over gloo)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, rank=rank, world_size=world)

    from train_synthetic import Camera, Gaussians
    from eogs2_amd.losses import photometric_loss
    from eogs2_amd.optim import reset_opacity
    from eogs2_amd.parallel import GradBucket, all_reduce_densification_stats
    from eogs2_amd.render import render
    from eogs2_amd.synthetic import ALT_SCALE, make_camera, make_scene
    from eogs2_amd.tsdf import TSDFVolume

    P, H, W = a.gaussians, a.size, a.size
    sc = make_scene(P, H, W, seed=0, opacity="trained", device=dev)
    cams = [Camera(make_camera(H, W, seed=20 + v, device=dev), H, W) for v in range(a.views)]
    pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, require_radii=True)
    bg = sc["bg"]
    target = Gaussians(sc["means3D"], sc["colors"][:, :3], sc["opacities"].squeeze(1).clamp(1e-4, 1 - 1e-4), sc["scales"],
                       sc["rotations"])
    with torch.no_grad():
        gts = [render(c, target, pipe, bg)["render"][:3].clone() for c in cams]
    # the trainee (same on every rank): perturbed colours, flat opacity, and a tenth of the Gaussians just above the prune
    # threshold so that the optimisation pushes some of them below it
    g = torch.Generator().manual_seed(1)
    noise = lambda *s: torch.randn(*s, generator=g).to(dev)
    op0 = torch.full((P,), 0.3, device=dev)
    weak = torch.rand(P, generator=g).to(dev) < 0.1
    op0[weak] = 1.0 / (1.0 + math.exp(5.8))
    model = Gaussians(sc["means3D"] + 2e-4 * noise(P, 3), (sc["colors"][:, :3] + 0.2 * noise(P, 3)).clamp(0.02, 0.98), op0,
                      sc["scales"], sc["rotations"])
    names = ("xyz", "f_dc", "opacity", "scaling", "rotation")

    def params():
        return [model._xyz, model._features_dc, model._opacity, model._scaling, model._rotation]

    bucket = GradBucket(params()) if world > 1 else None
    digest = hashlib.sha256()
    losses, counts, prunes, resets = [], [], 0, 0
    for it in range(1, a.iters + 1):
        v = ((it - 1) * world + rank) % a.views  # view-sharded: rank r takes view r of the step's batch
        out = render(cams[v], model, pipe, bg)
        loss, _ = photometric_loss(out["render"][:3], gts[v], 0.2)
        loss.backward()
        if bucket is not None:
            bucket.all_reduce(average=True)
        model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True)
        with torch.no_grad():
            lv = loss.detach().clone()
            if dist is not None:
                dist.all_reduce(lv)
                lv /= world
            losses.append(float(lv))
            radii = out["radii"].float()
            acc, den = torch.zeros(model._xyz.shape[0], 1, device=dev), torch.ones(model._xyz.shape[0], 1, device=dev)
            all_reduce_densification_stats(acc, den, radii)  # max over the ranks' views (train_pan.py:679-690)
            model.max_radii2D = torch.maximum(model.max_radii2D, radii)
            transparent = model._opacity.squeeze() < -6.0  # opt.min_opacity (gs_config/train.yaml:113)
            if bool(transparent.any()):
                keep = ~transparent
                digest.update(np.packbits(keep.cpu().numpy()).tobytes())
                model.prune(keep)
                prunes += 1
                if bucket is not None:  # the parameters are new tensors: the exchange buffer follows them
                    bucket.close()
                    bucket = GradBucket(params())
            if it % a.reset == 0:  # train_pan.py:726-732
                model._opacity = reset_opacity(model.optimizer)["opacity"]
                resets += 1
                if bucket is not None:
                    bucket.close()
                    bucket = GradBucket(params())
            counts.append(int(model._xyz.shape[0]))
    state = hashlib.sha256()
    finite = True
    for p in params():
        finite = finite and bool(torch.isfinite(p).all())
        state.update(p.detach().cpu().numpy().tobytes())
    for st in model.optimizer.state.values():
        finite = finite and all(bool(torch.isfinite(t).all()) for t in st.values() if torch.is_tensor(t))

    # TSDF post-processing: every view's altitude render into one volume (tsdf.py:459-498; weights: the accumulated opacity)
    vol = TSDFVolume(np.array([[-1.0, 1.0], [-1.0, 1.0], [-0.1, 0.2]]), 2.0 / 127, 4.0, device=dev)
    with torch.no_grad():
        for c in cams:
            r = render(c, model, pipe, torch.zeros_like(bg))["render"]
            A = c.affine[:3, :3].t().contiguous().clone()
            b = c.affine[3, :3].clone()
            A[2], b[2] = A[2] / ALT_SCALE, b[2] / ALT_SCALE  # altitude in scene units
            ri = types.SimpleNamespace(affine_model=(A, b), model_scale=1.0,
                                       altitude_img=(r[3] / r[4].clamp_min(1e-6) / ALT_SCALE)[None, None],
                                       get_weights=lambda r=r: r[4].clamp(1e-3, 1)[None, None])  # (a zero weight on an untouched voxel is 0/0 = NaN in the reference's update too)
            vol.integrate(ri)
    touched = vol._weight_vol > 0
    line = {"rank": rank, "world": world, "iters": a.iters, "loss_first": float(np.mean(losses[:20])),
            "loss_last": float(np.mean(losses[-20:])), "gaussians_start": P, "gaussians_end": counts[-1], "prunes": prunes,
            "resets": resets, "finite": finite, "prune_digest": digest.hexdigest(), "state_digest": state.hexdigest(),
            "tsdf_touched_frac": float(touched.float().mean()),
            "tsdf_finite": bool(torch.isfinite(vol._tsdf_vol[touched]).all()),
            "tsdf_surface_frac": float(((vol._tsdf_vol < 1.0) & touched).float().mean())}
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
