"""End-to-end use of the MI355X path on synthetic data: the structure of one EOGS++ training iteration
(src/gaussiansplatting/train_pan.py:262-400,663-690) with every heavy step on the HIP library.

    python examples/train_synthetic.py [--gaussians 200000] [--size 512] [--iters 200]

Per iteration: render the view through `eogs2_amd.render.render` (raw-parameter front end, §8 f1), render a sun-like
virtual camera at twice the resolution and resample it onto the view (`eogs2_amd.resample`, §8 f2), run the camera's
render pipeline — learnable colour correction, shadow map from the altitude difference, in-shadow tint
(`eogs2_amd.shade.render_pipeline`, §8 f2) —, photometric loss against a target image
(`eogs2_amd.losses.photometric_loss`) plus the sun-camera consistency pair and the translucent-shadow regulariser
(`eogs2_amd.shade.suncamera_l`, `translucentshadows_l`), `FusedAdam` step on the Gaussians and Adam on the camera
parameters (§8 f3), transparent-Gaussian prune by stream compaction (`prune_optimizer`, §8 f3). The target is the shaded
render of the unperturbed scene under an identity colour correction, so the loss must fall. Initial scales come from
`simple_knn._C.distCUDA2` (§8 f4).
"""
import argparse
import math
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from eogs2_amd.losses import photometric_loss  # noqa: E402
from eogs2_amd.optim import FusedAdam, alive_rows, prune_optimizer, retire_rows  # noqa: E402
from eogs2_amd.render import render  # noqa: E402
from eogs2_amd.graph import Branches  # noqa: E402
from eogs2_amd.resample import render_resample_virtual_camera, resample  # noqa: E402
from eogs2_amd.shade import randomcam_l, render_pipeline, suncamera_l, translucentshadows_l  # noqa: E402
from eogs2_amd.synthetic import make_camera, make_scene  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

C0 = 0.28209479177387814


class Camera:
    """The attributes gaussian_renderer/renderer.py reads from an AffineCamera."""

    def __init__(self, vm, H, W):
        self.FoVx = self.FoVy = 1.0
        self.affine = self.world_view_transform = self.full_proj_transform = vm
        self.learn_wv_only_lastparam = False
        self.image_height, self.image_width = H, W
        self.camera_center = torch.zeros(3, device=vm.device)


class Gaussians:
    """The attributes renderer.py / gaussian_model.py use: raw parameters, one optimizer group each."""

    active_sh_degree = 0

    def __init__(self, xyz, rgb, opacity, scales, rotations):
        P = xyz.shape[0]
        self._xyz = torch.nn.Parameter(xyz.clone())
        self._features_dc = torch.nn.Parameter(((rgb - 0.5) / C0).reshape(P, 1, 3).contiguous())
        self._features_rest = torch.nn.Parameter(torch.zeros(P, 0, 3, device=xyz.device))
        self._opacity = torch.nn.Parameter(torch.log(opacity / (1 - opacity)))
        self._scaling = torch.nn.Parameter(torch.log(scales))
        self._rotation = torch.nn.Parameter(rotations.clone())
        lrs = dict(xyz=2e-5, f_dc=2.5e-3, f_rest=1.25e-4, opacity=2.5e-2, scaling=5e-3, rotation=1e-3)
        groups = [dict(params=[getattr(self, "_" + n)], lr=lr, name=k)
                  for k, n, lr in (("xyz", "xyz", lrs["xyz"]), ("f_dc", "features_dc", lrs["f_dc"]),
                                   ("f_rest", "features_rest", lrs["f_rest"]), ("opacity", "opacity", lrs["opacity"]),
                                   ("scaling", "scaling", lrs["scaling"]), ("rotation", "rotation", lrs["rotation"]))]
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)  # gaussian_model.py:262 with the fused step
        self.max_radii2D = torch.zeros(P, device=xyz.device)

    get_xyz = property(lambda s: s._xyz)

    def prune(self, keep):  # gaussian_model.py:488-505
        t, (self.max_radii2D,) = prune_optimizer(self.optimizer, keep, extra=(self.max_radii2D,))
        self._xyz, self._features_dc, self._features_rest = t["xyz"], t["f_dc"], t["f_rest"]
        self._opacity, self._scaling, self._rotation = t["opacity"], t["scaling"], t["rotation"]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=200_000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="record renders + losses + backward once into a HIP graph and replay it (eogs2_amd.graph.GraphedStep); "
                         "the optimizers stay outside, the graph is recorded again after every prune")
    ap.add_argument("--sun-altitude-only", action="store_true",
                    help="the sun camera renders and resamples its altitude channel alone — what the reference's shipped "
                         "configuration consumes of it (iterstart_L_sun_resample is never reached, gs_config/train.yaml:123); "
                         "the sun-camera RGB consistency term is then absent, as it is there")
    ap.add_argument("--random-camera", action="store_true",
                    help="third render of the iteration: a random virtual camera at the view's size, resampled onto the view, "
                         "with the masked altitude / RGB consistency pair (train_pan.py:375-391, loss/main_loss.py:151-164)")
    ap.add_argument("--no-prune", action="store_true", help="keep every Gaussian (timing runs)")
    ap.add_argument("--prune-every", type=int, default=50, metavar="N",
                    help="iterations between prune points (the reference: 1, train_pan.py:673-678 — it asks the device "
                         "`transparent_mask.any()` every iteration)")
    ap.add_argument("--defer-prune", type=int, default=0, metavar="K",
                    help="at the prune points retire the transparent Gaussians (opacity 0: eogs2_amd.optim.retire_rows) and "
                         "compact only at every K-th of them and at the end: same renders, same updates, but no shape changes, "
                         "no wait for the device, and a recorded graph (--graph) keeps replaying in between")
    ap.add_argument("--parallel-renders", action="store_true",
                    help="the renders of the iteration (independent given the parameters: only the RESAMPLES need the view's "
                         "altitude) are queued on streams of their own, the largest first (eogs2_amd.graph.Branches); with --graph "
                         "they become parallel branches of the recorded graph, and autograd runs their backward passes on the same streams")
    ap.add_argument("--require-radii", action="store_true",
                    help="pipe.require_radii: every render also returns radii and `visibility_filter` = nonzero(radii > 0), which waits "
                         "for the device (renderer.py:128-130). The reference's shipped configuration has it OFF "
                         "(gs_config/train.yaml:39: require_radii = not only_prune, only_prune: True): off by default here too")
    a = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    P, H, W = a.gaussians, a.size, a.size
    sc = make_scene(P, H, W, seed=0, opacity="trained", device=dev)
    cam = Camera(sc["viewmatrix"], H, W)
    sun = Camera(make_camera(2 * H, 2 * W, seed=5, device=dev), 2 * H, 2 * W)
    cam2sun = torch.eye(3, device=dev)
    cam2sun[:2, 2] = (sun.affine[2, :2] - cam.affine[2, :2]) / 350.0  # altitude-dependent shift between the two views
    rnd = Camera(make_camera(H, W, seed=9, device=dev), H, W)
    cam2rnd = torch.eye(3, device=dev)
    cam2rnd[:2, 2] = (rnd.affine[2, :2] - cam.affine[2, :2]) / 350.0
    pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, require_radii=a.require_radii)
    bg = sc["bg"]
    U, V = torch.meshgrid(torch.linspace(-1, 1, W, device=dev), torch.linspace(-1, 1, H, device=dev), indexing="xy")

    branches = Branches(3, device=dev) if a.parallel_renders else None

    def view(m, cc_cam):
        """train_pan.py:279-330: view render, sun render resampled onto the view, camera render pipeline."""
        if branches is not None:
            # the three renders side by side, the 2H x 2W one first; then what render_resample_virtual_camera does with them
            sun_img, out, rnd_img = branches.run([
                lambda: render(sun, m, pipe, bg, altitude_only=a.sun_altitude_only)["render"],
                lambda: render(cam, m, pipe, bg),
                lambda: render(rnd, m, pipe, bg)["render"] if a.random_camera else None],
                shared=())  # forwards only: the iteration's one backward() follows the join
            img, altitude = out["render"][:3], out["render"][3]
            uva = torch.stack((U, V, altitude / 350.0), dim=-1)
            if a.sun_altitude_only:
                smp, sun_uv = resample(sun_img, cam2sun, uva, n_out=1, fill_channel=0)
                sun_rgb, sun_alt = None, smp[0]
            else:
                smp, sun_uv = resample(sun_img, cam2sun, uva)
                sun_rgb, sun_alt = smp[:3], smp[3]
            sun_altitude_diff = altitude - sun_alt
            shaded = render_pipeline(cc_cam, img, sun_altitude_diff)
            new = None
            if a.random_camera:
                smp, new_uv = resample(rnd_img, cam2rnd, uva)
                new = (altitude - smp[3], smp[:3], new_uv)
            return out, img, sun_rgb, sun_uv, sun_altitude_diff, shaded, new
        out = render(cam, m, pipe, bg)
        img, altitude = out["render"][:3], out["render"][3]
        uva = torch.stack((U, V, altitude / 350.0), dim=-1)
        sun_rgb, sun_alt, sun_uv = render_resample_virtual_camera(sun, cam2sun, uva, m, pipe, bg,
                                                                  altitude_only=a.sun_altitude_only)
        sun_altitude_diff = altitude - sun_alt
        shaded = render_pipeline(cc_cam, img, sun_altitude_diff)
        new = None
        if a.random_camera:  # train_pan.py:375-391 / loss/main_loss.py:123-164
            new_rgb, new_alt, new_uv = render_resample_virtual_camera(rnd, cam2rnd, uva, m, pipe, bg)
            new = (altitude - new_alt, new_rgb, new_uv)
        return out, img, sun_rgb, sun_uv, sun_altitude_diff, shaded, new

    def colour_camera(perturb):
        c = types.SimpleNamespace(use_cc=True, use_exposure=False, use_shadow=True)
        c.color_correction = torch.nn.Conv2d(3, 3, 1, bias=True).to(dev)  # affine_cameras.py:219-231
        with torch.no_grad():
            c.color_correction.weight.copy_((torch.eye(3) + perturb * torch.randn(3, 3, generator=gcam)).reshape(3, 3, 1, 1))
            c.color_correction.bias.zero_()
        c.inshadow_color_correction = torch.nn.Parameter(torch.full((3, 1, 1), 0.05, device=dev))
        return c

    gcam = torch.Generator().manual_seed(2)
    target_model = Gaussians(sc["means3D"], sc["colors"][:, :3], sc["opacities"].squeeze(1).clamp(1e-4, 1 - 1e-4),
                             sc["scales"], sc["rotations"])
    with torch.no_grad():
        gt = view(target_model, colour_camera(0.0))[5]["final"].clone()
    cc_cam = colour_camera(0.15)
    camera_optimizer = torch.optim.Adam([*cc_cam.color_correction.parameters(), cc_cam.inshadow_color_correction], lr=2e-3)

    # the trainee: perturbed colours / opacities / positions, scales re-initialised from the 3-NN statistic
    g = torch.Generator().manual_seed(1)
    noise = lambda *s: torch.randn(*s, generator=g).to(dev)
    dist2 = torch.clamp_min(distCUDA2(sc["means3D"]), 1e-7)  # gaussian_model.py:179-182
    model = Gaussians(sc["means3D"] + 2e-4 * noise(P, 3), (sc["colors"][:, :3] + 0.2 * noise(P, 3)).clamp(0.02, 0.98),
                      torch.full((P,), 0.3, device=dev), torch.sqrt(dist2)[:, None].repeat(1, 3), sc["rotations"])
    def fwd_bwd():
        """Everything between two optimizer steps; reads the model's and the camera's parameter tensors in place."""
        model.optimizer.zero_grad(set_to_none=True)
        camera_optimizer.zero_grad(set_to_none=True)
        out, img, sun_rgb, sun_uv, sun_altitude_diff, shaded, new = view(model, cc_cam)
        loss, _ = photometric_loss(shaded["final"], gt, 0.2)
        loss = loss + 1e-3 * translucentshadows_l(shaded["shadowmap"])
        if sun_rgb is not None:
            L_sun_alt, L_sun_rgb = suncamera_l(img, sun_rgb, sun_altitude_diff, sun_uv)
            loss = loss + 1e-4 * L_sun_alt + 1e-3 * L_sun_rgb
        if new is not None:
            L_new_alt, L_new_rgb = randomcam_l(new[0], img, new[1], new[2])
            loss = loss + 1e-4 * L_new_alt + 1e-3 * L_new_rgb
        loss.backward()
        return loss.detach(), out.get("radii")

    first = last = None
    step, stale = None, False  # the recorded graph of fwd_bwd; stale: recorded for parameter tensors that a prune replaced
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_steady = None  # (the clock of the last `timed` iterations: without the first ones, which allocate and record)
    timed = max(1, a.iters // 2)
    for it in range(1, a.iters + 1):
        if it == a.iters - timed + 1:
            torch.cuda.synchronize()
            t_steady = time.perf_counter()
        if a.graph:
            if step is None:
                from eogs2_amd.graph import GraphedStep

                step = GraphedStep(fwd_bwd, warmup=1)  # (the eager warm-up run changes nothing: no optimizer step inside)
            elif stale:
                step.record_again()  # new parameter tensors, new shapes (fwd_bwd reads them from `model`); same memory pool
            stale = False
            loss, radii = step()
        else:
            loss, radii = fwd_bwd()
        model.optimizer.step()
        camera_optimizer.step()
        with torch.no_grad():
            if radii is not None:  # train_pan.py:681-686 (densification statistics: only with require_radii)
                model.max_radii2D = torch.maximum(model.max_radii2D, radii.float())
            if it % a.prune_every == 0 and not a.no_prune:  # train_pan.py:673-678
                keep = model._opacity.squeeze() >= math.log(0.005 / 0.995)
                last_point = it + a.prune_every > a.iters
                if a.defer_prune and not last_point and (it // a.prune_every) % a.defer_prune:
                    retire_rows(model.optimizer, keep)
                elif a.defer_prune:
                    alive = alive_rows(model.optimizer) & keep
                    if not bool(alive.all()):
                        model.prune(alive)
                        stale = True
                elif not bool(keep.all()):
                    model.prune(keep)
                    stale = True  # new parameter tensors, new shapes: record again
        if it == 1 or it % 25 == 0 or it == a.iters:
            v = float(loss)
            first = v if first is None else first
            last = v
            if not a.quiet:
                print(f"iter {it:4d}  loss {v:.5f}  gaussians {model._xyz.shape[0]}  device memory in use "
                      f"{(lambda f, t: (t - f) / 2**20)(*torch.cuda.mem_get_info()):.0f} MiB")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dt = t1 - t0
    main.last_ms_per_iter = (t1 - t_steady) / timed * 1e3  # steady state: the last half of the run
    if not a.quiet:
        print(f"{a.iters} iterations in {dt:.2f} s ({dt / a.iters * 1e3:.2f} ms/iter over all, {main.last_ms_per_iter:.2f} ms/iter over the "
              f"last {timed}; {3 if a.random_camera else 2} renders + resample + render pipeline + losses + Adam each)")
    return first, last, model._xyz.shape[0]


if __name__ == "__main__":
    main()
