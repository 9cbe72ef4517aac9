"""Generates tests/golden/tsdf_*.npz by running the REFERENCE's own `tsdf.py` (imported from /root/reference, which only
exists in the build container) on seeded inputs. Only inputs and outputs are stored.

    python tests/golden/make_golden_tsdf.py

`src/gaussiansplatting/tsdf.py` imports `iio`, `omegaconf`, `hydra`, `eval.eval_dsm` (and optionally `clearml`) at module level — the
DSM file I/O and the Hydra entry point, none of which the classes driven here touch; they are stubbed in `sys.modules` (the
same device make_golden_shade.py uses for `gaussian_renderer`). Both classes hard-wire `torch.device("cuda:0")`
(tsdf.py:197, :383): the module's `torch` global is replaced by a forwarding module whose `device()` answers the CPU, so the
constructors and every statement of `RangeImageEOGS.__init__ / reconstruct_normals / get_weights / sample_sdf`
(tsdf.py:186-368) and `TSDFVolume.__init__ / integrate / update_tsdf` (:374-520) run unmodified, in fp32, as the reference
runs them. Stored per case: the volume's constructor results (voxel counts, the three axes), per view the affine model, the
altitude image and the weights `get_weights()` returned, `sample_sdf` at the voxel centres, and both volumes after every view.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REFROOT = "/root/reference/src/gaussiansplatting"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_ref():
    for name in ("iio", "omegaconf", "eval", "eval.eval_dsm", "hydra"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["omegaconf"].OmegaConf = types.SimpleNamespace(register_new_resolver=lambda *a, **k: None)
    sys.modules["hydra"].main = lambda **k: (lambda f: f)  # tsdf.py:787 decorates its command-line entry point
    sys.modules["eval.eval_dsm"].main_hydra_dsm = None
    sys.modules["eval"].eval_dsm = sys.modules["eval.eval_dsm"]
    spec = importlib.util.spec_from_file_location("ref_tsdf", os.path.join(REFROOT, "tsdf.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class _TorchOnCPU(types.ModuleType):  # torch.device("cuda:0") -> cpu; everything else is torch
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*a, **k):
            return torch.device("cpu")

    mod.torch = _TorchOnCPU("torch")
    return mod


def view(H, W, seed, shear=0.15):
    """A near-nadir affine model (the reference's Nadir coefficients, to_affine.py:244-249, plus shear) over a smooth
    altitude field with a cliff (a ridge of back-facing normals -> weights clamped to 0 by get_weights)."""
    g = torch.Generator().manual_seed(seed)
    coef = torch.tensor([[0.0, 0.9, 0.0], [0.9, 0.0, 0.0], [0.0, 0.0, 1.0]])
    coef[:2, 2] = shear * torch.randn(2, generator=g)
    intercept = torch.tensor([0.02, -0.03, 0.1]) + 0.01 * torch.randn(3, generator=g)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    alt = 0.15 * torch.sin(3 * xx + 0.3 * seed) * torch.cos(2 * yy) + 0.02 * torch.rand((H, W), generator=g)
    alt[:, W // 2:] += 0.12  # a cliff
    return coef, intercept, alt


def make_case(ref, name, H, W, bounds, vox, fact, scale, seeds):
    vol = ref.TSDFVolume(np.array(bounds, dtype=np.float64), vox, fact)
    d = dict(vol_bounds=np.array(bounds, dtype=np.float64), vox_size=np.float64(vox), trunc_margin_fact=np.float64(fact),
             model_scale=np.float64(scale), num_voxels=np.array(vol.num_voxels_per_dimension, dtype=np.int64),
             trunc_margin=np.float64(vol._trunc_margin), n_views=np.int64(len(seeds)))
    for i in range(3):
        d[f"axis{i}"] = vol.axes[i].numpy().copy()
    for v, seed in enumerate(seeds):
        coef, intercept, alt = view(H, W, seed)
        meta = {"img": f"view{v}", "model": {"scale": scale, "coef_": coef.tolist(), "intercept_": intercept.tolist()}}
        ri = ref.RangeImageEOGS(meta, alt.numpy())
        if v == 0:  # an explicit zero-weight region as well (back-facing pixels): 0 / 0 = NaN in voxels nothing has written yet
            ri.pixels_angle[..., : H // 4, : W // 4] = -0.5
        wgt = ri.get_weights()
        sdf, mask, w_s = ri.sample_sdf(vol.world_coords)
        vol.integrate(ri)
        d.update({f"v{v}_coef": coef.numpy(), f"v{v}_intercept": intercept.numpy(), f"v{v}_altitude": ri.altitude_img.numpy().copy(),
                  f"v{v}_weights": wgt.numpy().copy(), f"v{v}_sdf": sdf.numpy().copy(), f"v{v}_mask": mask.numpy().copy(),
                  f"v{v}_sampled_weights": w_s.numpy().copy(), f"v{v}_tsdf_vol": vol._tsdf_vol.numpy().copy(),
                  f"v{v}_weight_vol": vol._weight_vol.numpy().copy()})
    np.savez_compressed(os.path.join(OUT, f"tsdf_{name}.npz"), **d)
    t = vol._tsdf_vol
    print(f"tsdf_{name}: volume {tuple(t.shape)}, NaN voxels {int(torch.isnan(t).sum())}, touched {int((vol._weight_vol > 0).sum())}")


def main():
    ref = load_ref()
    make_case(ref, "three_views_48x64", 48, 64, [[-1.4, 1.4], [-1.3, 1.3], [-0.3, 0.4]], 0.08, 3.0, 1.7, (1, 2, 3))
    make_case(ref, "ragged_33x17", 33, 17, [[-0.5, 0.5], [-0.5, 0.5], [-0.2, 0.3]], 0.05, 2.0, 1.0, (4, 5, 6))
    make_case(ref, "outside_view_32x32", 32, 32, [[-4.0, -3.5], [-0.2, 0.2], [0.0, 0.1]], 0.05, 3.0, 1.7, (7,))


if __name__ == "__main__":
    main()
