"""Shared helpers of the shade tests: fixture loading and one driver that runs either implementation."""
import glob
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixtures(prefix):
    return sorted(glob.glob(os.path.join(GOLD, f"shade_{prefix}*.npz")))


def load(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def matrix_of(fx, dtype, dev):
    kind = str(fx["kind"])
    if kind == "cc":
        M = np.concatenate([fx["weight"].reshape(3, 3), fx["bias"].reshape(3, 1)], axis=1)
    elif kind == "exposure":
        M = fx["exposure"][0]
    else:
        M = np.eye(3, 4, dtype=np.float32)
    return torch.tensor(M, dtype=dtype, device=dev)


def expected_matrix_grad(fx):
    kind = str(fx["kind"])
    if kind == "cc":
        return np.concatenate([fx["g_weight"].reshape(3, 3), fx["g_bias"].reshape(3, 1)], axis=1)
    if kind == "exposure":
        return fx["g_exposure"][0]
    return None


def run_shade(fn, fx, dtype, dev):
    """fn(raw, alt_diff|None, M, inshadow) -> (cc, shaded, shadow); returns outputs and gradients as float64 numpy."""
    t = lambda a: torch.tensor(a, dtype=dtype, device=dev)
    raw = t(fx["raw"]).requires_grad_(True)
    M = matrix_of(fx, dtype, dev).requires_grad_(True)
    has = "alt_diff" in fx
    alt = t(fx["alt_diff"]).requires_grad_(True) if has else None
    ins = t(fx["inshadow"]).requires_grad_(True)
    cc, shaded, shadow = fn(raw, alt, M, ins if has else None)
    loss = (shaded * t(fx["g_shaded"])).sum() + (cc * t(fx["g_cc"])).sum()
    if has:
        loss = loss + (shadow * t(fx["g_shadow"])).sum()
    loss.backward()
    n = lambda x: None if x is None else x.detach().double().cpu().numpy()
    out = dict(cc=n(cc), shaded=n(shaded), g_raw=n(raw.grad), g_M=n(M.grad))
    if has:
        out.update(shadow=n(shadow), g_alt_diff=n(alt.grad), g_inshadow=n(ins.grad))
    return out


def close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= tol, f"{what}: max |diff| / max |ref| = {err:.3e} > {tol}"
