"""Builds eogs2_amd/libeogs_rast_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libeogs_rast_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-slp-vectorize",  # SLP packs independent DPP reduction chains into v_pk_* ops, which cannot carry DPP
         "-Wall", "-Werror=unused-value", "-Wno-unused-function", "-I", os.path.join(ROOT, "include")]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash():
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h): identifies the kernel set a profile was taken on."""
    import hashlib

    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, extra=(), verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC, *FLAGS, *extra, "-o", OUT, *sources()]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a != "--force"])
