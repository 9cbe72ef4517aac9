"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see rast_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (eogs2_amd/, diff_gaussian_rasterization/) never does.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librast_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "rast_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "eogs_rast.h")
    stale = (not os.path.exists(LIB_PATH)) or (not os.path.exists(os.path.join(_HERE, "librast_oracle_fma.so"))) or (
        not os.path.exists(os.path.join(_HERE, "librast_oracle_f64.so"))) or any(
        os.path.getmtime(p) > os.path.getmtime(LIB_PATH) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return LIB_PATH


_abi = None
_abi_fma = None
_abi_f64 = None


def abi_f64():
    """RastABI over the arbiter build (rast_oracle.c, -DORACLE_F64): the fp32 forward and its decisions, the backward's
    differentiable quantities recomputed and chained in double. tests/parity_cases.py: who is right on an ill-conditioned gradient."""
    global _abi_f64
    if _abi_f64 is None:
        from eogs2_amd._abi import RastABI

        build()
        _abi_f64 = RastABI(os.path.join(_HERE, "librast_oracle_f64.so"))
        assert _abi_f64.cdll.eogs_oracle_is_f64() == 1
    return _abi_f64


def abi_fma():
    """RastABI over the FMA-contracted build of the same restatement (None if the host CPU has no FMA or the build is
    missing): a second valid fp32 rounding, used by the tests to measure rounding-level instability."""
    global _abi_fma
    if _abi_fma is None:
        path = os.path.join(_HERE, "librast_oracle_fma.so")
        build()
        try:
            has_fma = "fma" in open("/proc/cpuinfo").read()
        except OSError:
            has_fma = False
        if not has_fma or not os.path.exists(path):
            return None
        from eogs2_amd._abi import RastABI

        _abi_fma = RastABI(path)
    return _abi_fma


def abi():
    """RastABI over librast_oracle.so (host pointers)."""
    global _abi
    if _abi is None:
        from eogs2_amd._abi import RastABI

        _abi = RastABI(build())
    return _abi
