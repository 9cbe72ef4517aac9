"""GPU (MI355X): the C-ABI driven by a C++ host with no torch in the process (tests/capi_graph_host.cpp) — an eager step the
way the reference's forward works, then the same step recorded with hipStreamBeginCapture and replayed: bit-identical
results, the counts mirrored to the host while the replay runs, an outgrown replay reported and harmless. This is the
binding a maintainer with a C++ trainer would write; the Python wrapper does the same through ctypes."""
import os
import shutil
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_eager_and_recorded_step(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib_dir = os.path.join(ROOT, "eogs2_amd")
    assert os.path.exists(os.path.join(lib_dir, "libeogs_rast_hip.so")), "build the HIP library first (python -m eogs2_amd.build)"
    exe = str(tmp_path / "capi_graph_host")
    build = subprocess.run([hipcc, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "capi_graph_host.cpp"), "-o", exe, "-L", lib_dir, "-leogs_rast_hip",
                            "-Wl,-rpath," + lib_dir], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ALL CHECKS PASSED" in run.stdout, run.stdout[-3000:] + run.stderr[-2000:]
    assert "FAILED" not in run.stdout
    assert run.stdout.count("ok: ") >= 12
