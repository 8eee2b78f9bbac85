"""CPU: the packed lower layout of the Cholesky-path Schur assembly (csrc/lrn_common.h) checked on the host:
tests/host/packed_layout_check.cpp is compiled with hipcc (host code only) and run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_lower_layout_is_a_bijection(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "packed_layout_check")
    src = os.path.join(ROOT, "tests", "host", "packed_layout_check.cpp")
    subprocess.run([hipcc, "-std=c++17", "-O1", "-w", "-I", os.path.join(ROOT, "loraine.jl_amd", "csrc"), src, "-o", exe],
                   check=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad=0" in out.stdout
