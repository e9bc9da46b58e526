"""The C-ABI consumed from plain C (gcc, no Python, no torch types): host-only handle on CPU, full
evaluation on the GPU."""
import os
import subprocess

import pytest

from conftest import ROOT


def build(tmp_path):
    from gelato_amd import _lib
    assert os.path.exists(_lib.SO_PATH)
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lgelato_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return exe


def test_c_consumer_host_only(tmp_path):
    out = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi_smoke host OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_consumer_gpu(tmp_path):
    out = subprocess.run([build(tmp_path), "gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi_smoke gpu OK" in out.stdout, out.stdout + out.stderr
