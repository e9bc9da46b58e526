"""The engine's host code (LGR generator, pattern walker, index maps, descriptor validation, error paths) under
AddressSanitizer + UBSan + LeakSanitizer, through a host-only handle.  CPU build only: GPU ASan / xnack+ code
objects are not available on the pool, so the device code of this build is the ordinary gfx950 object and is
never launched here."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "gelato_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason="ROCm toolchain not present")
def test_host_code_under_asan_ubsan(tmp_path):
    lib = tmp_path / "libgelato_amd_asan.so"
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on",
             "-mllvm", "-disable-machine-licm", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-w"]
    subprocess.run([HIPCC, *flags, "-shared", "-o", str(lib), "gel_kernels.hip", "gel_host.hip"], cwd=CSRC,
                   check=True, capture_output=True, timeout=280)
    exe = tmp_path / "host_sanitize"
    subprocess.run([CLANG, "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", str(exe),
                    os.path.join(ROOT, "tests", "host_sanitize.c"), str(lib), "-lm",
                    "-Wl,-rpath,%s" % tmp_path, "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, timeout=120)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "HOST_SANITIZE_OK" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_oracle_under_asan_ubsan(tmp_path):
    """The CPU oracle itself (the checker every parity test leans on) with exactly sized heap buffers."""
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not present")
    exe = tmp_path / "oracle_sanitize"
    subprocess.run([gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fopenmp", "-o", str(exe),
                    os.path.join(ROOT, "tests", "oracle_sanitize.c"), os.path.join(ROOT, "oracle", "gelato_oracle.c"), "-lm"],
                   check=True, capture_output=True, timeout=240)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "ORACLE_SANITIZE_OK" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
