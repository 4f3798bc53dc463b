"""Host-side logic of the library that needs no GPU, compiled from the library's own header (jpeg-encoder_amd/csrc/host_internal.h):
the stripe tuner of large frames between page-locked buffers and the persistent worker threads of the batch calls."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stripe_tuner_and_worker_threads(tmp_path):
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("no hipcc on this host: host_internal.h includes the HIP runtime's headers")
    exe = tmp_path / "host_units"
    csrc = os.path.join(ROOT, "jpeg-encoder_amd", "csrc")
    # (compiled the way build.sh compiles the library's own host files: hipcc, -x hip, gfx950)
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-Wno-unused-const-variable",
                    "-x", "hip", "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "host_units.cpp"), "-o", str(exe), "-lpthread"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host units ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
