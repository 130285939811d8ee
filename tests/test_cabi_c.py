"""The C-ABI consumed from plain C (tests/cabi/cabi_rc.c, compiled with gcc against include/spicey_hip.h and linked
to libspicey_hip.so): no Python, no C++, no torch anywhere near the boundary."""
import os
import subprocess

import pytest

from conftest import REPO

SRC = os.path.join(REPO, "tests", "cabi", "cabi_rc.c")


def _build(tmp_path):
    from spicey_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    exe = str(tmp_path / "cabi_rc")
    libdir = os.path.dirname(lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", "-o", exe, SRC, "-L", libdir, "-lspicey_hip", "-lm",
                    f"-Wl,-rpath,{libdir}"], check=True)
    return exe


def test_c_consumer_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked run")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 4 and "no HIP device" in r.stdout  # SPICEY_ERR_NO_DEVICE: no CPU path in the library


@pytest.mark.gpu
def test_c_consumer_runs_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi ok" in r.stdout
