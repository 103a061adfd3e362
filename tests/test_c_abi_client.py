"""The drop-in boundary from the other side: a C99 program (tests/c_abi/client.c) that includes include/dlpd.h, dlopen()s
the library and calls it with plain pointers and sizes -- no Python, no torch.  On CPU it drives the emulated library
(host pointers), with ``-m gpu`` the product libdlpd.so on device memory it gets from the HIP runtime itself."""
import os
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = os.path.join(ROOT, "tests", "c_abi", "client.c")


def _build(tmp_path):
    exe = str(tmp_path / "dlpd_c_client")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-ldl", "-lm"])
    return exe


def test_c_client_drives_the_emulated_library(tmp_path, emu):
    r = subprocess.run([_build(tmp_path), emu.path, "host"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "c-abi client ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_client_drives_libdlpd_on_the_gpu(tmp_path):
    import __graft_entry__ as entry
    entry.build()
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([_build(tmp_path), entry.LIB, "hip"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "c-abi client ok (hip" in r.stdout, r.stdout + r.stderr
