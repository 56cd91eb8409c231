"""The C ABI from a plain C program (tests/c_host/remap_host.c: no Python, no torch, no HIP headers - only
include/photonbend_hip.h and the shared library): plan, batch remap, faithful mode, index map + gather, on two geometries."""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "remap_host.c")


def _compile(tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    from photonbend_amd.build import LIB_PATH, build_library

    build_library()
    libdir = os.path.dirname(LIB_PATH)
    exe = str(tmp_path / "remap_host")
    cmd = [gcc, "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-L", libdir, "-lphotonbend_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_header_compiles_as_c99_and_links(tmp_path):
    """CPU: include/photonbend_hip.h is valid C99 and every entry point the program uses resolves against the library."""
    _compile(tmp_path)


@pytest.mark.gpu
def test_c_host_program_remaps_and_agrees_with_itself(tmp_path):
    exe = _compile(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "c host ok" in res.stdout and res.stdout.count("fast == faithful, batch frame == single launch, == index-map gather") == 2, res.stdout
