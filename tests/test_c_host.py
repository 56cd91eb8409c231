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


def test_the_hashes_in_the_c_program_are_the_goldens():
    """CPU: the SHA-256 strings embedded in remap_host.c are tests/golden/{mid,full}.json's - the reference's own output hashes."""
    import json

    text = open(SRC).read()
    mid = json.load(open(os.path.join(ROOT, "tests", "golden", "mid.json")))["M_ident_eqd_rot0"]
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "full.json")))["c2"]
    for case in (mid, full):
        assert case["u8_sha256"] in text and case["frame_sha256"] in text


def test_sha256_of_the_c_program(tmp_path):
    """CPU: the program's own SHA-256 routine against hashlib (compiled alone, -DSHA_SELFTEST)."""
    import hashlib

    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    exe = str(tmp_path / "sha_selftest")
    res = subprocess.run([gcc, "-std=c99", "-O1", "-DSHA_SELFTEST", "-I", os.path.join(ROOT, "include"), SRC, "-lm", "-o", exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000, 100003):
        data = bytes((i * 131 + 7) & 0xFF for i in range(n))
        f = tmp_path / "blob"
        f.write_bytes(data)
        out = subprocess.run([exe, str(f)], capture_output=True, text=True)
        assert out.stdout.strip() == hashlib.sha256(data).hexdigest(), n


@pytest.mark.gpu
def test_c_host_program_remaps_agrees_with_itself_and_with_the_reference(tmp_path):
    exe = _compile(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "c host ok" in res.stdout and res.stdout.count("fast == faithful, batch frame == single launch, == index-map gather") == 2, res.stdout
    assert res.stdout.count("input frame == the golden's, output == the reference's") == 2, res.stdout
