"""The C-ABI consumed from plain C (gcc), exactly as a recompiled reference glue would."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "scs-python_amd", "scs")


def _compile(tmp_path):
    exe = str(tmp_path / "cabi_smoke")
    cmd = ["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c"),
           "-L", LIBDIR, "-lscs_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_c_consumer_compiles_and_links(tmp_path):
    """CPU: headers are valid C and every symbol the C program uses resolves against libscs_hip.so."""
    exe = _compile(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    # without a GPU the program reports it and exits 2 (no crash, no CPU fallback)
    assert out.returncode in (0, 2), out.stdout + out.stderr


@pytest.mark.gpu
def test_c_consumer_solves_on_device(tmp_path):
    exe = _compile(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)  # (the three members of its batch call are ONE group)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout and out.stdout.count("identical") == 3, out.stdout
