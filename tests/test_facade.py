"""The header-only C++ facade (reference names / signatures over the C ABI) compiles with plain g++ and
links against libsgs_hip.so; on a GPU box it also runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "facade_main")


def _build():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    lib = os.path.join(ROOT, "sg-slam_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "sg-slam_amd", "include"),
           os.path.join(ROOT, "tests", "facade_main.cpp"), "-o", EXE, "-L", lib, "-lsgs_hip", "-L/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_facade_compiles_and_links():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "facade ok" in out.stdout or "no device" in out.stdout


@pytest.mark.gpu
def test_facade_runs_on_gpu():
    if not os.path.exists(EXE):
        _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "facade ok" in out.stdout, out.stdout + out.stderr
