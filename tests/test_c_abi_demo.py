"""examples/c_abi_demo.cpp: a plain C++ host (no Python, no torch, system ROCm runtime) drives the C ABI.
CPU: it must build and fail LOUDLY without a device (exit 77, "no CPU fallback").  GPU: it must self-check and exit 0."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_demo(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / "c_abi_demo")
    lib = os.path.join(ROOT, "gpu_olap_engine_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.cpp"),
                           "-L", lib, "-lgolap_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", exe])
    return exe


def test_demo_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([build_demo(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_demo_runs_on_gpu(tmp_path):
    r = subprocess.run([build_demo(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "c_abi_demo ok" in r.stdout
