"""The C++ face of the boundary: include/reflector_ekf_slam_amd/*.hpp must compile against the C ABI
(the Eigen/ROS-typed adapters are guarded by __has_include and vanish on this image) and, on a GPU,
the RAII wrapper must run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import __graft_entry__
    if not os.path.exists(os.path.join(ROOT, "reflector_ekf_slam_amd", "librdet.so")):
        __graft_entry__.build()
    exe = str(tmp_path / "adapter_smoke")
    lib = os.path.join(ROOT, "reflector_ekf_slam_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"),
           "-L", lib, "-lrekf", "-lrdet", "-lrgrid", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    return exe


def test_cpp_wrappers_compile_and_link(tmp_path):
    exe = _build(tmp_path)
    assert subprocess.run([exe, "compile-only"]).returncode == 0


@pytest.mark.gpu
def test_cpp_wrapper_runs_on_gpu(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ADAPTER_OK n=7" in out.stdout, out.stdout + out.stderr
