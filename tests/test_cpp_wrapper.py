"""Builds and runs the C++ parity test of the drop-in wrapper (include/loik_amd/loik.hpp) on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_wrapper")


def build():
    src = os.path.join(ROOT, "tests", "cpp", "test_wrapper.cpp")
    libdir = os.path.join(ROOT, "loik_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
           "-L", libdir, "-lloik_amd", "-L", os.path.join(ROOT, "oracle"), "-lloik_ref",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_cpp_wrapper_compiles():
    """CPU: the header-only wrapper compiles against the C-ABI and links"""
    build()
    assert os.path.exists(EXE)


def test_pinocchio_adapter_round_trip():
    """CPU: include/loik_amd/pinocchio_adapter.hpp against a model type with pinocchio::Model's interface (the image has no
    Pinocchio / Eigen): every built-in table and a model with every joint type go through and come back equal"""
    src = os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "test_adapter")
    libdir = os.path.join(ROOT, "loik_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-lloik_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "all adapter checks passed" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_wrapper_matches_oracle():
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all wrapper checks passed" in out.stdout


def test_cpp_single_call_benchmark_compiles():
    """CPU: scripts/r06/bench_cpp_single_call.cpp -- what a C++ drop-in caller pays per problem through the mirror (profiles/r06_x_cpp_single_call.jsonl) --
    compiles against the header and links; its input file is in place"""
    src = os.path.join(ROOT, "scripts", "r06", "bench_cpp_single_call.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "bench_cpp_single_call")
    libdir = os.path.join(ROOT, "loik_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-lloik_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe) and os.path.exists(os.path.join(ROOT, "scripts", "r06", "bench_cpp_single_call_input.txt"))
