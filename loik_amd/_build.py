"""Build libloik_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libloik_amd.so")
SOURCES = ["loik_host.hip", "models.c"]
HEADERS = ["loik_device.hpp", "loik_tail.hpp", "loik_lean.hpp", "loik_flat.hpp", "loik_flat2.hpp", "loik_passes.hpp", os.path.join("..", "..", "include", "loik_amd.h"),
           os.path.join("..", "..", "include", "loik_amd_models.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source for gfx950.  hipcc cross-compiles without a GPU."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    inc = os.path.join(HERE, "..", "include")
    objs = []
    c_obj = os.path.join(LIBDIR, "models.o")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-I", inc, "-c", os.path.join(CSRC, "models.c"), "-o", c_obj])
    objs.append(c_obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-std=c++17", "-fPIC", "-shared", "-I", inc,
           "-Wall", "-Wno-unused-function", "-x", "hip", os.path.join(CSRC, "loik_host.hip"),
           "-x", "none", c_obj, "-o", LIB] + list(extra_flags)
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="-f" in sys.argv, verbose="-v" in sys.argv))
