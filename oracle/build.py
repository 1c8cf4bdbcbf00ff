"""Compile the C oracle (oracle/loik_ref.c) with gcc.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(native=False, out=None, force=False):
    """Build libloik_ref.so.  ``native=True`` adds -march=native (used by bench.py's cpu_baseline so the
    CPU number is an honest best effort on the box it runs on); the default build is portable because
    the prebuilt .so travels from the dev container to the GPU box."""
    src = os.path.join(HERE, "loik_ref.c")
    hdr = os.path.join(HERE, "loik_ref.h")
    if out is None:
        out = os.path.join(HERE, "libloik_ref_native.so" if native else "libloik_ref.so")
    if (not force and os.path.exists(out)
            and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return out
    cmd = ["gcc", "-O3", "-ffp-contract=off", "-fPIC", "-std=c11", "-fopenmp", "-shared", "-o", out, src, "-lm"]
    if native:
        cmd.insert(2, "-march=native")
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build())
