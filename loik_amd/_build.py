"""Build libloik_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libloik_amd.so")
SOURCES = ["loik_host.hip", "loik_flat_kernels.hip", "models.c"]
HEADERS = ["loik_device.hpp", "loik_tail.hpp", "loik_lean.hpp", "loik_flat.hpp", "loik_flat2.hpp", "loik_flat_inst.hpp", "loik_passes.hpp",
           os.path.join("..", "..", "include", "loik_amd.h"), os.path.join("..", "..", "include", "loik_amd_models.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# Code generation of the flat iteration kernels' translation unit (loik_flat_kernels.hip; why: csrc/loik_flat_inst.hpp): neighbouring LDS
# accesses stay single 64-bit instructions -- neither the IR load/store vectorizer (128-bit accesses, ds_read2_b64 where the alignment is
# 8) nor the machine-level SI load/store optimizer (ds_read2_b64 / ds_write2_b64 / ds_read2st64_b64) merges them.
# Round 6: the machine scheduler of that unit is LLVM's iterative ILP scheduler (-amdgpu-sched-strategy=iterative-ilp) instead of the default
# max-occupancy strategy -- the kernels' occupancy is fixed by their register budget (amdgpu_waves_per_eu), what they wait for is the dependent
# chain of an iteration (LDS round trips, fp64 latencies): the lone iteration of k_flat2 2.18 -> 2.04 us, of k_flat1 2.97 -> 2.67 us, the headline
# batch in arrival order 8.46 -> 8.06-8.14 ms, the whole body 15.5 -> 14.0-14.2 ms (profiles/r06_f_sched_strategy_ab.txt: max-ilp, max-memory-clause
# and iterative-minreg measured beside it).  Same instructions, another order: results are bit-identical (the parity tests and the fuzz).
FLAT_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-mllvm", "-amdgpu-load-store-vectorizer=0",
              "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS + [os.path.join("..", "_build.py")]:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False, extra_flags=(), flat_flags=None, host_flags=()):
    """Compile every HIP source for gfx950.  hipcc cross-compiles without a GPU.
    extra_flags go to both translation units; flat_flags (default FLAT_FLAGS) only to the flat iteration kernels', host_flags (default
    none) only to the other unit.
    A -DLOIKB_TAIL_PROF build (phase timelines: its device-side counters are globals every kernel writes and the host reads) is ONE
    translation unit, compiled with flat_flags throughout."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    inc = os.path.join(HERE, "..", "include")
    flat_flags = list(FLAT_FLAGS if flat_flags is None else flat_flags)
    extra_flags = list(extra_flags)
    c_obj = os.path.join(LIBDIR, "models.o")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-I", inc, "-c", os.path.join(CSRC, "models.c"), "-o", c_obj])
    base = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-std=c++17", "-fPIC", "-I", inc, "-Wall", "-Wno-unused-function"]
    if verbose:
        base.insert(1, "-Rpass-analysis=kernel-resource-usage")
    # (-DLOIKB_TAIL_PROF alone: one unit, the default schedule everywhere -- the phase timelines of rounds 3-6.  With -DLOIKB_TAIL_PROF_TWO_UNITS beside
    #  it: the two units as shipped, each with its own copy of the counters: the timeline of the SHIPPED schedule of the flat kernels)
    single_tu = any(f.startswith("-DLOIKB_TAIL_PROF") for f in extra_flags) and "-DLOIKB_TAIL_PROF_TWO_UNITS" not in extra_flags
    if single_tu:
        # (the iterative scheduler is for the flat unit only: it crashes the compiler on one of the other kernels)
        flat_flags = [f for i, f in enumerate(flat_flags) if not (f.startswith("-amdgpu-sched-strategy") or (f == "-mllvm" and i + 1 < len(flat_flags) and flat_flags[i + 1].startswith("-amdgpu-sched-strategy")))]
        cmd = base + ["-shared", "-x", "hip", os.path.join(CSRC, "loik_host.hip"), "-x", "none", c_obj, "-o", LIB] + flat_flags + extra_flags
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return LIB
    host_obj, flat_obj = os.path.join(LIBDIR, "loik_host.o"), os.path.join(LIBDIR, "loik_flat_kernels.o")
    cmds = [base + ["-DLOIKB_FLAT_SEPARATE_TU", "-c", "-x", "hip", os.path.join(CSRC, "loik_host.hip"), "-o", host_obj] + extra_flags + list(host_flags),
            base + ["-DLOIKB_FLAT_SEPARATE_TU", "-c", "-x", "hip", os.path.join(CSRC, "loik_flat_kernels.hip"), "-o", flat_obj] + flat_flags + extra_flags]
    import sys
    import tempfile
    procs, errs = [], []
    for cmd in cmds:   # (the two compile side by side: ~1 minute each)
        if verbose:
            print(" ".join(cmd))
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen(cmd, stderr=errs[-1]))
    rcs = [p.wait() for p in procs]
    for e in errs:   # (the host pass of the flat kernels' unit does not know the device's feature and says so four times: not news)
        e.seek(0)
        for line in e:
            if "is not a recognized feature for this target" not in line:
                sys.stderr.write(line)
        e.close()
    if any(rcs):
        raise subprocess.CalledProcessError(next(r for r in rcs if r), cmds[[bool(r) for r in rcs].index(True)])
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", host_obj, flat_obj, c_obj, "-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="-f" in sys.argv, verbose="-v" in sys.argv))
