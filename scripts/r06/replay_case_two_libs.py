"""Replays ONE case of a fuzz run on two builds of the library (the tree's and loik_amd/lib/libloik_amd_before_solveinit.so) and compares what the
device returned, bit for bit: is a deviation from the oracle older than the change between the builds?  usage: replay_case_two_libs.py ncase seed only"""
import sys, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 4:   # child: one replay, results to an .npz
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import numpy as np
    if sys.argv[4] == "old":
        import loik_amd.capi as capi
        capi._LIB_PATH = os.path.join(ROOT, "loik_amd", "lib", "libloik_amd_before_solveinit.so")
        capi.ABI_VERSION = int(os.environ.get("OLD_ABI", "601"))
    import fuzz_engines
    box = {}
    fuzz_engines.fuzz(int(sys.argv[1]), int(sys.argv[2]), verbose=False, only=int(sys.argv[3]), flat_bias=0.7, capture=lambda d: box.update(d))
    g = box["got"]
    np.savez("/tmp/replay_%s.npz" % sys.argv[4], iter=np.asarray(g["iter"]), z=np.asarray(g["z"]), conv=np.asarray(g["converged"]), dz=box["dz"])
    sys.exit(0)
import numpy as np
for which in ("new", "old"):
    subprocess.check_call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:4] + [which])
a, b = np.load("/tmp/replay_new.npz"), np.load("/tmp/replay_old.npz")
print("iterations identical:", bool(np.array_equal(a["iter"], b["iter"])), " z identical:", bool(np.array_equal(a["z"], b["z"])),
      " flags identical:", bool(np.array_equal(a["conv"], b["conv"])), " max |dz| against the oracle, new / old: %.3e / %.3e" % (a["dz"].max(), b["dz"].max()))
