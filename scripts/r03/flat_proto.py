"""Design study (round 3): the ADMM iteration WITHOUT level loops, in numpy, against the CPU oracle.

The two recursions of an iteration (leaf->root p / r, root->leaf nu / v; loik-loid-optimized.hxx:31-81, :102-163) are the
sparse LDL^T solve of  (J^T H J + mu I) nu = -(J^T p^base + w - mu z)  in tree order.  With everything expressed at the
world origin the transports disappear:

    tau_a  = (w_a - mu z_a) + S^w_a . sum_{d in subtree(a)} p^base,w_d          (subtree sum of 6-vectors)
    r'     = W tau,          W = (I + L)^-1,   L_{a,d} = S^w_a . UDinv^w_d   (d a descendant of a)   -- scalars per pair
    nu     = -W^T (Dinv r')
    v^w_i  = sum_{a in ancestors*(i)} S^w_a nu_a                                (path sum of 6-vectors)
    f^w_i  = sum_{d in subtree(i)} phi^w_d ,  phi_d = H^base_d v_d + p^base_d   (force balance; subtree sum)

and g_i = A^T y_i - phi_i (BwdPass2 in closed form).  This script checks that the reorganised arithmetic reproduces the
oracle's iteration counts / flags / z on the headline workload (it is what k_flat computes on the device).
Usage: python scripts/r03/flat_proto.py [B] [seed]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import loik_amd  # noqa: E402
from loik_amd import workloads  # noqa: E402
from oracle import ref  # noqa: E402
sys.path.insert(0, "tests")
from flat_numpy import Flat  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0x101C + 3
    wl = workloads.talos_c3(B, seed=seed)
    t0 = time.time()
    out = ref.solve_batch(wl["model"], wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=8, **wl["params"])
    t1 = time.time()
    fl = Flat(wl).solve()
    t2 = time.time()
    same = fl["iters"] == out["iters"]
    err = np.abs(fl["z"] - out["z"]).max(1)
    print("B=%d oracle %.1fs proto %.1fs" % (B, t1 - t0, t2 - t1))
    print("identical iteration counts: %d / %d (%.2f %%)" % (same.sum(), B, 100 * same.mean()))
    print("max |dz| same-count: %.3e   others: %.3e" % (err[same].max(), err[~same].max() if (~same).any() else 0))
    print("flags equal (same): conv %s pinf %s" % (np.array_equal(fl["converged"][same], out["converged"][same]),
                                                 np.array_equal(fl["primal_infeasible"][same], out["primal_infeasible"][same])))
    print("converged %d / %d ; pinf %d / %d ; at max_iter %d / %d" % (fl["converged"].sum(), out["converged"].sum(),
          fl["primal_infeasible"].sum(), out["primal_infeasible"].sum(), (fl["iters"] >= 999).sum(), (out["iters"] >= 999).sum()))
    off = np.nonzero(~same)[0]
    for b in off[:20]:
        print("  off-count instance %d: proto %d oracle %d  |dz| %.2e conv %d/%d pinf %d/%d" % (
            b, fl["iters"][b], out["iters"][b], err[b], fl["converged"][b], out["converged"][b],
            fl["primal_infeasible"][b], out["primal_infeasible"][b]))


if __name__ == "__main__":
    main()
