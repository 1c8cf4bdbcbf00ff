"""Where does the one deviating instance of a pinned fuzz case part from the oracle?  Both solvers with logging = 1 on that instance alone:
the lists of LoikSolverInfo (mu, primal / dual residual per iteration) side by side.  usage: near_tie_probe.py <fixture name> <instance in the fixture>"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import loik_amd
from oracle import ref
from test_fuzz_regressions import load_case
name = sys.argv[1]
fx, model, prm, env, kw, refs, args = load_case(name)
dz_full = fx["gpu_dz_full_batch"]
b = int(sys.argv[2]) if len(sys.argv) > 2 else int(np.argmax(dz_full))
print("fixture", name, "instance", b, "(batch instance %d)" % int(fx["pick"][b]), "|dz| in the full batch %.3e" % dz_full[b])
sub = lambda a, nd: a[b:b + 1] if np.ndim(a) == nd and np.shape(a)[0] == np.shape(args[0])[0] else a
a1 = (sub(args[0], 2), args[1], args[2], args[3], sub(args[4], 4), sub(args[5], 3), sub(args[6], 2), sub(args[7], 2))
r = ref.RefSolver(model, **prm)   # (the oracle keeps its lists always)
if refs is None:
    r.Solve(a1[0][0], a1[1], a1[2], a1[3], a1[4] if a1[4].ndim == 3 else a1[4][0], a1[5][0], a1[6] if a1[6].ndim == 1 else a1[6][0], a1[7] if a1[7].ndim == 1 else a1[7][0])
else:
    r.SolveInit(a1[0][0], a1[1], a1[2], a1[3], a1[4] if a1[4].ndim == 3 else a1[4][0], a1[5][0], a1[6] if a1[6].ndim == 1 else a1[6][0], a1[7] if a1[7].ndim == 1 else a1[7][0])
    r.UpdateReferences(*refs); r.Solve()
for k, v in env.items():
    os.environ[k] = v
NB = 64   # (the instance 64 times: a logged solve stays on the flat engine from 64 instances)
rep = lambda a, nd: np.repeat(a, NB, axis=0) if np.ndim(a) == nd and np.shape(a)[0] == 1 else a
a64 = (rep(a1[0], 2), a1[1], a1[2], a1[3], rep(a1[4], 4), rep(a1[5], 3), rep(a1[6], 2), rep(a1[7], 2))
s = loik_amd.BatchedLoik(model, NB, **prm, **kw, logging=1, eq_c_capacity=int(fx["nc"]) + int(fx["spare"]))
if refs is None:
    s.Solve(*a64)
else:
    s.SolveInit(*a64); s.UpdateReferences(*refs); s.Solve()
print("plan:", s.plan()[:160])
si = s.solver_info(); g_mu = si["mu_list"][0]; g_p = si["primal_residual_list"][0]; g_d = si["dual_residual_list"][0]
o_mu = np.asarray(r.solver_info(6)); o_p = np.asarray(r.solver_info(2)); o_d = np.asarray(r.solver_info(5))
logged = np.flatnonzero(np.asarray(g_mu) != 0)   # (a hand-over's first iterations run in k_solve, which keeps no lists: zeros)
n = min(len(o_mu), int(logged[-1]) + 1 if logged.size else 0)
print("iterations: oracle %d, here %d (%d of them logged);  final |dz| %.3e" % (len(o_mu), int(np.asarray(s.get("iter"))[0]), logged.size, np.abs(np.asarray(s.get("z"))[0] - r.z).max()))
first = next((k for k in range(n) if g_mu[k] != 0 and g_mu[k] != o_mu[k]), None)
print("first iteration whose mu differs:", first)
lo = max(0, (first if first is not None else n) - 3)
for k in range(lo, min(n, lo + 8)):
    print("  it %3d  mu here %-8g oracle %-8g   primal %.6e / %.6e   dual %.6e / %.6e   primal/dual here %.6f oracle %.6f" % (k + 1, g_mu[k], o_mu[k], g_p[k], o_p[k], g_d[k], o_d[k], g_p[k] / g_d[k], o_p[k] / o_d[k]))
