"""the planner-style outer loop -- T successive targets per instance, warm-started tailored
solves (loik-loid-optimized.hpp:596-695), q advanced by q <- q + dt z between steps.
Compares (a) q resident on the device (loikb_integrate + Solve(None, ...)): no per-step upload of q, targets b are the
only per-step input, with (b) the reference caller's pattern: integrate on the host, upload q every step.
Prints one JSON line per variant: planner steps x instances per second, converged solves/s, per-step times."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loik_amd
from loik_amd import workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dt = 0.1
wl = workloads.talos_c3(B, seed=21)
model, link = wl["model"], int(wl["c_ids"][0])
prm = dict(wl["params"], warm_start=True)
rng = np.random.default_rng(77)
mode = sys.argv[3] if len(sys.argv) > 3 else "random"
if mode == "tracking":
    # servo-style: the same joint-space target velocity at every step, only q (hence J(q), hence b) drifts -- the case
    # the warm start is made for
    nu_star = np.repeat(rng.uniform(-0.5, 0.5, size=(1, B, model.nv)), T, axis=0)
else:
    # a fresh random target every step (the C4 generator): the warm start buys little
    nu_star = rng.uniform(-0.5, 0.5, size=(T, B, model.nv))


def run(resident):
    s = loik_amd.BatchedLoik(model, B, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    q = wl["q"].copy()
    t_solve, t_total, solved, iters = [], [], 0, 0
    for t in range(T):
        # re-target on the host (a planner would do this on the device too; it is outside the timed solve either way)
        b_t = workloads.link_velocity(model, q, nu_star[t], link)[:, None, :]
        t0 = time.perf_counter()
        if resident:
            if t > 0:
                s.integrate(dt)
            s.Solve(None, link, wl["Ais"], b_t)
        else:
            s.Solve(q, link, wl["Ais"], b_t)
        t1 = time.perf_counter()
        z = s.get("z")
        q = q + dt * z
        t_total.append(time.perf_counter() - t0)
        t_solve.append(t1 - t0)
        solved += int(s.get("converged").sum())
        iters += int(s.get("iter").sum())
    s.close()
    return dict(config="C4 outer loop talos32 B=%d T=%d warm-started tailored solves, %s targets" % (B, T, mode),
                q="resident on device (loikb_integrate)" if resident else "integrated on host, uploaded every step",
                planner_steps_per_s=B * T / sum(t_solve), solves_per_s=solved / sum(t_solve),
                ms_per_step=[round(x * 1e3, 2) for x in t_solve], mean_iterations=iters / (B * T),
                solved_fraction=solved / (B * T))


for resident in (True, False, True, False):
    print(json.dumps(run(resident)), flush=True)
