"""ctypes binding of include/loik_amd.h (the drop-in C-ABI) + a host-side mirror of the reference's solver API.

`BatchedLoik` keeps the reference's method names and argument meaning
(`SolveInit`, `Solve`, getters: /root/reference/include/loik/loik-loid-optimized.hpp:335-361, :368-455, :475-580,
:596-695; /root/reference/include/loik/task-solver-base.hpp:87-141) so the parity tests read like the
reference's own tests.  All compute happens in libloik_amd.so on the GPU; nothing here falls back to a CPU path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libloik_amd.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class LoikError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("loik_amd error %d: %s" % (code, msg))
        self.code = code


class ModelDesc(C.Structure):
    _fields_ = [("njoints", C.c_int), ("nq", C.c_int), ("nv", C.c_int), ("parents", _ip), ("jtype", _ip),
                ("axis", _dp), ("idx_q", _ip), ("idx_v", _ip), ("placement", _dp),
                ("comp_first", _ip), ("comp_count", _ip), ("comp_jtype", _ip), ("comp_axis", _dp), ("comp_placement", _dp),
                ("pitch", _dp), ("comp_pitch", _dp)]


class Options(C.Structure):
    _fields_ = [("max_iter", C.c_int),
                ("tol_abs", C.c_double), ("tol_rel", C.c_double), ("tol_primal_inf", C.c_double),
                ("tol_dual_inf", C.c_double), ("rho", C.c_double), ("mu", C.c_double),
                ("mu_equality_scale_factor", C.c_double), ("mu_update_strat", C.c_int),
                ("num_eq_c", C.c_int), ("eq_c_dim", C.c_int), ("warm_start", C.c_int),
                ("tol_tail_solve", C.c_double), ("verbose", C.c_int), ("logging", C.c_int),
                ("batch", C.c_int), ("device", C.c_int), ("precision", C.c_int), ("flags", C.c_int),
                ("max_launch_iters", C.c_int), ("compact_min_instances", C.c_int),
                ("tail_max_instances", C.c_int), ("eq_c_capacity", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("instance_iterations", C.c_ulonglong), ("launches", C.c_int), ("n_unfinished", C.c_int),
                ("compactions", C.c_int), ("tail_instances", C.c_int), ("tail_ms", C.c_double),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("bytes_per_instance_iteration", C.c_double),
                ("tail_instance_iterations", C.c_ulonglong), ("tail_launches", C.c_int), ("team", C.c_int), ("chunks", C.c_int),
                ("solve_busy_ms", C.c_double), ("tail_busy_ms", C.c_double), ("lean_launches", C.c_int),
                ("lean_escaped", C.c_int), ("hslots_ms", C.c_double), ("lean_requeues", C.c_int), ("flat_launches", C.c_int), ("queue_dry_ms", C.c_double),
                ("flat_split_launches", C.c_int), ("flat_ordered", C.c_int), ("flat_built", C.c_int),
                ("flat_probe_launches", C.c_int), ("probe_ms", C.c_double)]


ABI_VERSION = 602   # LOIKB_VERSION of include/loik_amd.h this binding matches (struct layouts, entry points)

# enums of loik_amd.h
F64, F32 = 0, 1
OPT_FIXED_ITERS, OPT_NO_H_CACHE, OPT_NO_COMPACTION, OPT_OWN_STREAM, OPT_F32_ACCURATE, OPT_ORDER_FROM_PREVIOUS = 1, 2, 4, 8, 16, 32
IN_DEVICE, A_SHARED, BOUNDS_SHARED, B_SHARED, Q_SHARED = 1, 2, 4, 8, 16
OUT_DEVICE = 1

_VEC_FIELDS = ["z", "nu", "w", "Stf_plus_w", "r", "Dinv", "vis", "fis", "g", "pis", "UDinv", "His", "liMi", "yis",
               "Aty", "iter", "converged", "primal_infeasible", "status"]
_SCALAR_FIELDS = ["primal_residual", "dual_residual", "primal_residual_task", "primal_residual_slack",
                  "dual_residual_v", "dual_residual_nu", "tol_primal", "tol_dual", "mu", "mu_eq", "mu_ineq",
                  "delta_x_qp_inf_norm", "delta_z_qp_inf_norm", "delta_y_qp_inf_norm", "A_qp_T_delta_y_qp_inf_norm",
                  "ub_qp_T_delta_y_qp_plus", "lb_qp_T_delta_y_qp_minus", "delta_fis_inf_norm", "delta_yis_inf_norm",
                  "delta_w_inf_norm", "delta_vis_inf_norm", "delta_nu_inf_norm", "Av_inf_norm", "nu_inf_norm",
                  "Href_v_inf_norm", "g_inf_norm", "Stf_plus_w_inf_norm", "primal_infeasibility_cond_1",
                  "primal_infeasibility_cond_2", "tail_solve_iter"]
FIELD_ID = {n: i for i, n in enumerate(_VEC_FIELDS)}
FIELD_ID["q"] = 96
FIELD_ID["mu_updates"] = 97
FIELD_ID["primal_residual_vec"] = 98
FIELD_ID["dual_residual_vec"] = 99
FIELD_ID.update({n: 32 + i for i, n in enumerate(_SCALAR_FIELDS)})

# every symbol include/loik_amd.h and include/loik_amd_models.h declare
EXPORTED_SYMBOLS = [
    "loikb_create", "loikb_destroy", "loikb_set_stream", "loikb_solve_init", "loikb_solve", "loikb_solve_full",
    "loikb_solve_tailored", "loikb_set_max_iter", "loikb_set_rho", "loikb_set_mu", "loikb_set_tol",
    "loikb_set_tol_primal_inf", "loikb_set_tol_tail_solve", "loikb_set_warm_start", "loikb_get", "loikb_get_results", "loikb_get_stats",
    "loikb_batch", "loikb_nv", "loikb_njoints", "loikb_last_error", "loikb_status_string", "loikb_version",
    "loikb_device_count", "loikb_sweep_schedule", "loikb_integrate", "loikb_synchronize", "loikb_plan_string", "loikb_pass",
    "loikb_update_references", "loikb_update_eq_constraint", "loikb_add_eq_constraint", "loikb_remove_eq_constraint",
    "loikb_num_eq_c", "loikb_eq_c_capacity", "loikb_active_constraint_ids", "loikb_get_solver_info", "loikb_solver_info_rows_cap", "loikb_solver_info_truncated", "loikb_builtin_model", "loikb_builtin_joint_name",
    "loikb_builtin_joint_id", "loikb_flat_schedule"]

_lib = None


def lib():
    """Load libloik_amd.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError("loik_amd: %s is missing -- build the HIP extension first "
                          "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
                          % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    L.loikb_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(Options), C.POINTER(C.c_void_p)]
    L.loikb_destroy.argtypes = [C.c_void_p]
    L.loikb_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    sig = [C.c_void_p, C.c_void_p, _dp, _dp, _ip, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
           C.c_int]
    L.loikb_solve_init.argtypes = sig
    L.loikb_solve_full.argtypes = sig
    L.loikb_solve.argtypes = [C.c_void_p]
    L.loikb_integrate.argtypes = [C.c_void_p, C.c_double]
    L.loikb_synchronize.argtypes = [C.c_void_p]
    L.loikb_plan_string.argtypes = [C.c_void_p]
    L.loikb_pass.argtypes = [C.c_void_p, C.c_int]
    L.loikb_plan_string.restype = C.c_char_p
    L.loikb_sweep_schedule.argtypes = [_ip, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip]
    L.loikb_solve_tailored.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.loikb_update_references.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
    L.loikb_update_eq_constraint.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.loikb_add_eq_constraint.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.loikb_remove_eq_constraint.argtypes = [C.c_void_p, C.c_int]
    L.loikb_num_eq_c.argtypes = [C.c_void_p]
    L.loikb_eq_c_capacity.argtypes = [C.c_void_p]
    L.loikb_active_constraint_ids.argtypes = [C.c_void_p, _ip, C.c_int]
    L.loikb_get_solver_info.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _ip]
    L.loikb_solver_info_rows_cap.argtypes = [C.c_void_p]
    L.loikb_solver_info_truncated.argtypes = [C.c_void_p]
    L.loikb_set_max_iter.argtypes = [C.c_void_p, C.c_int]
    for n in ["loikb_set_rho", "loikb_set_mu", "loikb_set_tol_primal_inf", "loikb_set_tol_tail_solve"]:
        getattr(L, n).argtypes = [C.c_void_p, C.c_double]
    L.loikb_set_tol.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.loikb_set_warm_start.argtypes = [C.c_void_p, C.c_int]
    L.loikb_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.loikb_get_results.argtypes = [C.c_void_p, C.c_uint, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.loikb_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    for n in ["loikb_batch", "loikb_nv", "loikb_njoints"]:
        getattr(L, n).argtypes = [C.c_void_p]
    L.loikb_last_error.restype = C.c_char_p
    L.loikb_status_string.argtypes = [C.c_int]
    L.loikb_status_string.restype = C.c_char_p
    L.loikb_builtin_model.argtypes = [C.c_char_p, C.POINTER(ModelDesc), C.POINTER(_dp), C.POINTER(_dp)]
    L.loikb_builtin_joint_name.argtypes = [C.c_char_p, C.c_int]
    L.loikb_builtin_joint_name.restype = C.c_char_p
    L.loikb_builtin_joint_id.argtypes = [C.c_char_p, C.c_char_p]
    L.loikb_flat_schedule.argtypes = [_ip, C.c_int, _ip, C.c_int, _ip]
    if L.loikb_version() != ABI_VERSION:
        raise ImportError("loik_amd: %s has ABI version %d, this binding was written for %d -- rebuild the library"
                          % (_LIB_PATH, L.loikb_version(), ABI_VERSION))
    _lib = L
    return L


def device_count():
    return int(lib().loikb_device_count())


def sweep_schedule(parents, team, direction):
    """Step schedule of a tree sweep for a team of `team` wavefronts (host-only introspection, no device).
    Returns (joint[team][T], flags[team][T], slot[team][T], n_lds_slots); direction 0 = leaf->root, 1 = root->leaf."""
    parents = np.ascontiguousarray(parents, dtype=np.int32)
    nj = int(parents.shape[0])
    cap = nj
    joint = np.zeros((team, cap), dtype=np.int32)
    flags = np.zeros((team, cap), dtype=np.int32)
    slot = np.zeros((team, cap), dtype=np.int32)
    nslots = C.c_int(0)
    T = lib().loikb_sweep_schedule(parents.ctypes.data_as(_ip), nj, int(team), int(direction), cap,
                                   joint.ctypes.data_as(_ip), flags.ctypes.data_as(_ip), slot.ctypes.data_as(_ip),
                                   C.byref(nslots))
    if T < 0:
        _check(T)
    return joint[:, :T].copy(), flags[:, :T].copy(), slot[:, :T].copy(), int(nslots.value)


def flat_schedule(parents):
    """The flat engine's schedule of a tree (host-only introspection, loikb_flat_schedule): None when the engine does not apply
    (loikb_last_error says why), else a dict: G, nanc, nscan, njmp and per lane depth / size / jmp / anc / red / helper / part."""
    parents = np.ascontiguousarray(parents, dtype=np.int32)
    meta = np.zeros(5, dtype=np.int32)
    L = lib()
    need = L.loikb_flat_schedule(parents.ctypes.data_as(_ip), int(parents.size), None, 0, meta.ctypes.data_as(_ip))
    if need < 0:
        _check(need)
    if not meta[0]:
        return None
    out = np.zeros(need, dtype=np.int32)
    _check(L.loikb_flat_schedule(parents.ctypes.data_as(_ip), int(parents.size), out.ctypes.data_as(_ip), need, meta.ctypes.data_as(_ip)))
    G = int(meta[1])
    rec = out.reshape(G, -1)
    return dict(G=G, nanc=int(meta[2]), nscan=int(meta[3]), njmp=int(meta[4]), depth=rec[:, 0].copy(), size=rec[:, 1].copy(),
                jmp=rec[:, 2:7].copy(), anc=rec[:, 7:23].copy(), red=rec[:, 23:31].copy(), helper=rec[:, 31].copy(),
                part=rec[:, 32:40].copy())


def _check(rc):
    if rc != 0:
        L = lib()
        msg = L.loikb_last_error().decode() if rc <= -20 or rc == -7 else L.loikb_status_string(rc).decode()
        if not msg:
            msg = L.loikb_status_string(rc).decode()
        raise LoikError(rc, msg)


# JointModelFreeFlyer / Spherical / Translation (LOIKB_J_FREEFLYER, _SPHERICAL, _TRANSLATION)
J_FREEFLYER, J_SPHERICAL, J_TRANSLATION = 9, 10, 11
# JointModelSphericalZYX, JointModelPlanar, JointModelRUBX / RUBY / RUBZ
J_SPHERICAL_ZYX, J_PLANAR, J_RUBX, J_RUBY, J_RUBZ = 12, 13, 14, 15, 16
J_COMPOSITE = 17  # JointModelComposite (Model(..., composite={joint: [(jtype, axis, placement12), ...]}); sub-joints: any type but a composite)
J_RUBU = 18       # JointModelRevoluteUnboundedUnaligned: nq 2 (cos, sin), nv 1, about `axis`
J_HX, J_HY, J_HZ, J_HU = 19, 20, 21, 22   # JointModelHelicalX / Y / Z / Unaligned: nq = nv = 1, S = [pitch a; a] (Model(..., pitch=[nj]))
JOINT_NQ = {J_FREEFLYER: 7, J_SPHERICAL: 4, J_TRANSLATION: 3, J_SPHERICAL_ZYX: 3, J_PLANAR: 4, J_RUBX: 2, J_RUBY: 2, J_RUBZ: 2, J_RUBU: 2}
JOINT_NV = {J_FREEFLYER: 6, J_SPHERICAL: 3, J_TRANSLATION: 3, J_SPHERICAL_ZYX: 3, J_PLANAR: 3}


class Model:
    """Kinematic tree with Pinocchio's member names: njoints, nq, nv, parents, jointPlacements (here `placement`,
    [nj][12] = R row-major + t), joint type / axis / idx_q / idx_v per joint."""

    def __init__(self, parents, jtype, axis, placement, names=None, q_lo=None, q_hi=None, name="custom", composite=None, pitch=None):
        self.parents = np.ascontiguousarray(parents, dtype=np.int32)
        self.jtype = np.ascontiguousarray(jtype, dtype=np.int32)
        self.axis = np.ascontiguousarray(axis, dtype=np.float64).reshape(-1, 3)
        self.placement = np.ascontiguousarray(placement, dtype=np.float64).reshape(-1, 12)
        self.njoints = int(self.parents.size)
        # JointModelHelical*: pitch [njoints] (translation along the axis per radian)
        self.pitch = None if pitch is None else np.ascontiguousarray(pitch, dtype=np.float64).reshape(self.njoints)
        # JointModelComposite: composite[i] = [(sub-joint type, axis [3], placement [12] relative to the previous sub-joint), ...]
        # (a helical sub-joint: a 4-tuple, the pitch last)
        self.composite = {int(i): [(int(e[0]), np.asarray(e[1], dtype=np.float64).reshape(3), np.asarray(e[2], dtype=np.float64).reshape(12))
                                   for e in subs] for i, subs in (composite or {}).items()}
        sub_pitch = {int(i): [float(e[3]) if len(e) > 3 else 0.0 for e in subs] for i, subs in (composite or {}).items()}
        self.comp_first = np.zeros(self.njoints, dtype=np.int32); self.comp_count = np.zeros(self.njoints, dtype=np.int32)
        ct, ca, cp, cpi = [], [], [], []
        for i in sorted(self.composite):
            self.comp_first[i] = len(ct); self.comp_count[i] = len(self.composite[i])
            for (t, a, P), ph in zip(self.composite[i], sub_pitch[i]):
                ct.append(t); ca.append(a); cp.append(P); cpi.append(ph)
        self.comp_pitch = np.ascontiguousarray(cpi if cpi else [0.0], dtype=np.float64)
        self.comp_jtype = np.ascontiguousarray(ct if ct else [0], dtype=np.int32)
        self.comp_axis = np.ascontiguousarray(ca if ca else [[0, 0, 0]], dtype=np.float64).reshape(-1, 3)
        self.comp_placement = np.ascontiguousarray(cp if cp else [[0] * 12], dtype=np.float64).reshape(-1, 12)

        def nq_of(i, t):
            if t == J_COMPOSITE:
                return sum(JOINT_NQ.get(st, 1) for st, _, _ in self.composite[i])
            return JOINT_NQ.get(t, 1)

        def nv_of(i, t):
            return sum(JOINT_NV.get(st, 1) for st, _, _ in self.composite[i]) if t == J_COMPOSITE else JOINT_NV.get(t, 1)
        # joints[i].nq() / nv() / idx_q() / idx_v() of Pinocchio: cumulative in joint order
        nqs = np.array([nq_of(i, int(t)) if i else 0 for i, t in enumerate(self.jtype)], dtype=np.int32)
        nvs = np.array([nv_of(i, int(t)) if i else 0 for i, t in enumerate(self.jtype)], dtype=np.int32)
        self.nqs, self.nvs = nqs, nvs
        self.nq, self.nv = int(nqs.sum()), int(nvs.sum())
        self.idx_q = np.ascontiguousarray(np.concatenate([[0], np.cumsum(nqs)[:-1]]), dtype=np.int32)
        self.idx_v = np.ascontiguousarray(np.concatenate([[0], np.cumsum(nvs)[:-1]]), dtype=np.int32)
        if self.njoints > 1:  # joint 0 (universe) has no coordinates; keep its index at 0 like before
            self.idx_q[0] = 0
            self.idx_v[0] = 0
        self.names = list(names) if names is not None else ["universe"] + ["joint%d" % i for i in range(1, self.njoints)]
        self.q_lo = None if q_lo is None else np.asarray(q_lo, dtype=np.float64)
        self.q_hi = None if q_hi is None else np.asarray(q_hi, dtype=np.float64)
        self.name = name

    def desc(self):
        return ModelDesc(self.njoints, self.nq, self.nv, self.parents.ctypes.data_as(_ip),
                         self.jtype.ctypes.data_as(_ip), self.axis.ctypes.data_as(_dp),
                         self.idx_q.ctypes.data_as(_ip), self.idx_v.ctypes.data_as(_ip),
                         self.placement.ctypes.data_as(_dp), self.comp_first.ctypes.data_as(_ip),
                         self.comp_count.ctypes.data_as(_ip), self.comp_jtype.ctypes.data_as(_ip),
                         self.comp_axis.ctypes.data_as(_dp), self.comp_placement.ctypes.data_as(_dp),
                         None if self.pitch is None else self.pitch.ctypes.data_as(_dp), self.comp_pitch.ctypes.data_as(_dp))

    def getJointId(self, name):
        return self.names.index(name)

    def random_configurations(self, rng, batch):
        """[batch][nq] configurations: uniform in [q_lo, q_hi] (unit box when none was given), the quaternion
        segments of free-flyer / spherical joints replaced by uniformly random unit quaternions (x, y, z, w)"""
        lo = -np.ones(self.nq) if self.q_lo is None else self.q_lo
        hi = np.ones(self.nq) if self.q_hi is None else self.q_hi
        q = rng.uniform(lo, hi, size=(batch, self.nq))
        def manifold_segments(t, o0):
            if t in (J_FREEFLYER, J_SPHERICAL):
                o = o0 + (3 if t == J_FREEFLYER else 0)
                qt = rng.normal(size=(batch, 4))
                q[:, o:o + 4] = qt / np.linalg.norm(qt, axis=1, keepdims=True)
            elif t in (J_PLANAR, J_RUBX, J_RUBY, J_RUBZ, J_RUBU):  # the (cos, sin) pair of a random angle
                o = o0 + (2 if t == J_PLANAR else 0)
                th = rng.uniform(-np.pi, np.pi, size=batch)
                q[:, o] = np.cos(th); q[:, o + 1] = np.sin(th)
        for i in range(1, self.njoints):
            t = int(self.jtype[i])
            if t == J_COMPOSITE:
                o = int(self.idx_q[i])
                for st, _, _ in self.composite[i]:
                    manifold_segments(st, o)
                    o += JOINT_NQ.get(st, 1)
            else:
                manifold_segments(t, int(self.idx_q[i]))
        return q


def builtin_model(name):
    """'panda7' | 'panda9' | 'talos32' | 'talos32_freeflyer' | 'talos44' (tables in loik_amd/csrc/models.c)"""
    L = lib()
    d = ModelDesc()
    lo, hi = _dp(), _dp()
    if L.loikb_builtin_model(name.encode(), C.byref(d), C.byref(lo), C.byref(hi)) != 0:
        raise KeyError(name)
    nj = d.njoints
    arr = lambda p, n, t: np.ctypeslib.as_array(p, shape=(n,)).astype(t).copy()
    names = [L.loikb_builtin_joint_name(name.encode(), i).decode() for i in range(nj)]
    return Model(arr(d.parents, nj, np.int32), arr(d.jtype, nj, np.int32), arr(d.axis, 3 * nj, np.float64),
                 arr(d.placement, 12 * nj, np.float64), names, arr(lo, d.nq, np.float64),
                 arr(hi, d.nq, np.float64), name=name)


def _ptr(a):
    """host numpy array, raw device pointer (int) or object with data_ptr() (torch tensor) -> (void*, is_device)"""
    if a is None:
        return None, False
    if isinstance(a, int):
        return C.c_void_p(a), True
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr()), bool(getattr(a, "is_cuda", False))
    return a.ctypes.data_as(C.c_void_p), False


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DeviceArray:
    """a float64 array resident in the HBM of `device` (hipMalloc + one hipMemcpy): what a caller who keeps its inputs on the GPU
    hands to SolveInit / the tailored Solve (LOIKB_IN_DEVICE) -- the bindings take anything with data_ptr() / is_cuda, e.g. a
    torch tensor; this is the same without importing torch (bench.py's C4 mode, the tests)"""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            h = C.CDLL("libamdhip64.so")
            h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            h.hipFree.argtypes = [C.c_void_p]
            cls._hip = h
        return cls._hip

    def __init__(self, a, device=0):
        a = _f64(a)
        self.shape, self.size, self.ndim, self.dtype, self.is_cuda, self.device = a.shape, a.size, a.ndim, a.dtype, True, int(device)
        h = self.hip()
        if h.hipSetDevice(self.device) != 0:
            raise RuntimeError("hipSetDevice(%d) failed" % self.device)
        p = C.c_void_p()
        if h.hipMalloc(C.byref(p), a.nbytes) != 0:
            raise MemoryError("hipMalloc of %d bytes failed" % a.nbytes)
        self._p = p
        if h.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) != 0:   # hipMemcpyHostToDevice
            raise RuntimeError("hipMemcpy failed")

    def data_ptr(self):
        return self._p.value

    def numel(self):
        return self.size

    def free(self):
        if self._p is not None and self._p.value:
            self.hip().hipFree(self._p)
        self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BatchedLoik:
    """`FirstOrderLoikOptimized` over a batch of independent instances, on one MI355X.

    Constructor keywords are the reference constructor's arguments (loik-loid-optimized.hpp:129-134) plus
    `batch`, `device`, `precision`, `flags`, `max_launch_iters`."""

    def __init__(self, model, batch, max_iter=200, tol_abs=1e-3, tol_rel=1e-3, tol_primal_inf=1e-2, tol_dual_inf=1e-2,
                 rho=1e-5, mu=1e-2, mu_equality_scale_factor=1e4, mu_update_strat=0, num_eq_c=1, eq_c_dim=6,
                 warm_start=False, tol_tail_solve=1e-1, verbose=False, logging=False, device=0, precision=F64, flags=0,
                 max_launch_iters=0, compact_min_instances=0, tail_max_instances=0, eq_c_capacity=0):
        self.L = lib()
        self.model = model
        self.batch = int(batch)
        self.nc = int(num_eq_c)
        self.opts = Options(max_iter, tol_abs, tol_rel, tol_primal_inf, tol_dual_inf, rho, mu, mu_equality_scale_factor,
                            mu_update_strat, num_eq_c, eq_c_dim, int(bool(warm_start)), tol_tail_solve,
                            int(bool(verbose)), int(bool(logging)), self.batch, device, precision, flags, max_launch_iters,
                            compact_min_instances, tail_max_instances, int(eq_c_capacity))
        self._desc = model.desc()
        h = C.c_void_p()
        _check(self.L.loikb_create(C.byref(self._desc), C.byref(self.opts), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.loikb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(self.L.loikb_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ------------------------------------------------------------------------------------------------------
    def SolveInit(self, q, H_ref, v_ref, active_task_constraint_ids, Ais, bis, lb, ub):
        keep, args = self._raw_args(q, H_ref, v_ref, active_task_constraint_ids, Ais, bis, lb, ub)
        _check(self.L.loikb_solve_init(*args))

    def _prep(self, a, what, per, shared_flag):
        """One per-instance input: (void*, is_device, flag).  The C-ABI takes bare pointers without lengths, so the sizes are
        validated HERE: a host array must hold exactly `per` elements (one value shared by the whole batch -> `shared_flag`)
        or `batch * per` (instance-major).  Device pointers / torch tensors are per-instance by contract; a torch tensor's
        numel is checked, a raw integer pointer cannot be."""
        B = self.batch
        if isinstance(a, int):
            return C.c_void_p(a), True, 0, None
        if hasattr(a, "data_ptr"):
            n = int(a.numel()) if hasattr(a, "numel") else None
            if n is not None and n != B * per:
                raise ValueError("%s: device tensor has %d elements, expected batch * %d = %d" % (what, n, per, B * per))
            if getattr(a, "is_cuda", False):
                return C.c_void_p(a.data_ptr()), True, 0, a
            a = a.numpy() if hasattr(a, "numpy") else np.asarray(a)
        a = _f64(a)
        if a.size == B * per and not (B == 1 and shared_flag in (A_SHARED, BOUNDS_SHARED)):
            return a.ctypes.data_as(C.c_void_p), False, 0, a         # (batch 1: q / b count as per-instance, A / box as shared)
        if a.size == per:
            return a.ctypes.data_as(C.c_void_p), False, shared_flag, a
        raise ValueError("%s has %d elements: expected %d (shared by the batch) or batch * %d = %d (instance-major)"
                         % (what, a.size, per, per, B * per))

    def _raw_args(self, q, H_ref, v_ref, c_ids, Ais, bis, lb, ub):
        """marshal SolveInit / Solve(q,H_ref,...) arguments; array lengths are checked here (see _prep), the reference's
        own validation (constraint count, bound dimension, duplicates) is the library's and comes back as its error codes"""
        B, nv, nq = self.batch, self.model.nv, self.model.nq
        H_ref = _f64(H_ref); v_ref = _f64(v_ref)
        if H_ref.size != 36 or v_ref.size != 6:
            raise ValueError("H_ref must be 6x6 and v_ref a 6-vector")
        H_ref = H_ref.reshape(36); v_ref = v_ref.reshape(6)
        c_ids = np.ascontiguousarray(c_ids, dtype=np.int32).reshape(-1)
        ncin = int(c_ids.size)
        # lb/ub dimension: the library compares it with model.nv and returns the reference's error (hpp:328-335)
        nbound = nv
        if not (isinstance(lb, int) or hasattr(lb, "data_ptr")):
            n = int(np.asarray(lb).size)
            if n != nv and n != nv * B:
                if int(np.asarray(ub).size) != n:
                    raise ValueError("lb and ub differ in size")
                nbound = n
        qp, qd, qf, k0 = self._prep(q, "q", nq, Q_SHARED)
        Ap, Ad, Af, k1 = self._prep(Ais, "Ais", 36 * ncin, A_SHARED) if ncin else (None, None, 0, None)
        bp, bd, bf, k2 = self._prep(bis, "bis", 6 * ncin, B_SHARED) if ncin else (None, None, 0, None)
        lp, ld, lf, k3 = self._prep(lb, "lb", nbound, BOUNDS_SHARED)
        up, ud, uf, k4 = self._prep(ub, "ub", nbound, BOUNDS_SHARED)
        if lf != uf:
            raise ValueError("lb and ub must both be shared ([nv]) or both be per instance ([batch][nv])")
        flags = qf | Af | bf | lf
        # device residency is one flag for all per-instance inputs: those that are not shared must agree
        devs = [d for d, f in ((qd, qf), (Ad, Af), (bd, bf), (ld, lf), (ud, uf)) if d is not None and not f]
        if any(devs):
            if not all(devs):
                raise ValueError("per-instance inputs must be all host or all device arrays")
            flags |= IN_DEVICE
        keep = [H_ref, v_ref, c_ids, k0, k1, k2, k3, k4]
        args = (self.h, qp, H_ref.ctypes.data_as(_dp), v_ref.ctypes.data_as(_dp), c_ids.ctypes.data_as(_ip), ncin, Ap,
                bp, lp, up, nbound, flags)
        return keep, args

    def Solve(self, *a):
        """Solve() | Solve(q,H_ref,v_ref,ids,Ais,bis,lb,ub) | Solve(q,c_id,Ai,bi); q=None in the tailored form uses
        the configurations resident on the device (outer loop: integrate(dt) then Solve(None, c_id, Ai, bi))"""
        if len(a) == 0:
            _check(self.L.loikb_solve(self.h))
        elif len(a) == 8:
            keep, args = self._raw_args(*a)
            _check(self.L.loikb_solve_full(*args))
        elif len(a) == 4:
            q, c_id, Ai, bi = a
            qp, qd, qf = None, None, 0
            if q is not None:  # None: the q resident on the device
                qp, qd, qf, k0 = self._prep(q, "q", self.model.nq, Q_SHARED)
            if int(c_id) < 0:  # no constraint update: solve on the set AddEqConstraint / RemoveEqConstraint left
                flags = qf | (IN_DEVICE if qd and not qf else 0)
                _check(self.L.loikb_solve_tailored(self.h, qp, -1, None, None, flags))
                return
            Ap, Ad, Af, k1 = self._prep(Ai, "Ai", 36, A_SHARED)
            bp, bd, bf, k2 = self._prep(bi, "bi", 6, B_SHARED)
            flags = qf | Af | bf
            devs = [d for d, f in ((qd, qf), (Ad, Af), (bd, bf)) if d is not None and not f]
            if any(devs):
                if not all(devs):
                    raise ValueError("per-instance inputs must be all host or all device arrays")
                flags |= IN_DEVICE
            _check(self.L.loikb_solve_tailored(self.h, qp, int(c_id), Ap, bp, flags))
        else:
            raise TypeError("Solve() takes 0, 4 or 8 arguments")

    # IkProblemFormulationOptimized's editing methods (ik-id-description-optimized.hpp; `problem_` is protected upstream)
    def UpdateReferences(self, H_refs, v_refs):
        """one weight [6][6] and one target [6] per joint of the model incl. the universe (hpp:103-121); in force for Solve() /
        the tailored Solve until the next SolveInit broadcasts one pair again"""
        H = _f64(np.asarray(H_refs, dtype=np.float64).reshape(-1, 36)); v = _f64(np.asarray(v_refs, dtype=np.float64).reshape(-1, 6))
        n = H.shape[0] if H.shape[0] == v.shape[0] else -1
        _check(self.L.loikb_update_references(self.h, H.ctypes.data_as(_dp), v.ctypes.data_as(_dp), n))

    def _edit_args(self, Ai, bi):
        Ap, Ad, Af, k1 = (None, None, 0, None) if Ai is None else self._prep(Ai, "Ai", 36, A_SHARED)
        bp, bd, bf, k2 = self._prep(bi, "bi", 6, B_SHARED)
        flags = Af | bf
        devs = [d for d, f in ((Ad, Af), (bd, bf)) if d is not None and not f]
        if any(devs):
            if not all(devs):
                raise ValueError("per-instance inputs must be all host or all device arrays")
            flags |= IN_DEVICE
        return Ap, bp, flags, (k1, k2)

    def UpdateEqConstraint(self, c_id, *a):
        """UpdateEqConstraint(c_id, Ai, bi) (hpp:178-218) | UpdateEqConstraint(c_id, bi) (hpp:224-238)"""
        Ai, bi = (None, a[0]) if len(a) == 1 else a
        Ap, bp, flags, keep = self._edit_args(Ai, bi)
        _check(self.L.loikb_update_eq_constraint(self.h, int(c_id), Ap, bp, flags))

    def AddEqConstraint(self, c_id, Ai, bi):
        """hpp:244-286; needs a free slot (constructor keyword eq_c_capacity)"""
        Ap, bp, flags, keep = self._edit_args(Ai, bi)
        _check(self.L.loikb_add_eq_constraint(self.h, int(c_id), Ap, bp, flags))

    def RemoveEqConstraint(self, c_id):
        """hpp:292-319; False when there was nothing to remove (upstream: a warning on stderr)"""
        rc = self.L.loikb_remove_eq_constraint(self.h, int(c_id))
        if rc < 0:
            _check(rc)
        return rc == 0

    def active_task_constraint_ids(self):
        n = self.L.loikb_num_eq_c(self.h)
        out = np.zeros(max(n, 1), dtype=np.int32)
        self.L.loikb_active_constraint_ids(self.h, out.ctypes.data_as(_ip), n)
        return [int(x) for x in out[:n]]

    # pass-level public methods of the reference (loik-loid-optimized.hpp:192-264): the debug path of loik_passes.hpp
    def _pass(self, k): _check(self.L.loikb_pass(self.h, k))
    def BeginIteration(self): self._pass(0)   # iter_++, UpdatePrev(), ResetInfNorms() (hpp:381-388)
    def FwdPass1(self): self._pass(1)
    def BwdPassOptimizedVisitor(self): self._pass(2)
    def FwdPass2OptimizedVisitor(self): self._pass(3)
    def BoxProj(self): self._pass(4)
    def DualUpdate(self): self._pass(5)
    def ComputeResiduals(self): self._pass(6)
    def CheckConvergence(self): self._pass(7)
    def CheckFeasibility(self): self._pass(8)
    def UpdateMu(self): self._pass(9)

    def plan(self):
        """which kernels this handle's solves use, and why (loikb_plan_string)"""
        return self.L.loikb_plan_string(self.h).decode()

    def synchronize(self):
        """hipDeviceSynchronize on the solver's device (bench.py's timing bracket)"""
        _check(self.L.loikb_synchronize(self.h))

    def integrate(self, dt):
        """outer loop on the device: q <- q (+) dt * z of the last solve, q stays resident in HBM"""
        _check(self.L.loikb_integrate(self.h, float(dt)))

    # ------------------------------------------------------------------------------------------------------
    def set_max_iter(self, n):
        _check(self.L.loikb_set_max_iter(self.h, int(n)))
        self.opts.max_iter = int(n)
    def set_rho(self, x): _check(self.L.loikb_set_rho(self.h, float(x)))
    def set_mu(self, x): _check(self.L.loikb_set_mu(self.h, float(x)))
    def set_tol(self, tol_abs, tol_rel): _check(self.L.loikb_set_tol(self.h, float(tol_abs), float(tol_rel)))
    def set_tol_primal_inf(self, x): _check(self.L.loikb_set_tol_primal_inf(self.h, float(x)))
    def set_tol_tail_solve(self, x): _check(self.L.loikb_set_tol_tail_solve(self.h, float(x)))
    def set_warm_start(self, w): _check(self.L.loikb_set_warm_start(self.h, int(bool(w))))

    def get(self, name, out=None):
        """one field for the whole batch as a numpy array (or into a device pointer / torch tensor `out`)"""
        fid = FIELD_ID[name]
        B, nb, nv, nc = self.batch, self.model.njoints - 1, self.model.nv, self.L.loikb_num_eq_c(self.h)
        # per DoF: [B][nv]; per link: [B][nb].  r / Dinv / UDinv are inter-sweep temporaries of the device's
        # elimination: per DoF, equal to upstream's per-joint values for 1-DoF joints only
        shapes = {"z": (B, nv), "nu": (B, nv), "w": (B, nv), "Stf_plus_w": (B, nv), "r": (B, nv), "Dinv": (B, nv),
                  "vis": (B, nb, 6), "fis": (B, nb, 6), "g": (B, nb, 6), "pis": (B, nb, 6), "UDinv": (B, nv, 6),
                  "His": (B, nb, 21), "liMi": (B, nb, 12), "yis": (B, nc, 6), "Aty": (B, nc, 6),
                  "q": (B, self.model.nq), "primal_residual_vec": (B, 6 * nb + nv),
                  "dual_residual_vec": (B, 6 * nb + nv)}
        is_int = name in ("iter", "converged", "primal_infeasible", "status", "mu_updates")
        if out is not None:
            p, dev = _ptr(out)
            _check(self.L.loikb_get(self.h, fid, p, OUT_DEVICE if dev else 0))
            return out
        arr = np.empty(shapes.get(name, (B,)), dtype=np.int32 if is_int else np.float64)
        _check(self.L.loikb_get(self.h, fid, arr.ctypes.data_as(C.c_void_p), 0))
        return arr

    RESULT_FIELDS = ("z", "nu", "w", "vis", "fis", "yis", "scalars")   # (bit k of loikb_get_results' mask: LOIKB_RES_*)
    NSCALARS, SCALAR_ITER, SCALAR_STATUS, SCALAR_MU_UPDATES = 33, 30, 31, 32   # (LOIKB_RES_NSCALARS ...: the layout of "scalars")

    def get_results(self, fields=RESULT_FIELDS[:6]):
        """the members of the reference's data object a solve leaves behind (z, nu, w, vis, fis, yis: loik-loid-data-optimized.hpp:118-178), any
        subset, in ONE call (loikb_get_results): {name: array}, the same values as get(name).  "scalars": [B][33] -- the 30 scalar getters'
        fields in FIELD_ID order from "primal_residual", then iter, the status bits, mu_updates (what get_iter() / get_convergence_status() / ...
        read), from the same gather"""
        B, nb, nv, nc = self.batch, self.model.njoints - 1, self.model.nv, self.L.loikb_num_eq_c(self.h)
        shapes = {"z": (B, nv), "nu": (B, nv), "w": (B, nv), "vis": (B, nb, 6), "fis": (B, nb, 6), "yis": (B, nc, 6), "scalars": (B, self.NSCALARS)}
        mask, out, ptrs = 0, {}, []
        for k, name in enumerate(self.RESULT_FIELDS):
            if name in fields:
                mask |= 1 << k
                out[name] = np.empty(shapes[name], dtype=np.float64)
                ptrs.append(out[name].ctypes.data_as(_dp))
            else:
                ptrs.append(None)
        unknown = [f for f in fields if f not in self.RESULT_FIELDS]
        if unknown:
            raise ValueError("get_results: not a result member: %s" % unknown)
        _check(self.L.loikb_get_results(self.h, mask, *ptrs))
        return out

    def His_full(self):
        """ik_id_data.His[i] as full symmetric 6x6 blocks: [B][nb][6][6]"""
        packed = self.get("His")
        B, nb = packed.shape[:2]
        full = np.zeros((B, nb, 6, 6))
        k = 0
        for i in range(6):
            for j in range(i, 6):
                full[:, :, i, j] = packed[:, :, k]
                full[:, :, j, i] = packed[:, :, k]
                k += 1
        return full

    SOLVER_INFO_LISTS = ["primal_residual_task_list", "primal_residual_slack_list", "primal_residual_list", "dual_residual_nu_list",
                         "dual_residual_v_list", "dual_residual_list", "mu_list", "mu_eq_list", "mu_ineq_list"]

    def solver_info(self):
        """LoikSolverInfo of the last solve (constructor keyword logging=True): {list name: [B][max_iter - 1]}, 'rows': [B]
        (max_iter as it was when that solve ran: the library says how many rows it holds)"""
        cap = max(int(self.L.loikb_solver_info_rows_cap(self.h)), 1)
        out = {}
        rows = np.zeros(self.batch, dtype=np.int32)
        for k, name in enumerate(self.SOLVER_INFO_LISTS):
            a = np.zeros((self.batch, cap))
            _check(self.L.loikb_get_solver_info(self.h, k, a.ctypes.data_as(_dp), cap, rows.ctypes.data_as(_ip)))
            out[name] = a
        out["rows"] = rows
        out["truncated_instances"] = int(self.L.loikb_solver_info_truncated(self.h))   # 0 but for a warm start whose mu left the decades
        return out

    def stats(self):
        st = Stats()
        _check(self.L.loikb_get_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    # reference getter names (task-solver-base.hpp:87-102), one value per instance
    def get_iter(self): return self.get("iter")
    def get_convergence_status(self): return self.get("converged").astype(bool)
    def get_primal_infeasibility_status(self): return self.get("primal_infeasible").astype(bool)
    def get_primal_residual(self): return self.get("primal_residual")
    def get_dual_residual(self): return self.get("dual_residual")
    def get_mu(self): return self.get("mu")
