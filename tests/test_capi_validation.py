"""CPU-only: the ctypes binding validates array lengths before handing bare pointers to the C-ABI (which has no way to)."""
import numpy as np
import pytest

import loik_amd
from loik_amd import capi


def _binding(model, B, nc=1):
    s = object.__new__(capi.BatchedLoik)  # no device needed: only the marshalling is exercised
    s.model, s.batch, s.nc, s.h = model, B, nc, None
    return s


def test_lengths_are_checked(talos):
    B, nv, nq = 5, talos.nv, talos.nq
    s = _binding(talos, B)
    ok = dict(q=np.zeros((B, nq)), H_ref=np.eye(6), v_ref=np.zeros(6), c_ids=[3], Ais=np.eye(6)[None], bis=np.zeros((B, 1, 6)),
              lb=-np.ones(nv), ub=np.ones(nv))
    order = ["q", "H_ref", "v_ref", "c_ids", "Ais", "bis", "lb", "ub"]
    keep, args = s._raw_args(*[ok[k] for k in order])
    assert args[-1] == capi.A_SHARED | capi.BOUNDS_SHARED and args[-2] == nv
    for name, bad in [("q", np.zeros((B - 1, nq))), ("q", np.zeros((B, nq + 1))), ("bis", np.zeros((B + 2, 1, 6))),
                      ("Ais", np.zeros((B - 1, 1, 6, 6))), ("lb", np.zeros((B - 1, nv))), ("H_ref", np.eye(5))]:
        with pytest.raises(ValueError):
            s._raw_args(*[bad if k == name else ok[k] for k in order])
    # lb per instance with ub shared used to set BOUNDS_SHARED for both and read row 0 of lb only
    with pytest.raises(ValueError):
        s._raw_args(*[np.tile(ok["lb"], (B, 1)) if k == "lb" else ok[k] for k in order])
    keep, args = s._raw_args(*[np.tile(ok[k], (B, 1)) if k in ("lb", "ub") else ok[k] for k in order])
    assert not (args[-1] & capi.BOUNDS_SHARED)
    # a wrong bound DIMENSION goes through to the library, which answers with the reference's error (hpp:328-335)
    keep, args = s._raw_args(*[np.ones(nv + 1) * (1 if k == "ub" else -1) if k in ("lb", "ub") else ok[k] for k in order])
    assert args[-2] == nv + 1
    # shared q / b for the whole batch
    keep, args = s._raw_args(*[np.zeros(nq) if k == "q" else np.zeros((1, 6)) if k == "bis" else ok[k] for k in order])
    assert args[-1] & capi.Q_SHARED and args[-1] & capi.B_SHARED


def test_batch_one_conventions(talos):
    s = _binding(talos, 1)
    keep, args = s._raw_args(np.zeros(talos.nq), np.eye(6), np.zeros(6), [3], np.eye(6)[None], np.zeros((1, 6)),
                             -np.ones(talos.nv), np.ones(talos.nv))
    assert args[-1] == capi.A_SHARED | capi.BOUNDS_SHARED  # q and b of the one instance are "per instance"
