"""cvxopt-shaped entry point: ``qp(P, q, G, h[, A, b]) -> {'x': (n,1), 'status': ...}``.

The reference reaches its solver only through ``cvx.solvers.qp(*array_to_matrix(...))``
and reads only ``solution['x']`` (lib/linearMPC.py:304-306, :503-506).  Installing
this module as ``sys.modules['cvxopt']`` therefore runs the
reference's own classes on the HIP path without touching them:

* inequality-only calls whose G is the box pattern blockdiag([I; -I]) (the
  regulator of a stable plant, :476-482) go to the batched GPU solver (B = 1);
* calls with equality constraints (the target selector, :304-305) go to the
  small host solver;
* anything else (dense G of the re-parameterised regulator) raises.
"""
import types

import numpy as np

from . import host_qp

_solvers = {}


def matrix(a, *args, **kw):
    """cvxopt.matrix stand-in: the reference only wraps numpy arrays (:15-20)."""
    return np.asarray(a, dtype=float)


def _box_nu(G, n):
    """nu if G == blockdiag([I_nu; -I_nu]) for some nu dividing n, else None."""
    if G.shape != (2 * n, n):
        return None
    first = np.flatnonzero(G[:, 0])
    if first.size != 2 or first[0] != 0:
        return None
    nu = int(first[1])
    if nu <= 0 or n % nu:
        return None
    E = np.vstack((np.eye(nu), -np.eye(nu)))
    N = n // nu
    for k in range(N):
        blk = G[2 * nu * k:2 * nu * (k + 1)]
        if not (np.array_equal(blk[:, nu * k:nu * (k + 1)], E) and
                np.count_nonzero(blk) == 2 * nu):
            return None
    return nu


def qp(P, q, G, h, A=None, b=None, **kw):
    P, q, G, h = (np.asarray(P, float), np.asarray(q, float).reshape(-1),
                  np.asarray(G, float), np.asarray(h, float).reshape(-1))
    n = q.size
    if A is not None:
        x = host_qp.solve_small_qp(P, q, G, h, A, b)
        return {"x": x.reshape(-1, 1), "status": "optimal"}
    nu = _box_nu(G, n)
    if nu is None:
        raise NotImplementedError("qp shim: only box G = blockdiag([I;-I]) is accelerated")
    hb = h.reshape(-1, 2 * nu)
    if not np.allclose(hb, hb[0]):
        raise NotImplementedError("qp shim: bounds must be the same for every stage (tile of [uub; -ulb])")
    key = (P.shape, hash(P.tobytes()))
    if key not in _solvers:
        from .qp import BatchedBoxQP
        # q is handed over directly: tq = I picks it out of x0 = q
        _solvers[key] = BatchedBoxQP(P, np.eye(n), nu, max_batch=128, Kunc=None)
    out = _solvers[key].solve_batch(q.reshape(1, -1), -hb[0, nu:].reshape(1, -1), hb[0, :nu].reshape(1, -1))
    st = {0: "optimal", 1: "unknown", 2: "unknown"}[int(out["status"][0])]
    if out["status"][0] == 2:
        raise ArithmeticError("KKT factorisation failed")
    return {"x": out["u"].reshape(-1, 1), "status": st, "iterations": int(out["ipm_iters"][0])}


solvers = types.SimpleNamespace(qp=qp, options={})
