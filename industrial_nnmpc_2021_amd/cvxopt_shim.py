"""cvxopt-shaped entry point: ``qp(P, q, G, h[, A, b]) -> {'x': (n,1), 'status': ...}``.

The reference reaches its solver only through ``cvx.solvers.qp(*array_to_matrix(...))``
and reads only ``solution['x']`` (lib/linearMPC.py:304-306, :503-506).  Installing
this module as ``sys.modules['cvxopt']`` therefore runs the
reference's own classes on the HIP path without touching them:

* inequality-only calls whose G is the box pattern blockdiag([I; -I]) (the
  regulator of a stable plant, :476-482) go to the batched GPU solver (B = 1);
* inequality-only calls whose G is tE (I + tK tB) (the re-parameterised regulator
  of an unstable plant, :476-479) are mapped to the equivalent box QP in input
  space (unit block lower triangular change of variables) and go to the same solver;
* calls with equality constraints (the target selector, :304-305) go to the
  small host solver;
* anything else raises.
"""
import collections
import types

import numpy as np

from . import host_qp

_CACHE_MAX = 4                                  # solver handles kept (each owns device workspace); least recently used goes
_solvers = collections.OrderedDict()


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


def _reparam_map(G, n):
    """(nu, Mg) if G == tE Mg with Mg unit block lower triangular (blocks nu x nu) -- the inequality matrix of the
    re-parameterised regulator, G = tE (I + tK tB) (reference :476-479) -- else None.  Row block k of G is [Mg_k; -Mg_k]."""
    if G.shape != (2 * n, n):
        return None
    # nu from the first row block: rows 0..nu-1 are e_j' (stage 0 of a unit block lower triangular map)
    nu = 0
    while nu < n and np.count_nonzero(G[nu]) == 1 and G[nu, nu] == 1.0:
        nu += 1
    # (the -I half of stage 0 follows the I half: the run above stops at row nu)
    if nu == 0 or n % nu or 2 * nu > G.shape[0]:
        return None
    N = n // nu
    rows = (np.arange(N)[:, None] * 2 * nu + np.arange(nu)[None, :]).ravel()
    Mg = G[rows]
    if not np.array_equal(G[rows + nu], -Mg):
        return None
    blocks = Mg.reshape(N, nu, N, nu)
    for k in range(N):
        if not np.array_equal(blocks[k, :, k, :], np.eye(nu)) or np.count_nonzero(blocks[k, :, k + 1:, :]):
            return None
    return nu, Mg


def _solver_for(tag, P, nu, make_P):
    key = (tag, P.shape, hash(P.tobytes()))
    if key in _solvers:
        _solvers.move_to_end(key)
        return _solvers[key]
    from .qp import BatchedBoxQP
    Pb, aux = make_P()
    # q is handed over directly: tq = I picks it out of x0 = q
    entry = (BatchedBoxQP(Pb, np.eye(Pb.shape[0]), nu, max_batch=128, seg_max=128, Kunc=None), aux)
    _solvers[key] = entry
    while len(_solvers) > _CACHE_MAX:
        _, (old, _) = _solvers.popitem(last=False)
        old.close()
    return entry


def qp(P, q, G, h, A=None, b=None, **kw):
    P, q, G, h = (np.asarray(P, float), np.asarray(q, float).reshape(-1),
                  np.asarray(G, float), np.asarray(h, float).reshape(-1))
    n = q.size
    if A is not None:
        x = host_qp.solve_small_qp(P, q, G, h, A, b)
        return {"x": x.reshape(-1, 1), "status": "optimal"}
    nu = _box_nu(G, n)
    if nu is not None:
        hb = h.reshape(-1, 2 * nu)
        if not np.allclose(hb, hb[0]):
            raise NotImplementedError("qp shim: bounds must be the same for every stage (tile of [uub; -ulb])")
        solver, _ = _solver_for("box", P, nu, lambda: (P, None))
        out = solver.solve_batch(q.reshape(1, -1), -hb[0, nu:].reshape(1, -1), hb[0, :nu].reshape(1, -1))
        x = out["u"].reshape(-1)
    else:
        rp = _reparam_map(G, n)
        if rp is None:
            raise NotImplementedError("qp shim: G must be blockdiag([I;-I]) or tE (I + tK tB) with a unit block lower "
                                      "triangular map (the regulator's two forms, lib/linearMPC.py:476-482)")
        # Re-parameterised regulator (u = Kx + v, :366-382): the rows of G v <= h are the box rows of w = Mg v shifted
        # per stage, lo_k <= w_k <= up_k with up = h_upper, lo = -h_lower (h = tile([uub; -ulb]) - tE tK tA x0, :484-493).
        # up_k - lo_k is the same for every stage, so wt_k = w_k + (up_0 - up_k) lies in the stage-independent box
        # [lo_0, up_0]: a box QP in wt with the shared Hessian Mg^-T P Mg^-1 -- same optimum, same active rows.
        nu, Mg = rp
        hb = h.reshape(-1, 2 * nu)
        up, lo = hb[:, :nu], -hb[:, nu:]
        if not np.allclose(up - lo, (up - lo)[0]):
            raise NotImplementedError("qp shim: dense-G rows do not come from one input box per stage")
        c = (up[0] - up).reshape(-1)                                   # wt = Mg v + c

        def make():
            import scipy.linalg
            Ps = np.tril(P) + np.tril(P, -1).T                         # cvxopt reads the lower triangle
            Y = scipy.linalg.solve_triangular(Mg, np.eye(n), lower=True, unit_diagonal=True)
            Pw = Y.T @ Ps @ Y
            return 0.5 * (Pw + Pw.T), Y
        solver, Y = _solver_for("reparam" + str(hash(Mg.tobytes())), P, nu, make)
        Pw = solver_P = None
        # objective in wt: 1/2 (wt - c)' Pw (wt - c) + q' Y (wt - c)  ->  linear term Y'q - Pw c
        Ps = np.tril(P) + np.tril(P, -1).T
        qw = Y.T @ q - Y.T @ (Ps @ (Y @ c))
        out = solver.solve_batch(qw.reshape(1, -1), lo[0].reshape(1, -1), up[0].reshape(1, -1))
        x = Y @ (out["u"].reshape(-1) - c)
    st = {0: "optimal", 1: "unknown", 2: "unknown"}[int(out["status"][0])]
    if out["status"][0] == 2:
        raise ArithmeticError("KKT factorisation failed")
    return {"x": x.reshape(-1, 1), "status": st, "iterations": int(out["ipm_iters"][0])}


solvers = types.SimpleNamespace(qp=qp, options={})
