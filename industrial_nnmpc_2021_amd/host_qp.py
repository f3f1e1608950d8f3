"""Small dense QP on the host (fp64):  min 1/2 x'Px + q'x  s.t. Gx <= h, Ax = b.

Only for the target-selector problem of the reference (lib/linearMPC.py:298-311:
n = Nx + Nu unknowns, 2 Nu box rows, Nx + Nz equalities, P only positive
SEMI-definite), which is not on the accelerated hot path.  Mehrotra primal-dual
interior point on the dense KKT system, finished by an active-set solve when the
active set it indicates passes the KKT sign checks.
"""
import numpy as np


def _kkt_solve(P, A, GtDG, r1, r2):
    n, pe = r1.size, r2.size
    K = np.zeros((n + pe, n + pe))
    K[:n, :n] = P + GtDG
    K[:n, n:] = A.T
    K[n:, :n] = A
    sol = np.linalg.lstsq(K, np.concatenate((r1, r2)), rcond=None)[0]
    return sol[:n], sol[n:]


def solve_small_qp(P, q, G, h, A, b, tol=1e-10, max_iter=100):
    P = np.asarray(P, float)
    P = np.tril(P) + np.tril(P, -1).T            # cvxopt reads the lower triangle
    q = np.asarray(q, float).reshape(-1)
    G = np.asarray(G, float)
    h = np.asarray(h, float).reshape(-1)
    A = np.asarray(A, float).reshape(-1, q.size)
    b = np.asarray(b, float).reshape(-1)
    n, m, pe = q.size, h.size, b.size
    x, y = _kkt_solve(P, A, G.T @ G, -q + G.T @ h, b)
    s = np.maximum(h - G @ x, 1.0)
    z = np.ones(m)
    for _ in range(max_iter):
        rx = P @ x + q + G.T @ z + A.T @ y
        ry = A @ x - b
        rz = s + G @ x - h
        gap = float(s @ z)
        if max(np.abs(rx).max(), np.abs(ry).max() if pe else 0.0, np.abs(rz).max(), gap) <= tol:
            break
        d = z / s
        GtDG = G.T @ (d[:, None] * G)
        mu, sigma, dsa, dza = gap / m, 0.0, 0.0, 0.0
        for i in (0, 1):
            rc = -s * z + sigma * mu - dsa * dza
            dx, dy = _kkt_solve(P, A, GtDG, -rx - G.T @ ((rc + z * rz) / s), -ry)
            ds = -rz - G @ dx
            dz = (rc - z * ds) / s
            t = max(0.0, np.max(-ds / s), np.max(-dz / z))
            if i == 0:
                a = 1.0 if t == 0 else min(1.0, 1.0 / t)
                dsa, dza = ds, dz
                sigma = min(1.0, max(0.0, 1.0 - a + float(ds @ dz) / gap * a * a)) ** 3
            else:
                a = 1.0 if t == 0 else min(1.0, 0.995 / t)
        xn, yn, sn, zn = x + a * dx, y + a * dy, s + a * ds, z + a * dz
        if not (np.all(np.isfinite(xn)) and np.all(sn > 0) and np.all(zn > 0)):
            break
        x, y, s, z = xn, yn, sn, zn
    # active-set finish: exact solve on the indicated set, accepted only if KKT-consistent
    act = z > s
    for _ in range(20):
        idx = np.flatnonzero(act)
        Ae, be = np.vstack((A, G[idx])), np.concatenate((b, h[idx]))
        xa, mult = _kkt_solve(P, Ae, 0.0, -q, be)
        lam = np.zeros(m)
        lam[idx] = mult[pe:]
        viol = G @ xa - h
        new = (act & (lam > 0.0)) | (~act & (viol > 1e-11 * (1.0 + np.abs(h))))
        if np.array_equal(new, act):
            return xa
        act = new
    return x
