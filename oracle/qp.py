"""fp64 CPU QP solvers for  min 1/2 x'Px + q'x  s.t.  Gx <= h   (TEST ORACLE).

The reference hands this problem to ``cvxopt.solvers.qp(P, q, G, h)``
(lib/linearMPC.py:503-504; also :304-305 with equalities).  cvxopt is a
third-party dependency that is neither vendored nor version-pinned by the
reference and is absent here, so:

* ``coneqp_l``  restates the published cvxopt ``coneqp`` algorithm specialised
  to the non-negative-orthant ('l') cone with cvxopt's defaults (maxiters=100,
  abstol=1e-7, reltol=1e-6, feastol=1e-7, Mehrotra predictor-corrector with
  exponent 3 and step fraction 0.99, Nesterov-Todd scaling = diag(sqrt(s/z)),
  ``chol2``-style reduced KKT  (P + G' diag(z/s) G) dx = rhs  formed by a dense
  SYRK and factored by LAPACK potrf).  It reproduces the reference's stopping
  behaviour and arithmetic cost and is what the CPU baseline times.
  PARITY UNPINNED at this seam (no cvxopt, no reference tests).
* ``solve_exact`` returns the exact optimum (unique: P > 0): a tight Mehrotra
  PDIP followed by an active-set polish and a KKT verification.  All parity
  tests of the HIP path are anchored on this.
"""
import numpy as np
import scipy.linalg as sla


def _sym_lower(P):
    # cvxopt reads only the lower triangle of P.
    L = np.tril(P)
    return L + np.tril(P, -1).T


def coneqp_l(P, q, G, h, maxiters=100, abstol=1e-7, reltol=1e-6, feastol=1e-7,
             info=None):
    """cvxopt ``coneqp`` restated for the 'l' cone, inequality-only form."""
    P = _sym_lower(np.asarray(P, float))
    q = np.asarray(q, float).reshape(-1)
    G = np.asarray(G, float)
    h = np.asarray(h, float).reshape(-1)
    m = h.size
    STEP, EXPON = 0.99, 3
    resx0 = max(1.0, np.linalg.norm(q))
    resz0 = max(1.0, np.linalg.norm(h))

    def kkt(d):
        # (P + G' diag(d) G) factor: dense SYRK + potrf, as chol2 does.
        K = P + G.T @ (d[:, None] * G)
        return sla.cho_factor(K, lower=True, check_finite=False)

    # initial point: W = I  ->  [P G'; G -I][x; z] = [-q; h],  s = -z
    F = kkt(np.ones(m))
    x = sla.cho_solve(F, -q + G.T @ h, check_finite=False)
    z = G @ x - h
    s = -z
    ts = np.max(-s)
    if ts >= -1e-8 * max(np.linalg.norm(s), 1.0):
        s = s + (1.0 + ts)
    tz = np.max(-z)
    if tz >= -1e-8 * max(np.linalg.norm(z), 1.0):
        z = z + (1.0 + tz)
    gap = float(s @ z)

    status, it = 'unknown', 0
    for it in range(maxiters + 1):
        Px = P @ x
        f0 = 0.5 * float(x @ Px) + float(q @ x)
        rx = Px + q + G.T @ z
        rz = s + G @ x - h
        resx, resz = np.linalg.norm(rx), np.linalg.norm(rz)
        pcost = f0
        dcost = f0 + float(z @ rz) - gap
        if pcost < 0.0:
            relgap = gap / -pcost
        elif dcost > 0.0:
            relgap = gap / dcost
        else:
            relgap = None
        pres, dres = resz / resz0, resx / resx0
        if (pres <= feastol and dres <= feastol and
                (gap <= abstol or (relgap is not None and relgap <= reltol))):
            status = 'optimal'
            break
        if it == maxiters:
            break
        d = z / s
        F = kkt(d)
        mu = gap / m
        sigma = 0.0
        dsa = dza = None
        for i in (0, 1):
            rc = -s * z + sigma * mu
            if i == 1:
                rc = rc - dsa * dza
            rhs = -rx - G.T @ ((rc + z * rz) / s)
            dx = sla.cho_solve(F, rhs, check_finite=False)
            ds = -rz - G @ dx
            dz = (rc - z * ds) / s
            t = max(0.0, np.max(-ds / s), np.max(-dz / z))
            if i == 0:
                step = 1.0 if t == 0.0 else min(1.0, 1.0 / t)
                dsa, dza = ds, dz
                sigma = min(1.0, max(0.0, 1.0 - step + float(ds @ dz) / gap * step ** 2)) ** EXPON
            else:
                step = 1.0 if t == 0.0 else min(1.0, STEP / t)
        x = x + step * dx
        s = s + step * ds
        z = z + step * dz
        gap = float(s @ z)
    if info is not None:
        info.update(status=status, iterations=it, gap=gap, s=s, z=z)
    return x


def _kkt_residuals(P, q, G, h, x, lam):
    r_stat = P @ x + q + G.T @ lam
    viol = np.maximum(G @ x - h, 0.0)
    return np.max(np.abs(r_stat)), np.max(viol), np.max(np.maximum(-lam, 0.0))


def solve_exact(P, q, G, h, tol=1e-10, maxiters=60, info=None):
    """Exact optimum: tight PDIP -> active set -> equality-constrained polish.

    Returns ``x``; ``info`` (optional dict) receives the active index set
    (rows of G that hold with equality and non-negative multiplier), the
    multipliers and KKT residuals.  Tie rule for weakly active constraints:
    a row is reported active iff its multiplier after the polish is > 0 or it
    was needed to keep the polished point feasible.
    """
    P = _sym_lower(np.asarray(P, float))
    q = np.asarray(q, float).reshape(-1)
    G = np.asarray(G, float)
    h = np.asarray(h, float).reshape(-1)
    n, m = q.size, h.size
    x = np.linalg.solve(P, -q)
    s = np.maximum(h - G @ x, 1.0)
    z = np.ones(m)
    for _ in range(maxiters):
        rx = P @ x + q + G.T @ z
        rz = s + G @ x - h
        gap = float(s @ z)
        if max(np.max(np.abs(rx)), np.max(np.abs(rz)), gap) <= tol:
            break
        d = z / s
        try:
            F = sla.cho_factor(P + G.T @ (d[:, None] * G), lower=True)
        except np.linalg.LinAlgError:
            # z / s spans ~1e+-10 near the end; with an ill-conditioned dense G (the re-parameterised regulator at the CDU size:
            # cond(G'G) ~ 1e5) the reduced KKT matrix then loses positivity in fp64.  The iterate is good enough to name the
            # active set: the polish below starts from it, and the KKT residuals reported in `info` judge the result.
            break
        mu, sigma, dsa, dza = gap / m, 0.0, None, None
        for i in (0, 1):
            rc = -s * z + sigma * mu - (dsa * dza if i else 0.0)
            dx = sla.cho_solve(F, -rx - G.T @ ((rc + z * rz) / s))
            ds = -rz - G @ dx
            dz = (rc - z * ds) / s
            t = max(0.0, np.max(-ds / s), np.max(-dz / z))
            if i == 0:
                a = 1.0 if t == 0 else min(1.0, 1.0 / t)
                dsa, dza = ds, dz
                sigma = min(1.0, max(0.0, 1.0 - a + float(ds @ dz) / gap * a * a)) ** 3
            else:
                a = 1.0 if t == 0 else min(1.0, 0.995 / t)
        x, s, z = x + a * dx, s + a * ds, z + a * dz
    # ---- active-set polish (primal-dual active-set iterations from the PDIP guess)
    act = z > s
    lam = np.zeros(m)
    FP = sla.cho_factor(P, lower=True)
    x0 = sla.cho_solve(FP, -q)
    for _ in range(50):
        idx = np.flatnonzero(act)
        lam = np.zeros(m)
        if idx.size:
            Ga = G[idx]
            PiGt = sla.cho_solve(FP, Ga.T)
            la = np.linalg.solve(Ga @ PiGt, Ga @ x0 - h[idx])
            xn = x0 - PiGt @ la
            lam[idx] = la
        else:
            xn = x0
        slack = h - G @ xn
        new_act = (act & (lam > 0.0)) | (~act & (slack < -1e-13 * (1 + np.abs(h))))
        x = xn
        if np.array_equal(new_act, act):
            break
        act = new_act
    if info is not None:
        r = _kkt_residuals(P, q, G, h, x, lam)
        info.update(active=np.flatnonzero(act), lam=lam, kkt=r)
    return x


def box_as_Gh(n_stage, N, lb, ub):
    """G = blockdiag([I; -I]) and h = tile([ub; -lb]) in the reference's row
    order (lib/linearMPC.py:459-460, :487-488)."""
    nu = n_stage
    E = np.vstack((np.eye(nu), -np.eye(nu)))
    G = sla.block_diag(*([E] * N))
    h = np.tile(np.concatenate((np.ravel(ub), -np.ravel(lb))), N)
    return G, h


def solve_exact_box(P, q, lb, ub, tol=1e-11, maxiters=80, info=None):
    """Exact optimum of  min 1/2 x'Px + q'x,  lb <= x <= ub  (per-variable bounds).

    Same problem as ``solve_exact(P, q, *box_as_Gh(...))`` but exploits G = [I; -I]
    (the KKT matrix is P + diag(d)), so it is usable at the CDU size (n = 4480).
    ``info['active']`` lists active rows in the reference's G row order when
    ``info['nu']`` is given (rows k*2nu + c upper, k*2nu + nu + c lower).
    """
    P = _sym_lower(np.asarray(P, float))
    q = np.asarray(q, float).reshape(-1)
    lb = np.asarray(lb, float).reshape(-1)
    ub = np.asarray(ub, float).reshape(-1)
    n = q.size
    FP = sla.cho_factor(P, lower=True)
    x = np.clip(sla.cho_solve(FP, -q), lb + 0.05 * (ub - lb), ub - 0.05 * (ub - lb))
    g = P @ x + q
    mu0 = 0.1 * np.mean(np.abs(g)) + 1e-3
    su, sl = ub - x, x - lb
    zu, zl = np.maximum(-g, 0) + mu0 / su, np.maximum(g, 0) + mu0 / sl
    scale = max(1.0, np.abs(q).max())
    for _ in range(maxiters):
        su, sl = ub - x, x - lb
        g = P @ x + q
        rd = g + zu - zl
        gap = float(su @ zu + sl @ zl)
        mu = gap / (2 * n)
        if np.abs(rd).max() <= tol * scale and mu <= tol * scale:
            break
        F = sla.cho_factor(P + np.diag(zu / su + zl / sl), lower=True)
        smu, dua = 0.0, None
        for i in (0, 1):
            if i == 0:
                rhs = -g
            else:
                rhs = -rd + zu - smu / su - dua * dzu / su - zl + smu / sl - dua * dzl / sl
            du = sla.cho_solve(F, rhs)
            if i == 0:
                dzu, dzl = -zu + zu * du / su, -zl - zl * du / sl
                dua = du
                nzu, nzl = dzu, dzl
            else:
                rcu = -su * zu + smu + dua * dzu
                rcl = -sl * zl + smu - dua * dzl
                nzu, nzl = (rcu + zu * du) / su, (rcl - zl * du) / sl
            t = max(0.0, np.max(du / su), np.max(-du / sl), np.max(-nzu / zu), np.max(-nzl / zl))
            if i == 0:
                a = 1.0 if t == 0 else min(1.0, 1.0 / t)
                ga = float((su - a * du) @ (zu + a * dzu) + (sl + a * du) @ (zl + a * dzl))
                smu = min(1.0, max(0.0, ga / gap)) ** 3 * mu
            else:
                a = 1.0 if t == 0 else min(1.0, 0.995 / t)
        x, zu, zl = x + a * du, np.maximum(zu + a * nzu, 1e-300), np.maximum(zl + a * nzl, 1e-300)
    # active-set polish (primal-dual active-set steps on the exact free-block solve)
    au, al = zu > su, zl > sl
    both = au & al
    au, al = au & ~(both & (zl * su > zu * sl)), al & ~(both & (zl * su <= zu * sl))
    for _ in range(100):
        act = au | al
        xb = np.where(au, ub, np.where(al, lb, 0.0))
        fr = ~act
        xn = xb.copy()
        if fr.any():
            rhs = -(q[fr] + P[np.ix_(fr, act)] @ xb[act])
            xn[fr] = sla.cho_solve(sla.cho_factor(P[np.ix_(fr, fr)], lower=True), rhs)
        g = P @ xn + q
        vu, vl = fr & (xn > ub + 1e-12 * (1 + np.abs(ub))), fr & (xn < lb - 1e-12 * (1 + np.abs(lb)))
        du_, dl_ = au & (g >= 0), al & (g <= 0)
        x = xn
        if not (vu.any() or vl.any() or du_.any() or dl_.any()):
            break
        au, al = (au & ~du_) | vu, (al & ~dl_) | vl
    if info is not None:
        g = P @ x + q
        lam_u, lam_l = np.where(au, -g, 0.0), np.where(al, g, 0.0)
        info.update(au=au, al=al, kkt=(np.abs(g + lam_u - lam_l).max(), max(np.max(x - ub), np.max(lb - x), 0.0),
                                      max(np.max(-lam_u), np.max(-lam_l), 0.0)))
        nu = info.get("nu")
        if nu:
            k, c = np.arange(n) // nu, np.arange(n) % nu
            rows = np.zeros(2 * n, bool)
            rows[k * 2 * nu + c] = au
            rows[k * 2 * nu + nu + c] = al
            info["active"] = np.flatnonzero(rows)
    return x


def solve_exact_eq(P, q, G, h, A, b, tol=1e-11, maxiters=100, info=None):
    """Exact optimum of  min 1/2 x'Px + q'x  s.t. Gx <= h, Ax = b   (P >= 0 allowed).

    The target-selector QP of the reference (lib/linearMPC.py:298-311) has this
    form (cvxopt.solvers.qp(P, q, G, h, A, b), :304-305).  Dense KKT solves;
    meant for the small fixtures / the chain-driver golden only.
    """
    P = _sym_lower(np.asarray(P, float))
    q = np.asarray(q, float).reshape(-1)
    G = np.asarray(G, float)
    h = np.asarray(h, float).reshape(-1)
    A = np.asarray(A, float)
    b = np.asarray(b, float).reshape(-1)
    n, m, pe = q.size, h.size, b.size

    def kkt_solve(d, r1, r2):
        K = np.block([[P + G.T @ (d[:, None] * G), A.T], [A, np.zeros((pe, pe))]])
        sol = np.linalg.solve(K, np.concatenate((r1, r2)))
        return sol[:n], sol[n:]

    x, y = kkt_solve(np.ones(m), -q + G.T @ h, b)
    s = np.maximum(h - G @ x, 1.0)
    z = np.ones(m)
    for _ in range(maxiters):
        rx = P @ x + q + G.T @ z + A.T @ y
        ry = A @ x - b
        rz = s + G @ x - h
        gap = float(s @ z)
        if max(np.abs(rx).max(), np.abs(ry).max() if pe else 0.0, np.abs(rz).max(), gap) <= tol:
            break
        d = z / s
        mu, sigma, dsa, dza = gap / m, 0.0, None, None
        for i in (0, 1):
            rc = -s * z + sigma * mu - (dsa * dza if i else 0.0)
            dx, dy = kkt_solve(d, -rx - G.T @ ((rc + z * rz) / s), -ry)
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
            break                     # fp64 floor reached; the polish below finishes the job
        x, y, s, z = xn, yn, sn, zn
    # polish on the active inequality rows
    act = z > s
    for _ in range(50):
        idx = np.flatnonzero(act)
        Ae = np.vstack((A, G[idx]))
        be = np.concatenate((b, h[idx]))
        K = np.block([[P, Ae.T], [Ae, np.zeros((Ae.shape[0],) * 2)]])
        sol = np.linalg.lstsq(K, np.concatenate((-q, be)), rcond=None)[0]
        xn = sol[:n]
        lam = np.zeros(m)
        lam[idx] = sol[n + pe:]
        slack = h - G @ xn
        new_act = (act & (lam > 0.0)) | (~act & (slack < -1e-12 * (1 + np.abs(h))))
        x = xn
        if np.array_equal(new_act, act):
            break
        act = new_act
    if info is not None:
        info.update(active=np.flatnonzero(act))
    return x
