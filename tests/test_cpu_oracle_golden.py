"""CPU suite: oracle and host logic against the golden vectors produced by the reference
itself (tests/golden/make_golden.py) -- no GPU needed."""
import glob
import os

import numpy as np
import pytest
from scipy.optimize import lsq_linear

from oracle import condense as oc, qp as oqp, nn as onn
from industrial_nnmpc_2021_amd import condense as pc, linearMPC as lm, controller_evaluation as ce

CASES = ["stable_s0", "stable_s1", "unstable_s0", "unstable_s1"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("case", CASES)
def test_oracle_condensing_matches_reference(golden_dir, case):
    g = _load(golden_dir, f"regulator_{case}.npz")
    reg = oc.setup_regulator(g["A"], g["B"], g["Q"], g["R"], g["S"], int(g["N"]), g["ulb"], g["uub"])
    assert reg.reparameterize == bool(g["reparameterize"])
    for name, mine in (("P", reg.P), ("tq", reg.tq), ("G", reg.G), ("tA", reg.tA), ("tB", reg.tB),
                       ("Krep", reg.Krep), ("Pf", reg.Pf)):
        ref = g[name]
        assert np.abs(mine - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), name
    Aa, Ba, Qa, Ra, Ma = oc.augment_for_regulator(g["A"], g["B"], g["Q"], g["R"], g["S"])
    for name, mine in (("Aaug", Aa), ("Baug", Ba), ("Qaug", Qa), ("Raug", Ra), ("Maug", Ma)):
        assert np.array_equal(mine, g[name]), name
    for b in range(g["x0"].shape[0]):
        us = g["us"][b]
        reg.ulb, reg.uub = g["ulb"] - us, g["uub"] - us
        assert np.abs(reg.h(g["x0"][b]) - g["h"][b]).max() < 1e-10
        assert np.abs(reg.tq @ g["x0"][b] - g["q"][b]).max() <= 1e-10 * max(1, np.abs(g["q"][b]).max())


@pytest.mark.parametrize("case", CASES)
def test_product_condensing_matches_reference(golden_dir, case):
    """condense.py (recursion, no dense stacks) and the DenseQPRegulator mirror."""
    g = _load(golden_dir, f"regulator_{case}.npz")
    reg = lm.LinearMPCController.setup_regulator(g["A"], g["B"], g["Q"], g["R"], g["S"], int(g["N"]),
                                                 g["ulb"], g["uub"])
    assert reg.reparameterize == bool(g["reparameterize"])
    scale = max(1.0, np.abs(g["P"]).max())
    assert np.abs(reg.P - g["P"]).max() <= 1e-10 * scale
    assert np.abs(reg.tq - g["tq"]).max() <= 1e-10 * max(1.0, np.abs(g["tq"]).max()) + 1e-12
    assert np.abs(reg.G - g["G"]).max() <= 1e-10
    assert np.abs(reg.Krep - g["Krep"]).max() <= 1e-9 and np.abs(reg.Pf - g["Pf"]).max() <= 1e-8 * scale
    for b in range(2):
        us = g["us"][b]
        reg.ulb, reg.uub = g["ulb"] - us, g["uub"] - us
        assert np.abs(reg._get_h(g["x0"][b]) - g["h"][b]).max() < 1e-10


@pytest.mark.parametrize("case", CASES)
def test_oracle_exact_qp_golden_kkt_and_bvls(golden_dir, case):
    g = _load(golden_dir, f"regulator_{case}.npz")
    e = _load(golden_dir, f"qp_exact_{case}.npz")
    P = np.tril(g["P"]) + np.tril(g["P"], -1).T
    G = g["G"]
    for b in range(g["x0"].shape[0]):
        q, h = g["q"][b].ravel(), g["h"][b].ravel()
        info = {}
        v = oqp.solve_exact(P, q, G, h, info=info)
        assert np.abs(v - e["v"][b]).max() <= 1e-9 * max(1, np.abs(v).max())
        act = np.zeros(h.size, bool)
        act[info["active"]] = True
        assert np.array_equal(act, e["active"][b])
        # KKT: stationarity, feasibility, multiplier sign, complementarity
        lam = info["lam"]
        assert np.abs(P @ v + q + G.T @ lam).max() <= 1e-8 * max(1, np.abs(q).max())
        assert (G @ v - h).max() <= 1e-9 and lam.min() >= 0 and np.abs(lam * (G @ v - h)).max() < 1e-7
        # cvxopt-style PDIP (default tolerances) lands near the same point
        vc = oqp.coneqp_l(P, q, G, h)
        assert np.abs(vc - v).max() <= 2e-2 * max(1, np.abs(v).max())
        if not bool(g["reparameterize"]):
            nu, N = g["B"].shape[1], int(g["N"])
            lbv, ubv = np.tile(-h[nu:2 * nu], N), np.tile(h[:nu], N)
            L = np.linalg.cholesky(P)
            r = lsq_linear(L.T, -np.linalg.solve(L, q), bounds=(lbv, ubv), method="bvls", tol=1e-14, max_iter=5000)
            assert np.abs(r.x - v).max() <= 1e-7 * max(1, np.abs(v).max())
            vb = oqp.solve_exact_box(P, q, lbv, ubv, info=(ib := {"nu": nu}))
            assert np.abs(vb - v).max() <= 1e-9 * max(1, np.abs(v).max())
            rows = np.zeros(h.size, bool)
            rows[ib["active"]] = True
            assert np.array_equal(rows, act)
        # the reference's own post-processing (un-reparameterisation + us) reproduced by the oracle
        reg = oc.setup_regulator(g["A"], g["B"], g["Q"], g["R"], g["S"], int(g["N"]), g["ulb"], g["uub"])
        useq = oc.control_sequence(reg, g["x"][b], g["uprev"][b], g["xs"][b], g["us"][b], g["ulb"], g["uub"],
                                   lambda P_, q_, G_, h_: oqp.solve_exact(P_, q_, G_, h_))
        assert np.abs(useq - e["useq"][b]).max() <= 1e-8 * max(1, np.abs(useq).max())


@pytest.mark.parametrize("name", ["with_uprev", "without_uprev"])
def test_oracle_nn_matches_reference(golden_dir, name):
    g = _load(golden_dir, f"nn_{name}.npz")
    W = [g[f"W{i}"] for i in range(int(g["nW"]))]
    u = onn.control_input(W, g["x"], g["uprev"], g["xs"], g["us"], g["xscale"], g["ulb"], g["uub"], bool(g["withuprev"]))
    assert np.abs(u - g["u"]).max() < 1e-12
    # steady-state property of the structured architecture: x = xs, uprev = us  =>  u = clip(us)
    assert np.array_equal(u[0], np.clip(g["us"][0], g["ulb"].ravel(), g["uub"].ravel()))


def test_prbs_sampler_matches_reference(golden_dir):
    g = _load(golden_dir, "prbs.npz")
    sig = ce.sample_prbs_like(num_change=6, num_steps=60, lb=np.array([[-1.], [0.]]), ub=np.array([[1.], [2.]]),
                              mean_change=10, sigma_change=1, seed=3)
    assert np.array_equal(sig, g["signal"])


def test_target_selector_and_split_match_reference_chain(golden_dir):
    g = _load(golden_dir, "chain.npz")
    nu = g["B"].shape[1]
    ts = lm.TargetSelector(A=g["A"], B=g["B"], C=g["C"], H=g["H"], Bd=g["Bd"], Cd=g["Cd"], usp=np.zeros((nu, 1)),
                           Rs=g["Rs"], Qs=g["Qs"], ulb=g["ulb"], uub=g["uub"], backend="host")   # (the hip backend: test_chain_target_gpu.py)
    T = g["x_0"].shape[0]
    for task in range(2):
        sp = g["setpoints"][task * T:(task + 1) * T]
        ds = g["disturbances"][task * T:(task + 1) * T]
        assert np.array_equal(sp, g[f"split_setpoints_{task}"])
        for t in range(T):
            xs, us = ts.solve(sp[t][:, None], ds[t][:, None])
            assert np.abs(xs[:, 0] - g[f"xs_{task}"][t]).max() < 1e-8
            assert np.abs(us[:, 0] - g[f"us_{task}"][t]).max() < 1e-8


def test_chain_recurrence_with_oracle_solver(golden_dir):
    """x+ = A x + B u + Bd d with u from the exact oracle reproduces the reference chain."""
    g = _load(golden_dir, "chain.npz")
    Nx, nu = g["B"].shape
    reg = oc.setup_regulator(g["A"], g["B"], g["Q"], g["R"], g["S"], int(g["N"]), g["ulb"], g["uub"])
    T = g["x_0"].shape[0]
    x, uprev = np.zeros((Nx, 1)), np.zeros((nu, 1))
    for t in range(T):
        assert np.abs(x[:, 0] - g["x_0"][t]).max() < 1e-7
        xs, us = g["xs_0"][t][:, None], g["us_0"][t][:, None]
        useq = oc.control_sequence(reg, x, uprev, xs, us, g["ulb"], g["uub"],
                                   lambda P_, q_, G_, h_: oqp.solve_exact(P_, q_, G_, h_))
        u = useq[:nu]
        assert np.abs(u[:, 0] - g["u_0"][t]).max() < 1e-7
        x = g["A"] @ x + g["B"] @ u + g["Bd"] @ g["disturbances"][t][:, None]
        uprev = u


def test_cdu_size_condensing_properties():
    """At the full CDU size the dense reference construction is infeasible (12.8 GB tQ);
    check size-independent properties of the recursion instead: symmetry, P >= R, and
    agreement of random quadratic-form evaluations with a direct horizon simulation."""
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd.linearMPC_build import augmented_matrices_for_regulator, dlqr
    pl = synthetic.plant("cdu", 0)
    Aa, Ba, Qa, Ra, Ma = augmented_matrices_for_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"])
    _, Pf = dlqr(Aa, Ba, Qa, Ra, Ma)
    N = 24                                                 # shorter horizon keeps the CPU suite fast
    P, tq = pc.condense(Aa, Ba, Qa, Ra, Ma, Pf, N)
    assert np.abs(P - P.T).max() < 1e-9 * np.abs(P).max()
    rng = np.random.default_rng(0)
    nu = Ba.shape[1]
    for _ in range(3):
        x0 = rng.standard_normal((Aa.shape[0], 1)); u = rng.standard_normal((N * nu, 1))
        x, V = x0, 0.0
        for k in range(N):
            uk = u[k * nu:(k + 1) * nu]
            V += 0.5 * (x.T @ Qa @ x + uk.T @ Ra @ uk + 2 * x.T @ Ma @ uk).item()
            x = Aa @ x + Ba @ uk
        V += 0.5 * (x.T @ Pf @ x).item()
        Vc = 0.5 * (u.T @ P @ u).item() + ((tq @ x0).T @ u).item()
        # constant term c(x0) = V(x0, 0): evaluate it by simulating u = 0
        x, c = x0, 0.0
        for k in range(N):
            c += 0.5 * (x.T @ Qa @ x).item()
            x = Aa @ x
        c += 0.5 * (x.T @ Pf @ x).item()
        assert abs((V - c) - Vc) <= 1e-9 * max(1.0, abs(V))
