#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory (run in the BUILD container only).

Imports the reference (read-only at /root/reference) with its un-installable
third-party modules stubbed (cvxopt, h5py; np.int restored), runs the
reference's own pure-numpy code on seeded inputs and stores inputs + outputs as
small .npz files.  Only data is stored -- no reference source travels.

  regulator_<case>.npz   a1-a7 of SURVEY section 8: reference-built P, tq, G, tA, tK,
                         Krep, Pf, reparameterize flag, h(x0), tq@x0 for seeded x0
  qp_exact_<case>.npz    exact optimum / active set of those QPs (fp64 oracle; the
                         test cross-checks them with scipy BVLS and KKT residuals)
  chain.npz              reference simulate_offline run with the exact oracle solver
                         injected at the cvxopt.solvers.qp seam
  nn_<case>.npz          reference NeuralNetworkController._get_control_input outputs
  prbs.npz               reference sample_prbs_like outputs
  target.npz             reference TargetSelector.solve (oracle at the qp seam) on seeded (ysp, dhat) pairs, each
                         cross-checked at generation time against scipy's trust-constr on the full-space problem
  closed_loop.npz        reference online_simulation with LinearMPCController and NeuralNetworkController on a
                         seeded plant (np.random.seed for the measurement noise): y, u, x, xhat, average stage costs
"""
import io
import os
import sys
import types
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import qp as oqp  # noqa: E402

REF = "/root/reference/lib"


def import_reference():
    cvx = types.ModuleType("cvxopt")
    cvx.matrix = lambda a: np.asarray(a, dtype=float)

    def qp(P, q, G, h, A=None, b=None):
        if A is None:
            x = oqp.solve_exact(P, q, G, h)
        else:
            x = oqp.solve_exact_eq(P, q, G, h, A, b)
            # a fixture must not contain an infeasible target problem (cvxopt would return status 'unknown' there)
            if np.abs(np.asarray(A) @ x - np.asarray(b).ravel()).max() > 1e-8 or (np.asarray(G) @ x - np.asarray(h).ravel()).max() > 1e-8:
                raise ArithmeticError("make_golden: infeasible equality-constrained QP in a fixture run")
        return {"x": np.asarray(x).reshape(-1, 1), "status": "optimal"}

    cvx.solvers = types.SimpleNamespace(qp=qp, options={})
    sys.modules["cvxopt"] = cvx
    sys.modules["h5py"] = types.ModuleType("h5py")
    mpl = types.ModuleType("matplotlib"); plt = types.ModuleType("matplotlib.pyplot")
    plt.rcParams = {}
    plt.rcParams = type("R", (dict,), {"update": lambda self, *a, **k: None})()
    sys.modules.setdefault("matplotlib", mpl); sys.modules.setdefault("matplotlib.pyplot", plt)
    be = types.ModuleType("matplotlib.backends"); bp = types.ModuleType("matplotlib.backends.backend_pdf")
    bp.PdfPages = object
    sys.modules.setdefault("matplotlib.backends", be); sys.modules.setdefault("matplotlib.backends.backend_pdf", bp)
    if not hasattr(np, "int"):
        np.int = int
    sys.path.insert(0, REF)
    import linearMPC as ref
    import controller_evaluation as ce
    return ref, ce


def plant(rng, Nx, Nu, Ny, rho):
    W = rng.standard_normal((Nx, Nx)) / np.sqrt(Nx)
    A = rho * W / np.max(np.abs(np.linalg.eigvals(W)))
    B = rng.standard_normal((Nx, Nu)) / np.sqrt(Nx)
    C = rng.standard_normal((Ny, Nx)) / np.sqrt(Nx)
    return A, B, C


CASES = {  # name: (Nx, Nu, Ny, N, rho, qweight, sweight)
    "stable_s0": (7, 3, 4, 9, 0.97, 2.0, 0.0),       # CDU-like tuning
    "stable_s1": (5, 2, 5, 12, 0.9, 1e3, 0.1),       # CSTRs-like tuning (rate penalty)
    "unstable_s0": (6, 2, 3, 8, 1.03, 2.0, 0.0),     # re-parameterised branch
    "unstable_s1": (4, 2, 4, 7, 1.05, 10.0, 0.1),
}


def make_target(ref):
    """TargetSelector.solve (lib/linearMPC.py:298-311) on seeded pairs; only pairs whose optimum an INDEPENDENT solver
    (scipy trust-constr on the full-space problem, equality + bound constraints) confirms are kept."""
    from scipy.optimize import minimize, LinearConstraint, Bounds
    rng = np.random.default_rng(21)
    Nx, Nu, Ny, Nz, Nd = 12, 5, 6, 2, 3
    A, B, C = plant(rng, Nx, Nu, Ny, 0.9)
    H = np.eye(Nz, Ny)
    Bd = rng.standard_normal((Nx, Nd)) / np.sqrt(Nx); Cd = 0.1 * rng.standard_normal((Ny, Nd))
    Qs, Rs, usp = np.eye(Ny), 1e-2 * np.eye(Nu), 0.1 * rng.standard_normal((Nu, 1))
    ulb, uub = -np.ones((Nu, 1)), np.ones((Nu, 1))
    ts = ref.TargetSelector(A=A, B=B, C=C, H=H, Bd=Bd, Cd=Cd, usp=usp, Rs=Rs, Qs=Qs, ulb=ulb, uub=uub)
    ysp_l, d_l, xs_l, us_l = [], [], [], []
    for t in range(400):
        if len(ysp_l) >= 24:
            break
        ysp = rng.uniform(0.3, 1.2) * rng.standard_normal((Ny, 1)); d = rng.standard_normal((Nd, 1))
        try:
            xs, us = ts.solve(ysp, d)
        except (np.linalg.LinAlgError, ArithmeticError):
            continue                                          # no steady state inside the input box
        (q, h, b) = ts._setup_changing_matrices(ysp, d)
        z0 = np.concatenate((xs, us)).ravel()
        f = lambda z: 0.5 * z @ (ts.P @ z) + q.ravel() @ z
        g = lambda z: ts.P @ z + q.ravel()
        res = minimize(f, np.zeros(Nx + Nu), jac=g, hess=lambda z: ts.P, method="trust-constr",
                       constraints=[LinearConstraint(ts.tA, b.ravel(), b.ravel())],
                       bounds=Bounds(np.concatenate((-np.inf * np.ones(Nx), ulb.ravel())), np.concatenate((np.inf * np.ones(Nx), uub.ravel()))),
                       options=dict(xtol=1e-12, gtol=1e-12, barrier_tol=1e-12, maxiter=3000))
        if np.abs(res.x - z0).max() > 2e-6 or np.abs(ts.tA @ z0 - b.ravel()).max() > 1e-9:
            continue
        ysp_l.append(ysp[:, 0]); d_l.append(d[:, 0]); xs_l.append(xs[:, 0]); us_l.append(us[:, 0])
    ysp_l, d_l, xs_l, us_l = map(np.array, (ysp_l, d_l, xs_l, us_l))
    nsat = (np.abs(np.abs(us_l) - 1.0) < 1e-9).sum(axis=1)
    np.savez_compressed(os.path.join(HERE, "target.npz"), A=A, B=B, C=C, H=H, Bd=Bd, Cd=Cd, Qs=Qs, Rs=Rs, usp=usp, ulb=ulb, uub=uub,
                        ysp=ysp_l, dhat=d_l, xs=xs_l, us=us_l)
    print("target", ysp_l.shape[0], "pairs, saturated inputs per pair:", np.bincount(nsat))


def make_closed_loop(ref, ce):
    """online_simulation (lib/linearMPC.py:703-718) with the QP controller (:519-701) and with the NN controller
    (lib/controller_evaluation.py:780-892) -- the oracle solves the QPs at the cvxopt seam."""
    import tempfile
    rng = np.random.default_rng(31)
    Nx, Nu, Ny, Nd, N, Nsim = 6, 2, 3, 2, 8, 30
    A, B, C = plant(rng, Nx, Nu, Ny, 0.9)
    Bd = rng.standard_normal((Nx, Nd)) / np.sqrt(Nx); Cd = np.zeros((Ny, Nd))
    H = np.eye(1, Ny)
    Q, R, S = 2.0 * C.T @ C, 0.1 * np.eye(Nu), 0.05 * np.eye(Nu)
    Rs, Qs = 1e-3 * np.eye(Nu), np.eye(Ny)
    Qwx, Qwd, Rv = 1e-4 * np.eye(Nx), 1e-2 * np.eye(Nd), 1e-4 * np.eye(Ny)
    ulb, uub = -np.ones((Nu, 1)), np.ones((Nu, 1))
    setpoints = np.repeat(rng.uniform(-0.4, 0.4, (3, Ny)), Nsim // 3 + 1, axis=0)[:Nsim]      # reachable inside the input box
    disturbances = np.repeat(rng.uniform(-0.5, 0.5, (2, Nd)), Nsim // 2 + 1, axis=0)[:Nsim]
    common = dict(A=A, B=B, C=C, H=H, Qwx=Qwx, Qwd=Qwd, Rv=Rv, xprior=np.zeros((Nx, 1)), dprior=np.zeros((Nd, 1)),
                  Rs=Rs, Qs=Qs, Bd=Bd, Cd=Cd, usp=np.zeros((Nu, 1)), uprev=np.zeros((Nu, 1)), Q=Q, R=R, S=S, ulb=ulb, uub=uub)
    # NN weights: small, so that the loop stays the plant's own stable dynamics plus a bounded input
    dims = [2 * Nx + 2 * Nu, 16, 16, Nu]
    W = []
    for i in range(3):
        W.append(0.3 * rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]))
        if i < 2:
            W.append(0.1 * rng.standard_normal(dims[i + 1]))
    xscale = rng.uniform(0.5, 2.0, Nx)
    out = {}
    old = sys.stdout
    for name in ("mpc", "nn"):
        np.random.seed(17)
        pl = ref.LinearPlantSimulator(A=A, B=B, C=C, Bp=Bd, Rv=Rv, sample_time=1.0, x0=np.zeros((Nx, 1)))
        if name == "mpc":
            ctl = ref.LinearMPCController(N=N, **common)
        else:
            ctl = ce.NeuralNetworkController(regulator_weights=W, xscale=xscale, nnwithuprev=True, **common)
        with tempfile.NamedTemporaryFile("w") as tf:
            try:
                ref.online_simulation(pl, ctl, setpoints=setpoints, disturbances=disturbances, Nsim=Nsim, stdout_filename=tf.name)
            finally:
                sys.stdout.close(); sys.stdout = old
        out[f"{name}_y"] = np.array(pl.y)[:, :, 0]; out[f"{name}_u"] = np.array(pl.u)[:, :, 0]; out[f"{name}_x"] = np.array(pl.x)[:, :, 0]
        out[f"{name}_xhat"] = np.array(ctl.filter.xhat)[:, :, 0]
        out[f"{name}_avg_cost"] = np.array(ctl.average_stage_costs).ravel()
        print("closed loop", name, "max |u|", np.abs(out[f"{name}_u"]).max(), "final avg cost", out[f"{name}_avg_cost"][-1])
    np.savez_compressed(os.path.join(HERE, "closed_loop.npz"), A=A, B=B, C=C, H=H, Bd=Bd, Cd=Cd, Q=Q, R=R, S=S, Rs=Rs, Qs=Qs,
                        Qwx=Qwx, Qwd=Qwd, Rv=Rv, ulb=ulb, uub=uub, N=N, Nsim=Nsim, setpoints=setpoints, disturbances=disturbances,
                        xscale=xscale, nW=len(W), **{f"W{i}": w for i, w in enumerate(W)}, **out)


def main():
    ref, ce = import_reference()
    for ci, (name, (Nx, Nu, Ny, N, rho, qw, sw)) in enumerate(CASES.items()):
        rng = np.random.default_rng(100 + ci)
        A, B, C = plant(rng, Nx, Nu, Ny, rho)
        Q, R, S = qw * C.T @ C, 0.1 * np.eye(Nu), sw * np.eye(Nu)
        ulb, uub = -np.ones((Nu, 1)), np.ones((Nu, 1))
        reg = ref.LinearMPCController.setup_regulator(A=A, B=B, Q=Q, R=R, S=S, N=N, ulb=ulb, uub=uub)
        Aa, Ba, Qa, Ra, Ma = ref.LinearMPCController.get_augmented_matrices_for_regulator(A, B, Q, R, S)
        nb = 6
        x = 2.0 * rng.standard_normal((nb, Nx, 1)); xs = 0.3 * rng.standard_normal((nb, Nx, 1))
        us = rng.uniform(-0.5, 0.5, (nb, Nu, 1)); uprev = us + rng.uniform(-0.3, 0.3, (nb, Nu, 1))
        hs, qs, useqs, x0s, ustar, act = [], [], [], [], [], []
        for b in range(nb):
            # exactly what get_control_sequence does (reference lib/linearMPC.py:682-689)
            reg.ulb = ulb - us[b]; reg.uub = uub - us[b]
            x0 = np.concatenate((x[b] - xs[b], uprev[b] - us[b]))
            h = reg._get_h(x0); q = reg.tq @ x0
            info = {}
            v = oqp.solve_exact(reg.P, q, reg.G, h, info=info)
            a = np.zeros(h.size, bool); a[info["active"]] = True
            useq = ref.LinearMPCController.get_control_sequence(reg, x[b], uprev[b], xs[b], us[b], ulb, uub)
            hs.append(h); qs.append(q); x0s.append(x0); ustar.append(v); act.append(a); useqs.append(useq)
        np.savez_compressed(os.path.join(HERE, f"regulator_{name}.npz"),
                            A=A, B=B, C=C, Q=Q, R=R, S=S, N=N, ulb=ulb, uub=uub,
                            Aaug=Aa, Baug=Ba, Qaug=Qa, Raug=Ra, Maug=Ma,
                            P=reg.P, tq=reg.tq, G=reg.G, tA=reg.tA, tB=reg.tB,
                            tK=np.zeros(0) if reg.tK is None else reg.tK, Krep=reg.Krep, Pf=reg.Pf,
                            reparameterize=reg.reparameterize, x=x, xs=xs, us=us, uprev=uprev,
                            x0=np.array(x0s), h=np.array(hs), q=np.array(qs))
        np.savez_compressed(os.path.join(HERE, f"qp_exact_{name}.npz"),
                            v=np.array(ustar), active=np.array(act), useq=np.array(useqs))
        print(name, "n", N * Nu, "reparam", reg.reparameterize, "active/problem", np.array(act).sum(1))

    # ---- chain driver: reference simulate_offline with the exact solver at the qp seam
    rng = np.random.default_rng(7)
    Nx, Nu, Ny, Nd, N, T = 6, 2, 3, 2, 8, 12
    A, B, C = plant(rng, Nx, Nu, Ny, 0.95)
    Bd = rng.standard_normal((Nx, Nd)) / np.sqrt(Nx); Cd = np.zeros((Ny, Nd))
    H = np.eye(1, Ny)                                   # Nz = 1 < Nu controlled outputs (as in the reference: Nz=4 < Nu=32)
    Q, R, S = 2.0 * C.T @ C, 0.1 * np.eye(Nu), 0.05 * np.eye(Nu)
    Rs, Qs = 1e-3 * np.eye(Nu), np.eye(Ny)
    ulb, uub = -np.ones((Nu, 1)), np.ones((Nu, 1))
    setpoints = np.repeat(rng.uniform(-2, 2, (3, Ny)), 2 * T // 3 + 1, axis=0)[:2 * T]
    disturbances = np.repeat(rng.uniform(-1, 1, (4, Nd)), 2 * T // 4 + 1, axis=0)[:2 * T]
    sim = ref.OfflineSimulator(A=A, B=B, C=C, H=H, Rs=Rs, Qs=Qs, Bd=Bd, Cd=Cd, usp=np.zeros((Nu, 1)),
                               uprev=np.zeros((Nu, 1)), Q=Q, R=R, S=S, ulb=ulb, uub=uub, N=N,
                               xprior=np.zeros((Nx, 1)), setpoints=setpoints, disturbances=disturbances,
                               num_data_gen_task=2, num_process_per_task=1)
    ref.H5pyTool = types.SimpleNamespace(save_training_data=lambda dictionary, filename: dictionary)
    outs = []
    for task in range(2):
        with contextlib.redirect_stdout(io.StringIO()):
            out = ref.simulate_offline(task, 0, "x.h5py", sim.x0, sim.uprev0, sim.A, sim.B, sim.Bd,
                                       sim.regulators[0], sim.ulb, sim.uub, sim.target_selectors[0],
                                       sim.setpoints[task][0], sim.disturbances[task][0])
        outs.append(out)
    np.savez_compressed(os.path.join(HERE, "chain.npz"), A=A, B=B, C=C, H=H, Bd=Bd, Cd=Cd, Q=Q, R=R, S=S,
                        Rs=Rs, Qs=Qs, ulb=ulb, uub=uub, N=N, setpoints=setpoints, disturbances=disturbances,
                        **{f"{k}_{t}": outs[t][k] for t in range(2) for k in ("x", "uprev", "xs", "us", "u")},
                        split_setpoints_0=sim.setpoints[0][0], split_setpoints_1=sim.setpoints[1][0])
    print("chain", {k: outs[0][k].shape for k in ("x", "uprev", "xs", "us", "u")})

    # ---- structured NN: reference numpy forward (lib/controller_evaluation.py:863-892)
    for name, (nx, nu, hid, withu) in {"with_uprev": (5, 3, 16, True), "without_uprev": (6, 2, 24, False)}.items():
        rng = np.random.default_rng(11 + withu)
        din = 2 * nx + (2 if withu else 1) * nu
        dims = [din, hid, hid, hid, nu]
        W = []
        for i in range(4):
            W.append(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]))
            if i < 3:
                W.append(0.2 * rng.standard_normal(dims[i + 1]))
        xscale = rng.uniform(0.5, 3.0, nx)
        ulb, uub = -np.ones((nu, 1)), np.ones((nu, 1))
        stub = types.SimpleNamespace(regulator_weights=W, nnwithuprev=withu, xscale=xscale[:, None], ulb=ulb, uub=uub)
        for meth in ("_get_regulator_nn_output", "_clip_control_input", "_get_control_input", "_get_scaled_x_xs"):
            setattr(stub, meth, types.MethodType(getattr(ce.NeuralNetworkController, meth), stub))
        nb = 9
        x = 2 * rng.standard_normal((nb, nx)); xs = rng.standard_normal((nb, nx))
        us = rng.uniform(-.8, .8, (nb, nu)); uprev = us + rng.uniform(-.5, .5, (nb, nu))
        x[0], uprev[0] = xs[0], us[0]                         # steady-state row: u must equal clip(us)
        u = []
        for b in range(nb):
            xsc, xssc = stub._get_scaled_x_xs(x[b][:, None], xs[b][:, None])
            u.append(stub._get_control_input(xsc, uprev[b][:, None], xssc, us[b][:, None]).ravel())
        np.savez_compressed(os.path.join(HERE, f"nn_{name}.npz"), nx=nx, nu=nu, withuprev=withu, xscale=xscale,
                            ulb=ulb, uub=uub, x=x, xs=xs, us=us, uprev=uprev, u=np.array(u),
                            **{f"W{i}": w for i, w in enumerate(W)}, nW=len(W))
        print("nn", name, np.array(u).shape)

    make_target(ref)
    make_closed_loop(ref, ce)

    # ---- PRBS sampler (lib/controller_evaluation.py:21-47)
    sig = ce.sample_prbs_like(num_change=6, num_steps=60, lb=np.array([[-1.], [0.]]), ub=np.array([[1.], [2.]]),
                              mean_change=10, sigma_change=1, seed=3)
    np.savez_compressed(os.path.join(HERE, "prbs.npz"), signal=sig)
    print("prbs", sig.shape)


if __name__ == "__main__":
    main()
