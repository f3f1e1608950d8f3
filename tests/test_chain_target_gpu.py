"""Rows f1 / f2 / f4 of SURVEY section 8 on the GPU: the batched target selector, the device-resident lock-step chains,
the closed loop (config 1) and the dense-G seam -- against vectors the reference itself produced (tests/golden/
make_golden.py: target.npz, chain.npz, closed_loop.npz, regulator_unstable_*.npz)."""
import io
import os
import contextlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_target_selector_hip_matches_reference_pairs(golden_dir):
    """nnmpc_ts_solve_batch (reduced problem, one wave per pair) against TargetSelector.solve of the reference; the
    fixture pairs were confirmed by scipy's trust-constr on the full-space problem when they were generated."""
    from industrial_nnmpc_2021_amd import linearMPC as lm
    g = _load(golden_dir, "target.npz")
    ts = lm.TargetSelector(A=g["A"], B=g["B"], C=g["C"], H=g["H"], Bd=g["Bd"], Cd=g["Cd"], usp=g["usp"], Rs=g["Rs"], Qs=g["Qs"],
                           ulb=g["ulb"], uub=g["uub"])
    M = g["ysp"].shape[0]
    # batched, with duplicates: 3 copies of every pair in shuffled order are solved once each
    idx = np.random.default_rng(0).permutation(np.tile(np.arange(M), 3))
    Xs, Us = ts.solve_batch(g["ysp"][idx], g["dhat"][idx])
    assert ts._device().last_distinct == M
    assert np.abs(Us - g["us"][idx]).max() < 1e-8 and np.abs(Xs - g["xs"][idx]).max() < 1e-8
    sat = np.abs(np.abs(g["us"]) - 1.0) < 1e-9
    assert sat.sum(axis=1).max() == 3 and sat.sum(axis=1).min() <= 1         # up to Nu - Nz inputs at a bound: the free ones just carry the Nz equalities
    # reference signature, one pair at a time (+ the one-entry cache of repeated inputs)
    for i in (0, 0, 5):
        xs, us = ts.solve(g["ysp"][i][:, None], g["dhat"][i][:, None])
        assert xs.shape == (g["A"].shape[0], 1) and np.abs(us[:, 0] - g["us"][i]).max() < 1e-8
    assert len(ts.xs) == 3
    # a setpoint no steady state inside the input box reaches: loud failure, not a silent wrong answer
    with pytest.raises(ArithmeticError):
        ts.solve_batch(50.0 * np.ones((1, g["ysp"].shape[1])), np.zeros((1, g["dhat"].shape[1])))


def test_target_selector_hip_random_problems_vs_full_space_oracle():
    """CDU-shaped target problem (Nx=252, Nu=32, Nz=4): KKT conditions of the FULL-space problem evaluated in numpy."""
    from industrial_nnmpc_2021_amd import linearMPC as lm, synthetic
    pl = synthetic.plant("cdu", 3)
    rng = np.random.default_rng(5)
    A, B, Cm = pl["A"], pl["B"], pl["C"]
    Nx, Nu = B.shape
    Ny, Nz, Nd = Cm.shape[0], 4, 5
    H = np.zeros((Nz, Ny)); H[np.arange(Nz), Ny - Nz + np.arange(Nz)] = 1.0      # the last 4 outputs (cdu_parameters.py:135-143)
    Bd = rng.standard_normal((Nx, Nd)) / np.sqrt(Nx); Cd = np.zeros((Ny, Nd))
    ts = lm.TargetSelector(A=A, B=B, C=Cm, H=H, Bd=Bd, Cd=Cd, usp=np.zeros((Nu, 1)), Rs=1e-3 * np.eye(Nu), Qs=np.eye(Ny),
                           ulb=pl["ulb"], uub=pl["uub"])
    M = 200
    Ysp = 0.15 * rng.standard_normal((M, Ny)); Dh = 0.3 * rng.standard_normal((M, Nd))
    Xs, Us = ts.solve_batch(Ysp, Dh)
    z = np.concatenate((Xs, Us), axis=1)
    worst_stat = 0.0
    for i in range(M):
        q, h, b = ts._setup_changing_matrices(Ysp[i][:, None], Dh[i][:, None])
        assert np.abs(ts.tA @ z[i] - b[:, 0]).max() < 1e-8                       # equalities (:262-266)
        assert (Us[i] <= 1 + 1e-9).all() and (Us[i] >= -1 - 1e-9).all()
        # stationarity: P z + q + tA' y + [0; mu] = 0 with mu >= 0 at uub, <= 0 at ulb, 0 on free inputs
        g = ts.P @ z[i] + q[:, 0]
        free = np.abs(np.abs(Us[i]) - 1.0) > 1e-9
        cols = np.concatenate((np.ones(Nx, bool), free))
        y, res = np.linalg.lstsq(ts.tA.T[cols], -g[cols], rcond=None)[:2]
        r = g + ts.tA.T @ y
        worst_stat = max(worst_stat, np.abs(r[cols]).max())
        mu = -r[Nx:][~free]
        assert (np.sign(mu) == np.sign(Us[i][~free])).all() or (~free).sum() == 0
    assert worst_stat < 1e-7
    assert (np.abs(np.abs(Us) - 1.0) < 1e-9).any()                                # some inputs do sit on their bounds


def test_device_chains_equal_host_driven_loop_and_reference_format(tmp_path, monkeypatch):
    """nnmpc_chain_run (state, targets, records in HBM) == the step-by-step loop; several calls continue a chain;
    generate_data / _post_process_data / _get_data_for_training produce the reference's data set layout."""
    from industrial_nnmpc_2021_amd import linearMPC as lm, synthetic, controller_evaluation as ce
    from industrial_nnmpc_2021_amd.chain import DeviceChains
    pl = synthetic.plant("mini_cdu", seed=4)
    Nx, Nu = pl["B"].shape
    rng = np.random.default_rng(2)
    nc, T, Nd = 10, 9, 2
    reg = lm.LinearMPCController.setup_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"], pl["N"], pl["ulb"], pl["uub"], max_batch=128)

    class FixedTarget:
        def __init__(self, xs, us): self.xs, self.us = xs, us
        def solve(self, ysp, d): return self.xs, self.us
    ts = [FixedTarget(0.2 * rng.standard_normal((Nx, 1)), rng.uniform(-.4, .4, (Nu, 1))) for _ in range(nc)]
    sp = [np.zeros((T, 1)) for _ in range(nc)]
    ds = [2.5 * rng.standard_normal((T, Nd)) * (np.arange(T)[:, None] % 4 == 0) for _ in range(nc)]
    Bd = rng.standard_normal((Nx, Nd))
    x0, u0 = 3.0 * rng.standard_normal((Nx, 1)), np.zeros((Nu, 1))
    host = lm.simulate_chains(x0, u0, pl["A"], pl["B"], Bd, reg, pl["ulb"], pl["uub"], ts, sp, ds, device_resident=False)
    dev = lm.simulate_chains(x0, u0, pl["A"], pl["B"], Bd, reg, pl["ulb"], pl["uub"], ts, sp, ds, device_resident=True)
    cold = lm.simulate_chains(x0, u0, pl["A"], pl["B"], Bd, reg, pl["ulb"], pl["uub"], ts, sp, ds, warm_start=False)
    assert (dev["status"] == 0).all() and np.abs(dev["u"]).max() > 0.999
    for k in ("x", "uprev", "xs", "us", "u"):
        assert dev[k].shape == host[k].shape
        assert np.abs(dev[k] - host[k]).max() < 1e-9, k
        assert np.abs(dev[k] - cold[k]).max() < 1e-9, k
    # two calls of T1 and T - T1 steps continue the same chains
    Xs = np.stack([np.tile(t.xs.T, (T, 1)) for t in ts], axis=1); Us = np.stack([np.tile(t.us.T, (T, 1)) for t in ts], axis=1)
    D = np.stack(ds, axis=1)
    ch = DeviceChains(reg._solver(), nc, pl["A"], pl["B"], Bd, pl["ulb"], pl["uub"], x0, u0)
    a = ch.run(Xs[:4], Us[:4], D[:4]); b = ch.run(Xs[4:], Us[4:], D[4:])
    assert np.abs(np.concatenate((a["u"], b["u"])) - np.swapaxes(dev["u"], 0, 1)).max() < 1e-9
    ch.reset()
    c = ch.run(Xs[:2], Us[:2], D[:2])
    assert np.abs(c["x"][0] - x0.T).max() == 0 and np.abs(c["u"] - a["u"][:2]).max() < 1e-12
    ch.close()
    # uncertified solves are loud: a regulator with a round budget of 1 and no fallback leaves some problems unfinished
    reg1 = lm.LinearMPCController.setup_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"], pl["N"], pl["ulb"], pl["uub"], max_batch=128,
                                                  solver_options=dict(method="asm", asm_max_rounds=1))
    with pytest.raises(RuntimeError, match="not certified"):
        lm.simulate_chains(x0, u0, pl["A"], pl["B"], Bd, reg1, pl["ulb"], pl["uub"], ts, sp, ds)
    # data set layout: per-chain files -> one file -> training arrays
    monkeypatch.chdir(tmp_path)
    Ny = 3
    Cm = rng.standard_normal((Ny, Nx)) / np.sqrt(Nx)
    sim = lm.OfflineSimulator(A=pl["A"], B=pl["B"], C=Cm, H=np.eye(1, Ny), Rs=1e-2 * np.eye(Nu), Qs=np.eye(Ny), Bd=Bd, Cd=np.zeros((Ny, Nd)),
                              usp=np.zeros((Nu, 1)), uprev=u0, Q=pl["Q"], R=pl["R"], S=pl["S"], ulb=pl["ulb"], uub=pl["uub"], N=pl["N"],
                              xprior=np.zeros((Nx, 1)), setpoints=0.05 * np.repeat(rng.standard_normal((4, Ny)), 6, axis=0),
                              disturbances=0.2 * np.repeat(rng.standard_normal((6, Nd)), 4, axis=0), num_data_gen_task=2, num_process_per_task=3)
    for task in range(2):
        files = sim.generate_data(task_number=task, data_filename="data.h5py")
        assert len(files) == 3
    data = ce._post_process_data(data_filename="data.h5py", num_data_gen_task=2, num_process_per_task=3)
    assert data["x"].shape == (24, Nx) and data["u"].shape == (24, Nu) and (data["status"] == 0).all()
    assert float(data["data_gen_time"]) > 0
    one = ce._load_training_data("1-2-data.h5py")
    assert np.array_equal(data["x"][20:24], one["x"])                       # task-major, then process: rows of chain (1, 2) last
    tr, xscale = ce._get_data_for_training(data=data, num_samples=20)
    assert tr["x"].shape == (20, Nx) and np.allclose(xscale, 0.5 * (data["x"][:20].max(0) - data["x"][:20].min(0)))
    assert np.allclose(tr["x"] * xscale, data["x"][:20]) and np.array_equal(tr["u"], data["u"][:20])
    raw = ce._get_data_for_training(data=data, num_samples=20, scale=False)
    assert isinstance(raw, dict) and np.array_equal(raw["x"], data["x"][:20])


def test_closed_loop_matches_reference_online_simulation(golden_dir):
    """Config 1 (cstrs_mpc.py / *_neural_network.py plumbing): online_simulation with the QP controller and with the NN
    controller reproduces the reference's closed-loop trajectories (same np.random stream for the measurement noise)."""
    from industrial_nnmpc_2021_amd import linearMPC as lm, controller_evaluation as ce
    g = _load(golden_dir, "closed_loop.npz")
    Nx, Nu = g["B"].shape
    Ny, Nd, Nsim = g["C"].shape[0], g["Bd"].shape[1], int(g["Nsim"])
    common = dict(A=g["A"], B=g["B"], C=g["C"], H=g["H"], Qwx=g["Qwx"], Qwd=g["Qwd"], Rv=g["Rv"], xprior=np.zeros((Nx, 1)),
                  dprior=np.zeros((Nd, 1)), Rs=g["Rs"], Qs=g["Qs"], Bd=g["Bd"], Cd=g["Cd"], usp=np.zeros((Nu, 1)), uprev=np.zeros((Nu, 1)),
                  Q=g["Q"], R=g["R"], S=g["S"], ulb=g["ulb"], uub=g["uub"])
    W = [g[f"W{i}"] for i in range(int(g["nW"]))]
    for name, tol in (("mpc", 1e-6), ("nn", 2e-4)):
        np.random.seed(17)
        plant = lm.LinearPlantSimulator(A=g["A"], B=g["B"], C=g["C"], Bp=g["Bd"], Rv=g["Rv"], sample_time=1.0, x0=np.zeros((Nx, 1)))
        if name == "mpc":
            ctl = lm.LinearMPCController(N=int(g["N"]), **common)
        else:
            ctl = ce.NeuralNetworkController(regulator_weights=W, xscale=g["xscale"], nnwithuprev=True, **common)
        with contextlib.redirect_stdout(io.StringIO()):
            lm.online_simulation(plant, ctl, setpoints=g["setpoints"], disturbances=g["disturbances"], Nsim=Nsim)
        y, u, x = np.array(plant.y)[:, :, 0], np.array(plant.u)[:, :, 0], np.array(plant.x)[:, :, 0]
        assert y.shape == g[f"{name}_y"].shape and u.shape == g[f"{name}_u"].shape
        assert np.abs(u - g[f"{name}_u"]).max() < tol, name
        assert np.abs(y - g[f"{name}_y"]).max() < tol and np.abs(x - g[f"{name}_x"]).max() < tol
        assert np.abs(np.array(ctl.filter.xhat)[:, :, 0] - g[f"{name}_xhat"]).max() < tol
        assert np.abs(np.array(ctl.average_stage_costs).ravel() - g[f"{name}_avg_cost"]).max() < 10 * tol
        assert len(ctl.computation_times) == Nsim and np.abs(u).max() > 0.999      # the input bounds are hit in closed loop


@pytest.mark.parametrize("case", ["unstable_s0", "unstable_s1"])
def test_cvxopt_seam_with_the_dense_G_of_an_unstable_plant(golden_dir, case):
    """qp(P, q, G, h) with G = tE (I + tK tB) (reference :476-479): the UNMODIFIED reference classes work for unstable
    plants through the shim -- the dense G is recognised, mapped to the box QP in input space and back."""
    from industrial_nnmpc_2021_amd import cvxopt_shim as cvx
    g, e = _load(golden_dir, f"regulator_{case}.npz"), _load(golden_dir, f"qp_exact_{case}.npz")
    assert bool(g["reparameterize"])
    for b in range(g["x0"].shape[0]):
        sol = cvx.solvers.qp(*[cvx.matrix(a) for a in (g["P"], g["q"][b], g["G"], g["h"][b])])
        v = np.asarray(sol["x"])
        assert v.shape == (g["P"].shape[0], 1) and sol["status"] == "optimal"
        assert np.abs(v - e["v"][b].reshape(-1, 1)).max() <= 1e-8 * max(1.0, np.abs(e["v"][b]).max())
    with pytest.raises(NotImplementedError):
        cvx.solvers.qp(g["P"], g["q"][0], np.vstack((g["G"][1:], g["G"][:1])), g["h"][0])   # not a regulator's G
