"""GPU parity of the reference-interface mirror (DenseQPRegulator / get_control_sequence /
simulate_offline / cvxopt-shaped qp / NN layers) against the golden vectors the
reference itself produced (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("case", ["stable_s0", "stable_s1", "unstable_s0", "unstable_s1"])
def test_get_control_sequence_matches_reference_golden(golden_dir, case):
    from industrial_nnmpc_2021_amd import linearMPC as lm
    g, e = _load(golden_dir, f"regulator_{case}.npz"), _load(golden_dir, f"qp_exact_{case}.npz")
    reg = lm.LinearMPCController.setup_regulator(g["A"], g["B"], g["Q"], g["R"], g["S"], int(g["N"]), g["ulb"], g["uub"],
                                                 max_batch=128)
    nb = g["x0"].shape[0]
    for b in range(nb):                       # reference call path, one problem at a time
        useq = lm.LinearMPCController.get_control_sequence(reg, g["x"][b], g["uprev"][b], g["xs"][b], g["us"][b],
                                                           g["ulb"], g["uub"])
        assert useq.shape == e["useq"][b].shape
        assert np.abs(useq - e["useq"][b]).max() <= 1e-8 * max(1.0, np.abs(useq).max())
        assert np.array_equal(reg.last_info["active"][0], e["active"][b])
    assert len(reg.x0) == nb and len(reg.useq) == nb
    # batched entry point, all problems in one call
    U, info = lm.LinearMPCController.get_control_sequence_batch(
        reg, g["x"][:, :, 0], g["uprev"][:, :, 0], g["xs"][:, :, 0], g["us"][:, :, 0], g["ulb"], g["uub"],
        first_move_only=False)
    assert (info["status"] == 0).all()
    assert np.abs(U - e["useq"][:, :, 0]).max() <= 1e-8 * max(1.0, np.abs(U).max())
    assert np.array_equal(info["active"], e["active"])


def test_offline_simulator_matches_reference_chain(golden_dir, tmp_path, monkeypatch):
    from industrial_nnmpc_2021_amd import linearMPC as lm
    g = _load(golden_dir, "chain.npz")
    nu, Nx = g["B"].shape[1], g["A"].shape[0]
    sim = lm.OfflineSimulator(A=g["A"], B=g["B"], C=g["C"], H=g["H"], Rs=g["Rs"], Qs=g["Qs"], Bd=g["Bd"], Cd=g["Cd"],
                              usp=np.zeros((nu, 1)), uprev=np.zeros((nu, 1)), Q=g["Q"], R=g["R"], S=g["S"],
                              ulb=g["ulb"], uub=g["uub"], N=int(g["N"]), xprior=np.zeros((Nx, 1)),
                              setpoints=g["setpoints"], disturbances=g["disturbances"],
                              num_data_gen_task=1, num_process_per_task=2)   # both chains in lock-step
    monkeypatch.chdir(tmp_path)
    files = sim.generate_data(task_number=0, data_filename="chain_data")
    assert len(files) == 2
    for c, f in enumerate(files):
        if f.endswith(".npz"):
            d = np.load(f)
        else:
            import h5py
            with h5py.File(f, "r") as h:
                d = {k: h[k][()] for k in h}
        for k in ("x", "uprev", "xs", "us", "u"):
            assert d[k].shape == g[f"{k}_{c}"].shape
            assert np.abs(d[k] - g[f"{k}_{c}"]).max() < 1e-7, (c, k)
        assert float(d["data_gen_time"]) > 0


def test_cvxopt_shaped_qp_seam(golden_dir):
    """qp(P, q, G, h) with the reference's dense box G, and the equality form of the target selector."""
    from industrial_nnmpc_2021_amd import cvxopt_shim as cvx
    g, e = _load(golden_dir, "regulator_stable_s1.npz"), _load(golden_dir, "qp_exact_stable_s1.npz")
    for b in range(3):
        sol = cvx.solvers.qp(*[cvx.matrix(a) for a in (g["P"], g["q"][b], g["G"], g["h"][b])])
        x = np.asarray(sol["x"])
        assert x.shape == (g["P"].shape[0], 1) and sol["status"] == "optimal"
        assert np.abs(x[:, 0] - e["v"][b]).max() <= 1e-8 * max(1.0, np.abs(e["v"][b]).max())
    # (the dense G of a re-parameterised regulator: test_chain_target_gpu.test_cvxopt_seam_with_the_dense_G_of_an_unstable_plant)


@pytest.mark.parametrize("name", ["with_uprev", "without_uprev"])
def test_nn_layers_match_reference_golden(golden_dir, name):
    from industrial_nnmpc_2021_amd.LinearMPCLayers import RegulatorLayerWithUprev, RegulatorLayerWithoutUprev
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    g = _load(golden_dir, f"nn_{name}.npz")
    W = [g[f"W{i}"] for i in range(int(g["nW"]))]
    nx, nu, withu = int(g["nx"]), int(g["nu"]), bool(g["withuprev"])
    # controller form: scaling + clipping (NeuralNetworkController._get_control_input)
    net = StructuredNN(W, nx, nu, nnwithuprev=withu, xscale=g["xscale"], ulb=g["ulb"], uub=g["uub"], max_batch=128)
    u = net.forward(g["x"], g["uprev"], g["xs"], g["us"])
    assert np.abs(u - g["u"]).max() < 2e-5          # f32 MFMA arithmetic vs the reference's f64
    assert np.abs(u[0] - np.clip(g["us"][0], -1, 1)).max() < 1e-12   # steady-state row is exact
    # Keras-layer form: no scaling, no clipping
    from oracle import nn as onn
    layer = (RegulatorLayerWithUprev if withu else RegulatorLayerWithoutUprev)([w.shape[1] for w in W[0:-1:2]] + [nu])
    layer.set_weights(W)
    inputs = [g["x"], g["uprev"], g["xs"], g["us"]] if withu else [g["x"], g["xs"], g["us"]]
    out = layer.call(inputs)
    ref = onn.control_input(W, g["x"], g["uprev"], g["xs"], g["us"], None, None, None, withu)
    assert np.abs(out - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())


def test_nn_cdu_shape_batch_vs_oracle():
    """[536, 832, 832, 832, 32] WithoutUprev (the architecture the reference trains for the CDU,
    cdu_train.py:77-80), ragged batch size, f32 tolerance stated: 1e-4 relative."""
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    from oracle import nn as onn
    rng = np.random.default_rng(3)
    nx, nu, hid = 252, 32, 832
    dims = [2 * nx + nu, hid, hid, hid, nu]
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    B = 1000
    x, xs = rng.standard_normal((B, nx)), 0.3 * rng.standard_normal((B, nx))
    us = rng.uniform(-.5, .5, (B, nu))
    xscale = rng.uniform(0.5, 2.0, nx)
    net = StructuredNN(W, nx, nu, nnwithuprev=False, xscale=xscale, ulb=-np.ones(nu), uub=np.ones(nu), max_batch=512)
    u = net.forward(x, None, xs, us)
    ref = onn.control_input(W, x, None, xs, us, xscale, -np.ones(nu), np.ones(nu), False)
    assert np.abs(u - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    assert net.forward(x[:0], None, xs[:0], us[:0]).shape == (0, nu)      # empty batch


def test_nn_bf16_path_tolerance():
    """bf16 weights/activations, f32 accumulate: stated tolerance 3e-2 relative to the output scale
    (the f32 path is held to 1e-4, see test_nn_cdu_shape_batch_vs_oracle)."""
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    from oracle import nn as onn
    rng = np.random.default_rng(4)
    nx, nu, hid = 252, 32, 832
    dims = [2 * nx + nu, hid, hid, hid, nu]
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    B = 700
    x, xs = rng.standard_normal((B, nx)), 0.3 * rng.standard_normal((B, nx))
    us = rng.uniform(-.5, .5, (B, nu))
    x[0] = xs[0]
    net = StructuredNN(W, nx, nu, nnwithuprev=False, max_batch=512, use_bf16=True)
    u = net.forward(x, None, xs, us)
    ref = onn.control_input(W, x, None, xs, us, None, None, None, False)
    assert np.abs(u - ref).max() <= 3e-2 * max(1.0, np.abs(ref).max())
    assert np.abs(u[0] - us[0]).max() < 1e-12       # steady-state row: both passes identical -> exact


def _bf16(a):
    """Round to nearest-even bf16 (returned as float32), like v_cvt_pk_bf16_f32 / the library's weight upload."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def _bf16_forward(W, x, uprev, xs, us, xscale):
    """The structured forward (lib/controller_evaluation.py:863-886) with the bf16 path's roundings: inputs, weights
    and hidden activations in bf16, sums and biases wider, head output unrounded."""
    sc = (1.0 / xscale).astype(np.float32) if xscale is not None else np.ones(x.shape[1], np.float32)

    def mlp(a, b):
        z = [a.astype(np.float32) * sc] + ([b.astype(np.float32)] if b is not None else []) + [xs.astype(np.float32) * sc, us.astype(np.float32)]
        h = _bf16(np.concatenate(z, axis=1)).astype(np.float64)
        nl = (len(W) + 1) // 2
        for l in range(nl - 1):
            h = h @ _bf16(W[2 * l]).astype(np.float64) + W[2 * l + 1].astype(np.float32).astype(np.float64)
            h = _bf16(np.maximum(h, 0.0)).astype(np.float64)
        return (h @ _bf16(W[-1]).astype(np.float64)).astype(np.float32).astype(np.float64)
    return us + (mlp(x, uprev) - mlp(xs, us if uprev is not None else None))


@pytest.mark.parametrize("hid,withu,B,mb", [(832, False, 700, 512), (832, True, 40, 128), (960, False, 1300, 1024),
                                            (448, True, 300, 256), (416, False, 257, 128), (1024, False, 520, 512)])
def test_nn_bf16_wide_tile_kernel_shapes(hid, withu, B, mb):
    """The 256 x 208-tile kernel of the hidden layers (widths >= 416): the reference's CDU widths 832...1024
    (cdu_train.py:77-80) incl. a partial last column tile, one row panel only, several sub-batches, both layer
    types -- against a numpy forward with the same bf16 roundings.  Tolerance 4e-3 of the output scale (f32
    accumulation order can move a hidden activation by one bf16 ulp, 2^-8 relative)."""
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    rng = np.random.default_rng(hid + B)
    nx, nu = 252, 32
    dims = [2 * nx + (2 if withu else 1) * nu, hid, hid, hid, nu]
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    x, xs = rng.standard_normal((B, nx)), 0.3 * rng.standard_normal((B, nx))
    us, up = rng.uniform(-.5, .5, (B, nu)), rng.uniform(-1, 1, (B, nu))
    xscale = rng.uniform(0.5, 2.0, nx)
    net = StructuredNN(W, nx, nu, nnwithuprev=withu, xscale=xscale, max_batch=mb, use_bf16=True)
    u = net.forward(x, up if withu else None, xs, us)
    ref = _bf16_forward(W, x, up if withu else None, xs, us, xscale)
    assert np.abs(u - ref).max() <= 4e-3 * max(1.0, np.abs(ref).max()), np.abs(u - ref).max()
    net.close()


@pytest.mark.parametrize("hid,withu,B,mb", [(832, False, 700, 512), (960, True, 300, 256), (416, False, 257, 128),   # wide-tile kernel
                                            (64, True, 200, 128), (192, False, 333, 256)])                          # 64 / 128-tile kernel
def test_nn_split_bf16_is_f32_grade(hid, withu, B, mb):
    """use_bf16 = "split": activations and weights as bf16 pairs hi + lo, one bf16 GEMM of three times the depth per layer
    (hi hi' + hi lo' + lo hi').  Against the fp64 oracle (lib/controller_evaluation.py:863-892) the result must be as good as
    the f32 path's: 1e-4 of the output scale (the plain bf16 path: 3e-2), on both GEMM kernels, with and without uprev,
    several sub-batches and a partial last column tile."""
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    from oracle import nn as onn
    rng = np.random.default_rng(7 * hid + B)
    nx, nu = (252, 32) if hid >= 416 else (12, 6)
    dims = [2 * nx + (2 if withu else 1) * nu, hid, hid, hid, nu]
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    x, xs = rng.standard_normal((B, nx)), 0.3 * rng.standard_normal((B, nx))
    us, up = rng.uniform(-.5, .5, (B, nu)), rng.uniform(-1, 1, (B, nu))
    x[0] = xs[0]; up[0] = us[0]                                # steady state: u = clip(us) exactly, whatever the weights
    xscale = rng.uniform(0.5, 2.0, nx)
    net = StructuredNN(W, nx, nu, nnwithuprev=withu, xscale=xscale, ulb=-np.ones(nu), uub=np.ones(nu), max_batch=mb, use_bf16="split")
    u = net.forward(x, up if withu else None, xs, us)
    net.close()
    ref = onn.control_input(W, x, up if withu else None, xs, us, xscale, -np.ones(nu), np.ones(nu), withu)
    assert np.abs(u - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), np.abs(u - ref).max()
    assert np.array_equal(u[0], np.clip(us[0], -1, 1))


def test_nn_bf16_wide_tile_kernel_row_slices(monkeypatch):
    """The wide-tile kernel addresses its input with 32-bit byte offsets; the launcher cuts inputs of 2^31 bytes and
    more into row slices.  NNMPC_WIDE_MAX_ROWS forces that path on a small batch: same result as one slice."""
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    rng = np.random.default_rng(11)
    nx, nu, hid, B = 252, 32, 832, 900
    dims = [2 * nx + nu, hid, hid, hid, nu]
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    x, xs = rng.standard_normal((B, nx)), 0.3 * rng.standard_normal((B, nx))
    us = rng.uniform(-.5, .5, (B, nu))
    net = StructuredNN(W, nx, nu, nnwithuprev=False, max_batch=1024, use_bf16=True)
    one = net.forward(x, None, xs, us)
    monkeypatch.setenv("NNMPC_WIDE_MAX_ROWS", "512")          # 2 * 1024 rows -> 4 slices
    cut = net.forward(x, None, xs, us)
    net.close()
    assert np.array_equal(one, cut)


def test_warm_started_chains_equal_cold_and_save_factorizations():
    """Chain driver with the shifted previous active set as warm start: same trajectories
    (every solve is KKT-certified), fewer Cholesky factorisations."""
    from industrial_nnmpc_2021_amd import linearMPC as lm, synthetic
    pl = synthetic.plant("mini_cdu", seed=4)
    Nx, Nu = pl["B"].shape
    rng = np.random.default_rng(2)
    nc, T = 24, 10
    reg = lm.LinearMPCController.setup_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"], pl["N"], pl["ulb"], pl["uub"],
                                                 max_batch=128, solver_options=dict(method="pdip"))
    reg_auto = lm.LinearMPCController.setup_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"], pl["N"], pl["ulb"],
                                                      pl["uub"], max_batch=128)

    class FixedTarget:                       # target pairs given directly (the host target QP is tested elsewhere)
        def __init__(self, xs, us): self.xs, self.us = xs, us
        def solve(self, ysp, d): return self.xs, self.us
    ts = [FixedTarget(0.2 * rng.standard_normal((Nx, 1)), rng.uniform(-.4, .4, (Nu, 1))) for _ in range(nc)]
    sp = [np.zeros((T, 1)) for _ in range(nc)]
    ds = [2.5 * rng.standard_normal((T, 2)) * (np.arange(T)[:, None] % 4 == 0) for _ in range(nc)]
    Bd = rng.standard_normal((Nx, 2))
    x0, u0 = 3.0 * rng.standard_normal((Nx, 1)), np.zeros((Nu, 1))
    # (the loop driven from the host reports the factorisations of every solve; the device-resident loop does not)
    cold = lm.simulate_chains(x0, u0, pl["A"], pl["B"], Bd, reg, pl["ulb"], pl["uub"], ts, sp, ds, warm_start=False, device_resident=False)
    warm = lm.simulate_chains(x0, u0, pl["A"], pl["B"], Bd, reg, pl["ulb"], pl["uub"], ts, sp, ds, warm_start=True, device_resident=False)
    assert (cold["status"] == 0).all() and (warm["status"] == 0).all()
    for k in ("x", "u", "uprev"):
        assert np.abs(cold[k] - warm[k]).max() < 1e-8
    assert np.abs(cold["u"]).max() > 0.999                      # bounds are hit along the chains
    assert warm["factorizations"][:, 1:].sum() < cold["factorizations"][:, 1:].sum()
    # default method (shared-inverse active-set pass, warm-started the same way): same chains, no n^3 work
    auto = lm.simulate_chains(x0, u0, pl["A"], pl["B"], Bd, reg_auto, pl["ulb"], pl["uub"], ts, sp, ds, warm_start=True)
    assert (auto["status"] == 0).all() and auto["factorizations"].sum() == 0
    for k in ("x", "u", "uprev"):
        assert np.abs(cold[k] - auto[k]).max() < 1e-8
