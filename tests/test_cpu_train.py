"""Training-step counterpart (PyTorch, stock ops) of cdu_train.py / cstrs_train.py: forward parity
with the reference-pinned numpy oracle in Keras weight order, and a short fit that learns."""
import numpy as np
import torch

from industrial_nnmpc_2021_amd.train import RegulatorModel, train_nn_controller
from oracle import nn as onn


def test_torch_model_matches_oracle_forward_and_keras_weight_order():
    rng = np.random.default_rng(0)
    for withu in (True, False):
        nx, nu = 5, 2
        m = RegulatorModel(nx, nu, [None, 16, 16, nu], nnwithuprev=withu)
        W = m.get_weights()
        assert [w.shape for w in W] == [(2 * nx + (2 if withu else 1) * nu, 16), (16,), (16, 16), (16,), (16, nu)]
        x, xs = rng.standard_normal((7, nx)), rng.standard_normal((7, nx))
        us = rng.uniform(-1, 1, (7, nu)); up = rng.uniform(-1, 1, (7, nu))
        t = lambda a: torch.as_tensor(a, dtype=torch.float64)
        got = m(t(x), t(up), t(xs), t(us)).detach().numpy()
        ref = onn.control_input(W, x, up, xs, us, None, None, None, withu)
        assert np.abs(got - ref).max() < 1e-12
        m2 = RegulatorModel(nx, nu, [None, 16, 16, nu], nnwithuprev=withu)
        m2.set_weights(W)
        assert np.abs(m2(t(x), t(up), t(xs), t(us)).detach().numpy() - got).max() == 0


def test_short_fit_learns_a_saturated_linear_law():
    rng = np.random.default_rng(1)
    nx, nu, n = 4, 2, 4096
    K = rng.standard_normal((nu, nx)) * 0.5
    x, xs = rng.standard_normal((n, nx)), 0.2 * rng.standard_normal((n, nx))
    us = rng.uniform(-.3, .3, (n, nu)); up = us + rng.uniform(-.2, .2, (n, nu))
    u = np.clip(us + (x - xs) @ K.T, -1, 1)            # structured: x = xs -> u = us
    data = dict(x=x, uprev=up, xs=xs, us=us, u=u)
    m = RegulatorModel(nx, nu, [None, 32, 32, nu], nnwithuprev=True)
    m, ttime, hist = train_nn_controller(m, data, epochs=30, batch_size=256, device="cpu")
    assert hist[-1][1] < 0.2 * hist[0][1] and ttime > 0
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)
    out = m(t(xs[:5]), t(us[:5]), t(xs[:5]), t(us[:5])).detach().numpy()
    assert np.abs(out - us[:5]).max() < 1e-12         # steady-state property holds by construction
