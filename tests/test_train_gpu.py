"""Row f3 of SURVEY section 8 on the GPU: the PyTorch-ROCm training step (counterpart of cdu_train.py:24-62) runs on the
device, learns, and hands its Keras-order weight list to the HIP forward (what the reference pickles after training,
cdu_train.py:107-116, and NeuralNetworkController loads, lib/controller_evaluation.py:780-839)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fit_on_the_gpu_then_deploy_through_the_hip_forward():
    import torch
    from industrial_nnmpc_2021_amd.train import RegulatorModel, train_nn_controller
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    from industrial_nnmpc_2021_amd import controller_evaluation as ce
    assert torch.cuda.is_available()
    rng = np.random.default_rng(1)
    nx, nu, n = 6, 3, 8192
    K = rng.standard_normal((nu, nx)) * 0.5
    x, xs = 1.5 * rng.standard_normal((n, nx)), 0.2 * rng.standard_normal((n, nx))
    us = rng.uniform(-.3, .3, (n, nu)); up = us + rng.uniform(-.2, .2, (n, nu))
    u = np.clip(us + (x - xs) @ K.T, -1, 1)            # a saturated linear law: structured (x = xs -> u = us)
    raw = dict(x=x, uprev=up, xs=xs, us=us, u=u)
    data, xscale = ce._get_data_for_training(data=raw, num_samples=n)          # x, xs divided by xscale (:254-271)
    m = RegulatorModel(nx, nu, [None, 64, 64, nu], nnwithuprev=True)
    m, ttime, hist = train_nn_controller(m, data, epochs=25, batch_size=512, device="cuda")
    assert next(m.parameters()).is_cuda and ttime > 0
    assert hist[-1][1] < 0.25 * hist[0][1]                                     # it learns
    W = m.get_weights()
    net = StructuredNN(W, nx, nu, nnwithuprev=True, xscale=xscale, ulb=-np.ones(nu), uub=np.ones(nu), max_batch=1024)
    k = 1000
    got = net.forward(x[:k], up[:k], xs[:k], us[:k])                           # the HIP forward scales x, xs itself
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        ref = m(t(data["x"][:k]), t(data["uprev"][:k]), t(data["xs"][:k]), t(data["us"][:k])).clamp(-1, 1).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-4
    assert np.abs(got - u[:k]).mean() < 0.1                                    # and the deployed controller follows the law
    net.close()
