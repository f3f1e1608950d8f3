"""Counterparts of the hot-path pieces of the reference's lib/controller_evaluation.py:
the NN controller (forward on the GPU) and the PRBS sampler / training-data scaling
that sit either side of the offline data-generation path."""
import itertools
import time

import numpy as np

from .linearMPC import LinearMPCController, _save_training_data
from .nn import StructuredNN


def _sample_repeats(num_change, num_simulation_steps, mean_change, sigma_change):
    """(reference :21-29; the removed np.int alias is not used)"""
    repeat = sigma_change * np.random.randn(num_change - 1) + mean_change
    repeat = np.floor(repeat)
    repeat = np.where(repeat <= 0., 0., repeat)
    repeat = np.append(repeat, num_simulation_steps - int(np.sum(repeat)))
    return repeat.astype(int)


def sample_prbs_like(*, num_change, num_steps, lb, ub, mean_change, sigma_change, seed=1):
    """PRBS-like piecewise-constant signal, same stream of numpy draws as the reference (:31-47)."""
    signal_dimension = lb.shape[0]
    lb, ub = lb.squeeze(), ub.squeeze()
    np.random.seed(seed)
    values = (ub - lb) * np.random.rand(num_change, signal_dimension) + lb
    repeat = _sample_repeats(num_change, num_steps, mean_change, sigma_change)
    return np.repeat(values, repeat, axis=0)


def _get_data_for_training(*, data, num_samples, scale=True):
    """First num_samples rows of the generated data; with ``scale`` also xscale = (max - min)/2 of x, x and xs divided
    by it, returned as (data, xscale) -- without, the dict alone (reference :254-271, same return convention)."""
    out = {k: np.asarray(data[k])[0:num_samples, :] for k in ("x", "uprev", "xs", "us", "u")}
    if not scale:
        return out
    xscale = 0.5 * (np.max(out["x"], axis=0) - np.min(out["x"], axis=0))
    out["x"] = out["x"] / xscale
    out["xs"] = out["xs"] / xscale
    return (out, xscale)


get_data_for_training = _get_data_for_training


def _load_training_data(filename):
    """One array per key (reference lib/python_utils.py:44-51); .h5 when h5py wrote it, else the .npz stand-in."""
    try:
        import h5py
        with h5py.File(filename, "r") as f:
            return {k: np.asarray(f.get(k)) for k in f.keys()}
    except (ImportError, OSError):
        with np.load(filename if filename.endswith(".npz") else filename + ".npz") as f:
            return {k: f[k] for k in f.files}


def _post_process_data(*, data_filename, num_data_gen_task, num_process_per_task):
    """Concatenate the per-chain files '<task>-<proc>-<data_filename>' that generate_data wrote, task-major like the
    reference (:273-295): rows of x / uprev / xs / us / u (and the solver's status) stacked, data_gen_time averaged;
    the result is saved under data_filename and returned."""
    training_data = {}
    for (task, process) in itertools.product(range(num_data_gen_task), range(num_process_per_task)):
        sub = _load_training_data(str(task) + '-' + str(process) + '-' + data_filename)
        for key, value in sub.items():
            training_data.setdefault(key, []).append(value)
    for key in training_data:
        if key == 'data_gen_time':
            training_data[key] = np.mean(np.asarray(training_data[key]))
        else:
            training_data[key] = np.concatenate(training_data[key], axis=0)
    _save_training_data(training_data, data_filename)
    return training_data


def relu(x):
    return np.where(x < 0, 0., x)


class NeuralNetworkController(LinearMPCController):
    """Closed-loop NN controller (reference :780-892); the structured forward runs on the GPU."""

    def __init__(self, *, A, B, C, H, Qwx, Qwd, Rv, xprior, dprior, Rs, Qs, Bd, Cd, usp, uprev,
                 ulb, uub, regulator_weights, xscale, nnwithuprev, Q, R, S):
        self.A, self.B, self.C, self.H = A, B, C, H
        self.Nx, self.Nu, self.Ny, self.Nd = A.shape[0], B.shape[1], C.shape[0], Bd.shape[1]
        self.Qwx, self.Qwd, self.Rv, self.xprior, self.dprior = Qwx, Qwd, Rv, xprior, dprior
        self.Qs, self.Rs, self.Bd, self.Cd, self.usp = Qs, Rs, Bd, Cd, usp
        self.uprev, self.ulb, self.uub, self.Q, self.R, self.S = uprev, ulb, uub, Q, R, S
        self.regulator_weights = regulator_weights
        self.xscale = xscale[:, np.newaxis]
        self.nnwithuprev = nnwithuprev
        self.filter = LinearMPCController.setup_filter(A=A, B=B, C=C, Bd=Bd, Cd=Cd, Qwx=Qwx, Qwd=Qwd, Rv=Rv,
                                                       xprior=xprior, dprior=dprior)
        self.target_selector = LinearMPCController.setup_target_selector(A=A, B=B, C=C, H=H, Bd=Bd, Cd=Cd, usp=usp,
                                                                         Qs=Qs, Rs=Rs, ulb=ulb, uub=uub)
        (_, _, self.Qaug, self.Raug, self.Maug) = LinearMPCController.get_augmented_matrices_for_regulator(A, B, Q, R, S)
        self.computation_times = []
        self.average_stage_costs = [np.zeros((1, 1))]
        # x, xs arrive already divided by xscale (control_law does it), so the kernel gets xscale = None
        self._net = StructuredNN(regulator_weights, self.Nx, self.Nu, nnwithuprev=nnwithuprev,
                                 ulb=ulb, uub=uub, max_batch=1024)

    def control_law(self, ysp, y):
        (xhat, dhat) = LinearMPCController.get_state_estimates(self.filter, y, self.uprev, self.Nx)
        (xs, us) = LinearMPCController.get_target_pair(self.target_selector, ysp, dhat)
        tstart = time.time()
        (xhat_scaled, xs_scaled) = self._get_scaled_x_xs(xhat, xs)
        useq_nn = self._get_control_input(xhat_scaled, self.uprev, xs_scaled, us)
        tend = time.time()
        avg_ell = LinearMPCController.get_updated_average_stage_cost(
            xhat, self.uprev, xs, us, useq_nn[0:self.Nu, :], self.Qaug, self.Raug, self.Maug,
            self.average_stage_costs[-1], len(self.average_stage_costs))
        self.average_stage_costs.append(avg_ell)
        self.uprev = useq_nn[0:self.Nu, :]
        self.computation_times.append(tend - tstart)
        return self.uprev

    def _get_scaled_x_xs(self, x, xs):
        return (x / self.xscale, xs / self.xscale)

    def _get_control_input(self, x, uprev, xs, us):
        """(Nx,1),(Nu,1),(Nx,1),(Nu,1) -> (Nu,1)   (reference :868-875)."""
        return self._get_control_input_batch(x.T, uprev.T, xs.T, us.T).T

    def _get_control_input_batch(self, X, Uprev, Xs, Us):
        """Rows are samples: (B, Nx), (B, Nu), (B, Nx), (B, Nu) -> (B, Nu)."""
        return self._net.forward(X, Uprev if self.nnwithuprev else None, Xs, Us)
