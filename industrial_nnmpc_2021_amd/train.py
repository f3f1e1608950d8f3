"""Training step of the structured NN controller in PyTorch(-ROCm), stock ops only.

Counterpart of the reference's ``cdu_train.py`` / ``cstrs_train.py``
(create_nn_controller :24-38, train_nn_controller :40-62): Keras
``RegulatorModel`` (lib/LinearMPCLayers.py:117-133) compiled with Adam + MSE,
``fit(batch_size=2048, validation_split=0.05)``, ``ModelCheckpoint(monitor=
'val_loss', save_best_only=True)``; afterwards the weights are exchanged as the
Keras ``get_weights()`` list, which is exactly what ``nn.StructuredNN`` /
``LinearMPCLayers.RegulatorLayer*`` take for the HIP forward.

This is the only place PyTorch does arithmetic (BASELINE.json north_star);
it is not part of the accelerated hot path.
"""
import copy
import time

import numpy as np
import torch


class RegulatorModel(torch.nn.Module):
    """u = us + MLP(x,[uprev],xs,us) - MLP(xs,[us],xs,us); hidden Dense(relu), bias-free head.

    ``regulator_dims = [d_in, h1, ..., nu]``: like the reference, element 0 is ignored
    (lib/LinearMPCLayers.py:128-131) and the input width follows from Nx, Nu, nnwithuprev.
    Float64 like the reference (``set_floatx('float64')``, :13).
    """

    def __init__(self, Nx, Nu, regulator_dims, nnwithuprev=True, dtype=torch.float64):
        super().__init__()
        self.Nx, self.Nu, self.nnwithuprev = Nx, Nu, nnwithuprev
        widths = [2 * Nx + (2 if nnwithuprev else 1) * Nu] + list(regulator_dims[1:])
        layers = []
        for i in range(len(widths) - 1):
            last = i == len(widths) - 2
            layers.append(torch.nn.Linear(widths[i], widths[i + 1], bias=not last, dtype=dtype))
        self.layers = torch.nn.ModuleList(layers)
        for lin in self.layers:                       # Keras Dense default: glorot_uniform, zero bias
            torch.nn.init.xavier_uniform_(lin.weight)
            if lin.bias is not None:
                torch.nn.init.zeros_(lin.bias)

    def _mlp(self, z):
        for lin in self.layers[:-1]:
            z = torch.relu(lin(z))
        return self.layers[-1](z)

    def forward(self, x, uprev, xs, us):
        if self.nnwithuprev:
            z1, z2 = torch.cat((x, uprev, xs, us), -1), torch.cat((xs, us, xs, us), -1)
        else:
            z1, z2 = torch.cat((x, xs, us), -1), torch.cat((xs, xs, us), -1)
        return us + self._mlp(z1) - self._mlp(z2)

    def get_weights(self):
        """Keras order: [W1 (in x h), b1, ..., Wout (h x Nu)] as float64 numpy arrays."""
        out = []
        for lin in self.layers:
            out.append(lin.weight.detach().cpu().double().numpy().T.copy())
            if lin.bias is not None:
                out.append(lin.bias.detach().cpu().double().numpy().copy())
        return out

    def set_weights(self, weights):
        it = iter(weights)
        with torch.no_grad():
            for lin in self.layers:
                lin.weight.copy_(torch.as_tensor(np.asarray(next(it)).T, dtype=lin.weight.dtype))
                if lin.bias is not None:
                    lin.bias.copy_(torch.as_tensor(np.asarray(next(it)), dtype=lin.bias.dtype))


def train_nn_controller(model, data, *, epochs=1500, batch_size=2048, validation_split=0.05, lr=1e-3,
                        device=None, seed=1, log=None):
    """Adam + MSE on ``data`` = dict(x, uprev, xs, us, u) (rows = samples, already scaled like
    the reference's _get_data_for_training).  Keras semantics: the LAST fraction of the rows is
    the validation set, the rest is reshuffled every epoch; the weights of the best validation
    epoch are restored at the end.  Returns (model, training_time, history)."""
    torch.manual_seed(seed)
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    model = model.to(device)
    dt = next(model.parameters()).dtype
    T = {k: torch.as_tensor(np.asarray(data[k]), dtype=dt, device=device) for k in ("x", "xs", "us", "u")}
    T["uprev"] = (torch.as_tensor(np.asarray(data["uprev"]), dtype=dt, device=device)
                  if model.nnwithuprev else torch.zeros_like(T["us"]))
    n = T["x"].shape[0]
    nval = int(n * validation_split)
    ntr = n - nval
    tr = {k: v[:ntr] for k, v in T.items()}
    va = {k: v[ntr:] for k, v in T.items()}
    opt = torch.optim.Adam(model.parameters(), lr=lr, eps=1e-7)   # Keras Adam defaults
    best, best_state, hist = float("inf"), None, []
    t0 = time.time()
    for ep in range(epochs):
        model.train()
        perm = torch.randperm(ntr, device=device)
        run = 0.0
        for i in range(0, ntr, batch_size):
            idx = perm[i:i + batch_size]
            pred = model(tr["x"][idx], tr["uprev"][idx], tr["xs"][idx], tr["us"][idx])
            loss = torch.mean((pred - tr["u"][idx]) ** 2)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            run += float(loss.detach()) * idx.numel()
        model.eval()
        with torch.no_grad():
            vl = float(torch.mean((model(va["x"], va["uprev"], va["xs"], va["us"]) - va["u"]) ** 2)) if nval else run / ntr
        hist.append((run / ntr, vl))
        if vl < best:                                   # ModelCheckpoint(save_best_only=True)
            best, best_state = vl, copy.deepcopy(model.state_dict())
        if log:
            log(f"epoch {ep + 1}/{epochs} loss {run / ntr:.3e} val_loss {vl:.3e}")
    if best_state is not None:
        model.load_state_dict(best_state)
    return model, time.time() - t0, hist
