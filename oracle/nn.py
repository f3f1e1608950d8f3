"""fp64 restatement of the structured-NN controller forward (TEST ORACLE).

Follows NeuralNetworkController._get_control_input / _get_regulator_nn_output /
_clip_control_input / _get_scaled_x_xs of the reference
(lib/controller_evaluation.py:863-892), which is itself the numpy twin of
RegulatorLayerWithUprev.call / RegulatorLayerWithoutUprev.call
(lib/LinearMPCLayers.py:40-61, :91-112).  Batched over rows instead of the
reference's single column vector.  ``weights`` is the Keras ``get_weights()``
list [W1 (in x h), b1, W2, b2, ..., Wout (h x nu)].
"""
import numpy as np


def mlp(weights, z):
    """Rows of z through Dense(relu)...Dense(relu), bias-free linear head (:877-886)."""
    for i in range(0, len(weights) - 1, 2):
        z = np.maximum(z @ weights[i] + weights[i + 1], 0.0)
    return z @ weights[-1]


def control_input(weights, x, uprev, xs, us, xscale=None, ulb=None, uub=None, nnwithuprev=True):
    """u = clip(us + NN(x, [uprev], xs, us) - NN(xs, [us], xs, us))   (:868-875, :888-892)."""
    if xscale is not None:
        x, xs = x / xscale, xs / xscale                                   # :863-866
    if nnwithuprev:
        z1 = np.concatenate((x, uprev, xs, us), axis=1)
        z2 = np.concatenate((xs, us, xs, us), axis=1)
    else:
        z1 = np.concatenate((x, xs, us), axis=1)
        z2 = np.concatenate((xs, xs, us), axis=1)
    u = us + mlp(weights, z1) - mlp(weights, z2)
    if ulb is not None:
        u = np.minimum(np.maximum(u, np.ravel(ulb)), np.ravel(uub))
    return u
