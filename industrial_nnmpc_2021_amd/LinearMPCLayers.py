"""Structured-NN regulator layers with the reference's names and call signatures
(lib/LinearMPCLayers.py), forward pass on the GPU.

The reference's classes are Keras layers (training graph, TF float64); these are
inference-side counterparts: same constructor argument (``layer_dims``), same
``call(inputs)`` input order ([x, uprev, xs, us] / [x, xs, us]), same output
``us + NN(x, ., xs, us) - NN(xs, ., xs, us)``, weights exchanged in Keras
``get_weights()`` order ([W1 (in x h), b1, ..., Wout (h x Nu)], the list the
reference pickles after training, cdu_train.py:107-116).
"""
import numpy as np

from .nn import StructuredNN


class _RegulatorLayer:
    _with_uprev = True

    def __init__(self, layer_dims, trainable=False, name=None, max_batch=65536):
        self.layer_dims = list(layer_dims)
        self.name = name
        self._weights = None
        self._net = None
        self._max_batch = max_batch

    def set_weights(self, weights):
        """Keras order; hidden widths must match layer_dims (reference :28-32)."""
        hidden = [w.shape[1] for w in weights[0:-1:2]] + [weights[-1].shape[1]]
        if hidden != self.layer_dims:
            raise ValueError(f"weights give layer widths {hidden}, layer_dims is {self.layer_dims}")
        self._weights = [np.asarray(w, np.float64) for w in weights]
        self._net = None

    def get_weights(self):
        return list(self._weights)

    def _ensure(self, nx, nu):
        if self._weights is None:
            raise RuntimeError("set_weights() first (weights come from training, reference cdu_train.py)")
        if self._net is None:
            self._net = StructuredNN(self._weights, nx, nu, nnwithuprev=self._with_uprev,
                                     max_batch=self._max_batch)
        return self._net

    def __call__(self, inputs):
        return self.call(inputs)


class RegulatorLayerWithUprev(_RegulatorLayer):
    """u = us + NN(x, uprev, xs, us) - NN(xs, us, xs, us)   (reference :15-61)."""
    _with_uprev = True

    def call(self, inputs):
        [x, uprev, xs, us] = [np.asarray(a, np.float64) for a in inputs]
        return self._ensure(x.shape[1], us.shape[1]).forward(x, uprev, xs, us)


class RegulatorLayerWithoutUprev(_RegulatorLayer):
    """u = us + NN(x, xs, us) - NN(xs, xs, us)   (reference :66-112)."""
    _with_uprev = False

    def call(self, inputs):
        [x, xs, us] = [np.asarray(a, np.float64) for a in inputs]
        return self._ensure(x.shape[1], us.shape[1]).forward(x, None, xs, us)


class RegulatorModel:
    """Counterpart of the Keras RegulatorModel (reference :117-133): regulator_dims[0]
    is ignored exactly like there; inputs [x, (uprev), xs, us]."""

    def __init__(self, Nx, Nu, regulator_dims, nnwithuprev=True):
        self.Nx, self.Nu, self.nnwithuprev = Nx, Nu, nnwithuprev
        cls = RegulatorLayerWithUprev if nnwithuprev else RegulatorLayerWithoutUprev
        self.regulator = cls(layer_dims=regulator_dims[1:])

    def set_weights(self, weights):
        self.regulator.set_weights(weights)

    def get_weights(self):
        return self.regulator.get_weights()

    def predict(self, x, batch_size=None):
        return self.regulator.call(list(x))

    __call__ = predict
