"""What a model registers with RecommenderABC._register_trainable(s) (DRecPy/Recommender/recommender_abc.py:266-282).

The reference registers tf.Variable / tf.keras Layer / tf.keras Model objects and differentiates them with a tape.  Here
the trainable state lives in device arrays owned by an engine and is updated by fused HIP kernels, so what is registered
are HANDLES over those arrays:

    Variable         <->  tf.Variable          -> RecommenderABC.trainable_weights
    TrainableLayer   <->  tf.keras.layers.Layer -> RecommenderABC.trainable_layers
    TrainableModel   <->  tf.keras.models.Model -> RecommenderABC.trainable_models

The three lists keep the reference's meaning: their concatenation, weights first (recommender_abc.py:194-196), is the
order of the per-step `optimizer.apply_gradients` calls (:328-334) and therefore fixes every variable's Adam step counter
t = n_registered * step + position + 1.
"""
import numpy as np


class Variable:
    """A trainable fp32 array on the device.  `tensor` may be re-bound by an engine to a slice of one of its own arrays
    (DmfEngine.bind_prediction_scale), after which engine and handle see the same memory."""

    def __init__(self, initial_value, name=None, device='cuda:0'):
        import torch
        self.name = name
        self.tensor = torch.as_tensor(np.asarray(initial_value, dtype=np.float32)).to(device)
        self._consumed_by = None

    @classmethod
    def over(cls, tensor, name=None):
        """Handle over an array an engine already owns (e.g. CDAE's W, W_, V, b, b_ tables)."""
        v = cls.__new__(cls)
        v.name, v.tensor, v._consumed_by = name, tensor, None
        return v

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    def numpy(self):
        return self.tensor.detach().cpu().numpy().copy()

    def assign(self, value):
        import torch
        self.tensor.copy_(torch.as_tensor(np.asarray(value, dtype=np.float32)).to(self.tensor.device).reshape(self.tensor.shape))
        return self

    def _rebind(self, view):
        view.copy_(self.tensor.reshape(view.shape))
        self.tensor = view

    def __mul__(self, other):
        return self.tensor * other

    __rmul__ = __mul__

    def __repr__(self):
        return f'<drecpy_amd.Variable {self.name or ""} shape={self.shape}>'


class _Handle:
    def __init__(self, name, weights_fn):
        self.name = name
        self._weights_fn = weights_fn
        self._consumed_by = None
        self.losses = []                 # Keras regularisation losses: computed inside the fused step instead

    @property
    def trainable_weights(self):
        """The device arrays (views) this handle stands for, in the layer's own order (kernel, bias, ...)."""
        return list(self._weights_fn())

    def __repr__(self):
        return f'<drecpy_amd.{type(self).__name__} {self.name}>'


class TrainableLayer(_Handle):
    """Stands for one tf.keras.layers.Layer of the reference model (e.g. Caser's embeddings, convolutions and dense layer)."""


class TrainableModel(_Handle):
    """Stands for one tf.keras.models.Model of the reference model (e.g. DMF's user_nn / item_nn Sequential towers)."""
