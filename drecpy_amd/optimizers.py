"""Optimizer objects for `fit(optimizer=...)` and `RecommenderABC._register_optimizer` (recommender_abc.py:153-156, 284).

The reference accepts any tf.keras optimizer.  The engine's update rules are hand-written kernels, so the accepted set is:

    None / 'adam' / Adam(...)          Keras Adam (default: learning_rate of fit(), beta_1 .9, beta_2 .999, epsilon 1e-7)
    'adagrad' / Adagrad(...)           Keras Adagrad (initial_accumulator_value .1, epsilon 1e-7)   — CDAE mode='sampled' only
    'rowwise_adagrad'                  one accumulator per table row (engine extension)            — CDAE mode='sampled' only

Anything else — a TensorFlow optimizer object included — is rejected by `resolve` with that list in the message.

`apply_gradients` is the reference's call (one call = one `iterations` tick shared by all pairs of the call) running
`drx_adam_dense` on the device: what `RecommenderABC._update_weights` uses for variables a model updates outside a fused
step.
"""
import numpy as np


class Adam:
    kind = 'adam'

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self._slots = {}              # id(Variable) -> (weak reference, m, v);  ('view', storage, offset, shape, strides) -> (storage, m, v)

    def reset(self):
        """A fresh fit(): the call counter and the moments start over (RecommenderABC.fit registers the optimizer anew)."""
        self.iterations = 0
        self._slots = {}

    def _moments(self, owner, p):
        """(m, v) of one variable.  A Variable OBJECT is keyed by identity (weak reference: a new object at a recycled id starts
        from zero).  A bare tensor is keyed by WHAT IT VIEWS — storage, offset, shape, strides: a layer / model handle hands out
        fresh view objects of the same memory on every `trainable_weights` read (trainables._Handle), and those must find their
        moments again (keyed by id() they silently restarted from zero every step).  The entry holds the storage, so
        the memory cannot be freed and handed to another tensor while its moments exist (reset() drops them with a new fit())."""
        import weakref
        import torch
        if torch.is_tensor(owner):
            st = owner.untyped_storage()
            key = ('view', st._cdata, owner.storage_offset(), tuple(owner.shape), tuple(owner.stride()))
            ent = self._slots.get(key)
            if ent is None:
                ent = self._slots[key] = (st, torch.zeros_like(p), torch.zeros_like(p))
            return ent[1], ent[2]
        ent = self._slots.get(id(owner))
        if ent is not None and (ent[0]() is not owner or ent[1].shape != p.shape or ent[1].device != p.device):
            ent = None                  # the id was recycled by another object, or the variable was rebound to another shape / device
        if ent is None:
            try:
                ref = weakref.ref(owner)
            except TypeError:
                ref = (lambda o: (lambda: o))(owner)
            ent = self._slots[id(owner)] = (ref, torch.zeros_like(p), torch.zeros_like(p))
            for k in [k for k, e in self._slots.items() if not isinstance(k, tuple) and e[0]() is None]:
                del self._slots[k]
        return ent[1], ent[2]

    def lr_t(self, t):
        """Keras-Adam step size for the 1-based call counter t, in fp32 like optimizer_v2/adam.py."""
        f = np.float32
        return float(f(self.learning_rate) * np.sqrt(f(1.0) - np.power(f(self.beta_2), f(t))) / (f(1.0) - np.power(f(self.beta_1), f(t))))

    def apply_gradients(self, grads_and_vars):
        import torch
        from . import _lib
        self.iterations += 1
        alpha = self.lr_t(self.iterations)
        for g, var in grads_and_vars:
            p = getattr(var, 'tensor', var)
            if not (torch.is_tensor(p) and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise Exception(f'apply_gradients needs contiguous fp32 device arrays (drecpy_amd Variable or tensor), got {type(var).__name__}')
            g = torch.as_tensor(g, dtype=torch.float32, device=p.device).reshape(p.shape).contiguous()
            m, v = self._moments(var, p)
            if p.data_ptr() % 16 or p.numel() % 4:                       # drx_adam_dense moves float4: pad odd-sized variables
                n4 = (p.numel() + 3) // 4 * 4
                buf = torch.zeros(4, n4, dtype=torch.float32, device=p.device)
                for r, t in enumerate((p, m, v, g)):
                    buf[r, :p.numel()] = t.reshape(-1)
                _lib.check(_lib.lib().drx_adam_dense(_lib.ptr(buf[0]), _lib.ptr(buf[1]), _lib.ptr(buf[2]), _lib.ptr(buf[3]), n4, alpha, 0.0,
                                                     self.beta_1, self.beta_2, self.epsilon, _lib.stream_ptr(p.device)), 'drx_adam_dense')
                for r, t in enumerate((p, m, v)):
                    t.reshape(-1).copy_(buf[r, :p.numel()])
            else:
                _lib.check(_lib.lib().drx_adam_dense(_lib.ptr(p), _lib.ptr(m), _lib.ptr(v), _lib.ptr(g), p.numel(), alpha, 0.0, self.beta_1,
                                                     self.beta_2, self.epsilon, _lib.stream_ptr(p.device)), 'drx_adam_dense')


class Adagrad:
    kind = 'adagrad'

    def __init__(self, learning_rate=0.001, initial_accumulator_value=0.1, epsilon=1e-7):
        self.learning_rate, self.initial_accumulator_value, self.epsilon = float(learning_rate), float(initial_accumulator_value), float(epsilon)
        self.iterations = 0

    def apply_gradients(self, grads_and_vars):
        raise Exception('Adagrad is applied only inside the fused sparse CDAE step (mode="sampled"); use Adam for _update_weights')


class RowwiseAdagrad(Adagrad):
    kind = 'rowwise_adagrad'


ACCEPTED = "None, 'adam', 'adagrad', 'rowwise_adagrad', drecpy_amd.optimizers.Adam(...), drecpy_amd.optimizers.Adagrad(...)"


def resolve(optimizer, learning_rate):
    """fit()'s `optimizer=` argument -> one of the objects above; raises for anything the kernels do not implement."""
    if optimizer is None or optimizer == 'adam':
        return Adam(learning_rate)
    if optimizer == 'adagrad':
        return Adagrad(learning_rate)
    if optimizer == 'rowwise_adagrad':
        return RowwiseAdagrad(learning_rate)
    if isinstance(optimizer, (Adam, Adagrad)):
        return optimizer
    raise Exception(f'Unsupported optimizer {optimizer!r}: the update rules are hand-written HIP kernels, accepted values are {ACCEPTED}. '
                    f'(The reference takes any tf.keras optimizer, recommender_abc.py:155-156; TensorFlow objects cannot drive this engine.)')
