"""The registration half of the plugin surface (DRecPy/Recommender/recommender_abc.py:66-69, 153-156, 266-285, 328-334) on the
host: lists, ordering, error sentences, the optimizer argument.  No GPU needed (handles over CPU arrays)."""
import numpy as np
import pytest

from drecpy_amd import optimizers
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import RecommenderABC, TrainableLayer, TrainableModel, Variable


def _dataset():
    rng = np.random.default_rng(0)
    return InteractionDataset.read_df({'user': rng.integers(0, 9, 60), 'item': rng.integers(0, 12, 60), 'interaction': rng.integers(1, 6, 60)},
                                      verbose=False)


class _Hooks(RecommenderABC):
    """A reference-style model: only the hooks of recommender_abc.py:287-312, no fused step."""
    registers = ()

    def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
        self._register_trainables(list(self.registers))

    def _sample_batch(self, batch_size, **kwds):
        return []

    def _predict_batch(self, batch_samples, **kwds):
        return [], []

    def _compute_batch_loss(self, predictions, desired_values, **kwds):
        return 0.0

    def _predict(self, uid, iid, **kwds):
        return 1.0


def test_no_trainables_means_no_training_like_the_reference():
    m = _Hooks(verbose=False)
    m.fit(_dataset(), epochs=3)                      # recommender_abc.py:159-163: info + return, the model can predict
    assert m.fitted and m.trainable_weights == [] and m.trainable_layers == [] and m.trainable_models == []
    assert isinstance(m.optimizer, optimizers.Adam) and m.optimizer.learning_rate == 0.001      # default optimizer (:153)


def test_a_hooks_only_model_whose_loss_is_not_differentiable_fails_with_a_sentence_not_silently():
    """No _do_batch: the generic tape step (torch.autograd in tf.GradientTape's place) runs the hooks — a loss that is a plain
    number cannot be differentiated w.r.t. the registered variable, and fit() says so."""
    class Tape(_Hooks):
        registers = (Variable([1.], device='cpu'),)
    m = Tape(verbose=False)
    with pytest.raises(NotImplementedError, match='_do_batch'):
        m.fit(_dataset(), epochs=3)


def test_register_trainable_types_lists_and_apply_order():
    m = _Hooks(verbose=False)
    with pytest.raises(Exception, match='Cannot register None as a trainable variable.'):
        m._register_trainable(None)
    with pytest.raises(Exception, match='Invalid trainable variable 3'):
        m._register_trainable(3)
    model, layer, var = TrainableModel('towers', lambda: []), TrainableLayer('emb', lambda: []), Variable([1., 2.], device='cpu')
    m._register_trainables([model, layer])
    m._register_trainable(var)                       # registered LAST ...
    assert (m.trainable_models, m.trainable_layers, m.trainable_weights) == ([model], [layer], [var])
    assert m._apply_order() == [var, layer, model]   # ... applied FIRST: weights + layers + models (recommender_abc.py:194-196)
    assert [m._apply_position(x) for x in (var, layer, model)] == [0, 1, 2]
    assert var.shape == (2,) and np.array_equal((var * 2).numpy(), [2., 4.])


def test_a_second_fit_starts_with_fresh_lists():
    class Own(_Hooks):
        def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
            self.v = Variable([0.5], device='cpu')
            self._register_trainable(self.v)

        def _do_batch(self, batch_samples, step=0, want_loss=False, **kwds):
            return 0.0
    m = Own(verbose=False)
    m.fit(_dataset(), epochs=2)
    m.fit(_dataset(), epochs=2)
    assert len(m.trainable_weights) == 1 and m.epoch_weights == {}


def test_optimizer_argument_is_resolved_after_pre_fit_and_unknown_objects_are_rejected():
    seen = {}

    class Own(_Hooks):
        def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
            seen['during_pre_fit'] = self.optimizer
            self._register_optimizer(optimizers.Adam(learning_rate=7.0))      # a model-specific optimizer ...

        def _do_batch(self, batch_samples, step=0, want_loss=False, **kwds):
            return 0.0
    m = Own(verbose=False)
    forced = optimizers.Adam(learning_rate=0.05, beta_1=0.8)
    m.fit(_dataset(), epochs=1, learning_rate=0.02, optimizer=forced)
    assert isinstance(seen['during_pre_fit'], optimizers.Adam) and seen['during_pre_fit'].learning_rate == 0.02
    assert m.optimizer is forced                     # ... is overridden by fit(optimizer=...) (recommender_abc.py:155-156)
    m.fit(_dataset(), epochs=1)
    assert m.optimizer.learning_rate == 7.0          # without the argument the model's own stays
    assert optimizers.resolve('adagrad', 0.1).kind == 'adagrad' and optimizers.resolve(None, 0.1).kind == 'adam'
    with pytest.raises(Exception, match='Unsupported optimizer'):
        m.fit(_dataset(), epochs=1, optimizer=object())
    assert abs(optimizers.Adam(1e-3).lr_t(1) - 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)) < 1e-8          # fp32 like Keras
