"""ModifiedDMF — the twin of DRecPy's examples/extending_recommender_dmf.py (BASELINE.json config 3): DMF extended with one
registered scalar that multiplies every prediction.

Reference (its lines in parentheses)                       here
  self._extra_weight = tf.Variable([1.])            (11)   Variable([1.])             — a device array handle
  self._register_trainable(self._extra_weight)      (12)   the same call             — lands in trainable_weights, so its Adam apply
                                                           comes FIRST each step: t = 3*step + 1, user_nn 3*step + 2, item_nn 3*step + 3
  predictions = [w * pred for pred in predictions]  (16)   the same expression in _predict_batch (inference); for TRAINING the engine
                                                           has no tape to see that expression, so the model tells the fused step what it
                                                           means with `bind_prediction_scale`: predictions are multiplied by the variable
                                                           and — because a LIST of (1,)-tensors reaches Keras' BCE as (B,1) against (B,)
                                                           targets — the loss is the (B,B) broadcast = BCE against the batch-mean target.

    python examples/extending_recommender_dmf.py [--movielens /data/ml-1m] [--epochs 200]"""
from _common import arguments, split, stopwatch

from drecpy_amd.Recommender import DMF, Variable


class ModifiedDMF(DMF):
    def __init__(self, **kwds):
        super(ModifiedDMF, self).__init__(**kwds)

    def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
        super(ModifiedDMF, self)._pre_fit(learning_rate, neg_ratio, reg_rate, **kwds)
        self._extra_weight = Variable([1.], name='extra_weight', device=self.device)
        self._register_trainable(self._extra_weight)
        self._engine.bind_prediction_scale(self._extra_weight, broadcast_targets=True)
        if kwds.get('initial_weights') is not None and 'extra_w' in kwds['initial_weights']:
            self._extra_weight.assign(kwds['initial_weights']['extra_w'])

    def _predict_batch(self, batch_samples, **kwds):
        predictions, desired_values = super(ModifiedDMF, self)._predict_batch(batch_samples, **kwds)
        predictions = list(self._extra_weight * predictions.reshape(-1, 1))      # [(w * pred) for pred in predictions], one launch
        return predictions, desired_values


def main():
    from drecpy_amd.Evaluation import ranking_evaluation
    args = arguments(default_epochs=200, dataset_name='ml-1m')
    train, test = split(args, 'ml-1m')
    model = ModifiedDMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=not args.quiet)
    with stopwatch(f'fit, {args.epochs} epochs of 256'):
        model.fit(train, epochs=args.epochs, batch_size=256, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
    print(f'  extra weight after training: {float(model._extra_weight.numpy()[0]):.6f}')
    scores = ranking_evaluation(model, test, k=[1, 5, 10], novelty=True, n_test_users=100, n_pos_interactions=1,
                                n_neg_interactions=100, generate_negative_pairs=True, seed=10, verbose=False)
    for name, value in scores.items():
        print(f'  {name:14s} {value}')


if __name__ == '__main__':
    main()
