"""RecommenderABC — the training runtime of the reference (DRecPy/Recommender/recommender_abc.py) with the
TensorFlow tape replaced by fused HIP steps.

Public surface kept: __init__(verbose, log_file, interaction_threshold, seed), fit(...), predict, rank, recommend,
save/load, and the plugin hooks _pre_fit / _sample_batch / _predict_batch / _compute_batch_loss / _compute_reg_loss /
_predict / _rank / _recommend (recommender_abc.py:287-326, 385-389, 413-461).  Public methods take RAW ids, hooks take
INTERNAL ids (recommender_abc.py:27).

What changes underneath: a model provides `_do_batch(batch_samples, step, want_loss)` — one fused
forward/loss/backward/update step on the GPU (BASELINE.json names this hook) — instead of being differentiated by a
`tf.GradientTape` (recommender_abc.py:191-204).  One fit() "epoch" is still ONE mini-batch (recommender_abc.py:186-205).
Weights are snapshotted on the device only at epochs an early-stopping rule can choose (those where the epoch callback
ran) instead of deep-copied every step (recommender_abc.py:336-341).
"""
import logging
import pickle
import random
import threading
from abc import ABC, abstractmethod
from datetime import datetime
from heapq import nlargest

from .early_stopping import InvalidEpochValidationResultsException
from .loss_tracker import LossTracker


class RecommenderABC(ABC):
    def __init__(self, **kwds):
        self.verbose = kwds.get('verbose', True)
        self.min_interaction = None
        self.max_interaction = None
        self.seed = kwds.get('seed', None)
        self.fitted = False
        self.n_users = 0
        self.n_items = 0
        self.n_rows = 0
        self.interaction_threshold = kwds.get('interaction_threshold', 1e-3)
        self.interaction_dataset = None
        self.epoch_weights = {}
        self.optimizer = None
        self._loss_tracker = None
        self._rng = random.Random(self.seed)          # recommender_abc.py:74
        self._device_lock = threading.RLock()         # rank()/predict() are called from 4 threads by the evaluators
        self._logger = logging.getLogger(f'{self.__class__.__name__}_CLOGGER')
        self._logger.propagate = False
        self._logger.setLevel(logging.INFO)
        self._logger.handlers.clear()
        ch = logging.StreamHandler()
        ch.setFormatter(logging.Formatter('[%(asctime)s] (%(levelname)s) %(name)s: %(message)s'))
        ch.setLevel(logging.INFO)
        self._logger.addHandler(ch)
        self._file_logger = None
        if kwds.get('log_file', False):
            fh = logging.FileHandler(f'drecpy_{datetime.now().strftime("%Y-%m-%d_%H-%M-%S")}_{self.__class__.__name__}.log')
            fh.setLevel(logging.INFO)
            fh.setFormatter(logging.Formatter('[%(asctime)s] (%(levelname)s) %(name)s: %(message)s'))
            self._file_logger = logging.getLogger(f'{self.__class__.__name__}_FLOGGER')
            self._file_logger.propagate = False
            self._file_logger.setLevel(logging.INFO)
            self._file_logger.addHandler(fh)

    # ---- fit (recommender_abc.py:97-264) --------------------------------------------------------------
    def fit(self, interaction_dataset, epochs=50, batch_size=32, learning_rate=0.001, neg_ratio=5, reg_rate=0.001,
            copy_dataset=False, **kwds):
        self.interaction_dataset = interaction_dataset
        if copy_dataset:
            self._info('Cloning new dataset instance...')
            self.interaction_dataset = interaction_dataset.__copy__()
        self.interaction_dataset.assign_internal_ids()

        self.min_interaction = self.interaction_dataset.min('interaction')
        if self.min_interaction == 1:
            self.min_interaction = 0
        self.max_interaction = self.interaction_dataset.max('interaction')
        self.n_users = self.interaction_dataset.count_unique('uid')
        self.n_items = self.interaction_dataset.count_unique('iid')
        self.n_rows = len(self.interaction_dataset)
        self._loss_tracker = LossTracker()
        self._log_initial_info()

        self._info('Creating auxiliary structures...')
        self.learning_rate, self.reg_rate, self.batch_size = learning_rate, reg_rate, batch_size
        self._register_optimizer(kwds.get('optimizer', 'adam'))      # default Adam(lr), recommender_abc.py:153
        self._pre_fit(learning_rate, neg_ratio, reg_rate, **kwds)
        self.fitted = True

        if not hasattr(self, '_do_batch'):
            self._info('No trainable step (_do_batch) defined: skipping further model training.')
            return

        epoch_callback_fn = kwds.get('epoch_callback_fn', None)
        epoch_callback_ret, registered = None, True
        epoch_callback_freq = kwds.get('epoch_callback_freq', 5)
        rule = kwds.get('early_stopping_rule', None)
        rule_freq = kwds.get('early_stopping_freq', 5)
        best_epoch = None
        progress_desc = ''

        if self.verbose and epoch_callback_fn is not None:
            epoch_callback_ret = epoch_callback_fn(self)
            assert type(epoch_callback_ret) is dict, \
                f'The return type of the epoch_callback_fn should be dict, but found {type(epoch_callback_ret)}'
            for metric in epoch_callback_ret:
                self._loss_tracker.add_epoch_callback_result(metric, epoch_callback_ret[metric], 0)

        _iter = range(1, epochs + 1)
        bar = None
        if self.verbose:
            try:
                from tqdm import tqdm
                bar = _iter = tqdm(range(1, epochs + 1), total=epochs, desc='Fitting model...', position=0, leave=True)
            except ImportError:
                pass
        track = self.verbose or rule is not None
        e = 0
        for e in _iter:
            batch_samples = self._sample_batch(batch_size, **kwds)
            with self._device_lock:
                loss = self._do_batch(batch_samples, step=e - 1, want_loss=track, **kwds)
            callback_now = epoch_callback_fn is not None and e % epoch_callback_freq == 0
            if rule is not None and callback_now:
                self._store_epoch_weights(e)
            if track:
                loss = float(loss)                      # the only device->host sync of the loop
                self._loss_tracker.add_epoch_loss(loss)
                if callback_now:
                    registered = False
                    epoch_callback_ret = epoch_callback_fn(self)
                    assert isinstance(epoch_callback_ret, dict), \
                        f'The return type of the epoch_callback_fn should be dict, but found {type(epoch_callback_ret)}'
                progress_desc = f'Fitting model... Epoch {e} Loss: {loss:.4f}'
                if epoch_callback_ret is not None:
                    for metric in epoch_callback_ret:
                        progress_desc += f' | {metric}: {epoch_callback_ret[metric]}'
                        if not registered:
                            self._loss_tracker.add_epoch_callback_result(metric, epoch_callback_ret[metric], e)
                    registered = True
            if rule is not None and e % rule_freq == 0:
                try:
                    best_epoch = rule.compute(self._loss_tracker.epoch_losses, self._loss_tracker.epoch_callback_results,
                                              self._loss_tracker.called_epochs)
                    if rule.stop_training(e, best_epoch, epochs):
                        break
                except InvalidEpochValidationResultsException as ex:
                    self._warn(f'Failed to compute early stopping rule {rule.__class__.__name__}: {ex}')
            if best_epoch is not None:
                progress_desc += f' | {rule.__class__.__name__} best epoch: {best_epoch}'
            if bar is not None:
                bar.set_description(progress_desc)
            self._info(progress_desc, log_console=False)

        if rule is not None and e % rule_freq != 0:
            try:
                best_epoch = rule.compute(self._loss_tracker.epoch_losses, self._loss_tracker.epoch_callback_results,
                                          self._loss_tracker.called_epochs)
            except InvalidEpochValidationResultsException as ex:
                self._warn(f'Failed to compute early stopping rule {rule.__class__.__name__}: {ex}')
        if best_epoch is not None and best_epoch != epochs:
            self._info(f'Reverting network weights to epoch {best_epoch} due to the evaluation of the early stopping '
                       f'rule {rule.__class__.__name__}.')
            self._revert_weights(best_epoch)
        if self.verbose and kwds.get('plot', False):
            self._loss_tracker.display_graph(model_name=self.__class__.__name__,
                                             stopping_epoch=best_epoch if best_epoch != epochs else None)
        self._info('Model fitted.')

    def _register_optimizer(self, optimizer):
        self.optimizer = optimizer

    # ---- hooks ------------------------------------------------------------------------------------
    @abstractmethod
    def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
        pass

    @abstractmethod
    def _sample_batch(self, batch_size, **kwds):
        pass

    @abstractmethod
    def _predict_batch(self, batch_samples, **kwds):
        pass

    @abstractmethod
    def _compute_batch_loss(self, predictions, desired_values, **kwds):
        pass

    def _compute_reg_loss(self, reg_rate, batch_size, trainable_models, trainable_layers, trainable_weights, **kwds):
        return 0

    @abstractmethod
    def _predict(self, uid, iid, **kwds):
        pass

    # ---- snapshots (recommender_abc.py:336-352) -----------------------------------------------------
    def _store_epoch_weights(self, epoch):
        if hasattr(self, '_engine'):
            self.epoch_weights[epoch] = self._engine.snapshot()

    def _revert_weights(self, epoch):
        if epoch == 0 or epoch not in self.epoch_weights:
            self._warn(f'No stored weights for epoch {epoch}; keeping the current ones.')
            return
        self._engine.restore(self.epoch_weights[epoch])
        self._info(f'Network weights reverted to epoch {epoch}.')

    # ---- front-ends (recommender_abc.py:354-461) ------------------------------------------------------
    def predict(self, user_id, item_id, skip_errors=False, **kwds):
        assert self.fitted is True, 'The model requires to be fitted before being able to make predictions.'
        assert skip_errors or self.interaction_dataset.user_to_uid(user_id) is not None, f'User {user_id} was not found.'
        assert skip_errors or self.interaction_dataset.item_to_iid(item_id) is not None, f'Item {item_id} was not found.'
        prediction = None
        try:
            uid = self.interaction_dataset.user_to_uid(user_id)
            iid = self.interaction_dataset.item_to_iid(item_id)
            prediction = self._predict(uid, iid, **kwds)
            if prediction is None:
                raise Exception(f'Failed to predict(user_id={user_id}, item_id={item_id}): None was returned.')
        except Exception as e:
            if not skip_errors:
                raise e
        return prediction

    def recommend(self, user_id, n=None, novelty=True, interaction_threshold=None, **kwds):
        assert self.fitted is True, 'The model requires to be fitted before being able to make predictions.'
        assert self.interaction_dataset.user_to_uid(user_id) is not None, f'User {user_id} was not found.'
        if n is None:
            n = self.n_items
        uid = self.interaction_dataset.user_to_uid(user_id)
        recs = self._recommend(uid, n, novelty, interaction_threshold)
        return [(r, self.interaction_dataset.iid_to_item(iid)) for r, iid in recs]

    def _recommend(self, uid, n, novelty, threshold):
        ranked = self._rank(uid, range(0, self.n_items), n, novelty)
        if threshold is None:
            return ranked
        return list(filter(lambda x: x[0] >= threshold, ranked))

    def rank(self, user_id, item_ids, novelty=True, skip_invalid_items=True, **kwds):
        assert self.fitted is True, 'The model requires to be fitted before being able to make predictions.'
        assert self.interaction_dataset.user_to_uid(user_id) is not None, f'User {user_id} was not found.'
        uid = self.interaction_dataset.user_to_uid(user_id)
        iids = []
        for item_id in item_ids:
            iid = self.interaction_dataset.item_to_iid(item_id)
            if iid is not None:
                iids.append(iid)
            elif not skip_invalid_items:
                raise Exception(f'Item {item_id} was not found.')
        n = kwds.get('n', len(iids))
        assert n <= len(iids), \
            f'The number of best items to return must be <= len(item_ids) (current value is {n} > {len(iids)})'
        ranked = self._rank(uid, iids, n, novelty)
        return [(r, self.interaction_dataset.iid_to_item(iid)) for r, iid in ranked]

    def _rank(self, uid, iids, n, novelty):
        if novelty:
            rated = self.interaction_dataset.select(f'uid == {uid}').values_list('iid', to_list=True)
            iids = set(iids).difference(set(rated))
        preds = filter(lambda x: x[0] is not None, [(self._predict(uid, iid), iid) for iid in iids])
        return nlargest(n, preds)

    def _standardize_value(self, value):
        return (value - self.min_interaction) / (self.max_interaction - self.min_interaction)

    def _rescale_value(self, value):
        return self.min_interaction + (self.max_interaction - self.min_interaction) * value

    # ---- logging ------------------------------------------------------------------------------------
    def _log_initial_info(self):
        self._info(f'Max. interaction value: {self.max_interaction}')
        self._info(f'Min. interaction value: {self.min_interaction}')
        self._info(f'Interaction threshold value: {self.interaction_threshold}')
        self._info(f'Number of unique users: {self.n_users}')
        self._info(f'Number of unique items: {self.n_items}')
        self._info(f'Number of training points: {self.n_rows}')
        sparsity = round(100 * (1 - (self.n_rows / (self.n_users * self.n_items))), 4)
        self._info(f'Sparsity level: approx. {sparsity}%')

    def _emit(self, level, msg, log_console, log_file):
        if not self.verbose:
            return
        if log_console:
            getattr(self._logger, level)(msg)
        if self._file_logger and log_file:
            getattr(self._file_logger, level)(msg)

    def _info(self, msg, log_console=True, log_file=True):
        self._emit('info', msg, log_console, log_file)

    def _warn(self, msg, log_console=True, log_file=True):
        self._emit('warning', msg, log_console, log_file)

    def _error(self, msg, log_console=True, log_file=True):
        self._emit('error', msg, log_console, log_file)

    # ---- save / load (recommender_abc.py:503-524: joblib of the whole object) --------------------------
    def save(self, save_path):
        state = {k: v for k, v in self.__dict__.items()
                 if k not in ('_engine', '_logger', '_file_logger', '_device_lock', '_sampler', '_mask_rng', 'epoch_weights')}
        if hasattr(self, '_engine'):
            state['__params__'] = self._engine.get_params()
        with open(save_path, 'wb') as f:
            pickle.dump({'cls': self.__class__, 'state': state}, f)

    @staticmethod
    def load(load_path):
        with open(load_path, 'rb') as f:
            blob = pickle.load(f)
        obj = blob['cls'].__new__(blob['cls'])
        RecommenderABC.__init__(obj, verbose=blob['state'].get('verbose', True), seed=blob['state'].get('seed'))
        params = blob['state'].pop('__params__', None)
        obj.__dict__.update(blob['state'])
        if params is not None:
            obj._restore_engine(params)
        return obj
