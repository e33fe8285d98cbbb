"""RecommenderABC — the training runtime of the reference (DRecPy/Recommender/recommender_abc.py) with the TensorFlow
tape replaced by fused HIP steps.

Public surface kept: __init__(verbose, log_file, interaction_threshold, seed), fit(...), predict, rank, recommend,
save/load, and the plugin hooks _pre_fit / _sample_batch / _predict_batch / _compute_batch_loss / _compute_reg_loss /
_predict / _rank / _recommend (recommender_abc.py:287-326, 385-389, 413-461).  Public methods take RAW ids, hooks take
INTERNAL ids (recommender_abc.py:27).

What changes underneath: a model provides `_do_batch(batch_samples, step, want_loss)` — one fused
forward/loss/backward/update step on the GPU (BASELINE.json names this hook) — instead of being differentiated by a
`tf.GradientTape` (recommender_abc.py:191-204).  One fit() "epoch" is still ONE mini-batch (recommender_abc.py:186-205).

The registration half of the surface is kept too (recommender_abc.py:66-69, 266-285, 328-334): `_register_trainable(s)`
fill `trainable_weights / trainable_layers / trainable_models` with HANDLES over device arrays (trainables.py) instead of
TensorFlow objects; their concatenation, weights first (:194-196), is the order of the per-step optimizer applies and fixes
each variable's Adam counter.  `_update_weights(gradients, trainable_weights)` applies gradients a model computed itself
with the registered optimizer (optimizers.py) on the device.  A model that defines NO `_do_batch` — only the reference's
hooks `_predict_batch` + `_compute_batch_loss` (+ `_compute_reg_loss`) — is trained by the GENERIC TAPE STEP
(`_tape_do_batch`): recommender_abc.py:186-205 with torch.autograd standing where tf.GradientTape stands, the hooks
written with torch operations on `Variable.tensor` instead of TensorFlow ones, the update still one `apply_gradients` per
registered item through the library's Adam kernel.  CDAE / DMF / Caser never take it (their steps are the fused HIP
kernels); a hook that returns a loss the registered variables cannot be differentiated through fails with a sentence
(NotImplementedError) instead of returning an untrained model.
Weights are snapshotted on the device only at epochs an early-stopping rule can choose (those where the epoch callback
ran) instead of deep-copied every step (recommender_abc.py:336-341).  The loss is read back from the device only when it
is logged or an early-stopping rule needs it.
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

_LOG_FORMAT = '[%(asctime)s] (%(levelname)s) %(name)s: %(message)s'
_UNPICKLED = ('_engine', '_logger', '_file_logger', '_device_lock', '_sampler', '_mask_rng', '_mask_rngs', '_drawahead', '_L', 'epoch_weights', '_pipeline', '_pending',
              '_host_pool', '_dist_model', 'optimizer', 'trainable_vars', 'trainable_weights', 'trainable_layers', 'trainable_models',
              '_extra_weight', 'W', 'W_', 'V', 'b', 'b_', 'user_nn', 'item_nn', '_layers',
              '_ahead', '_dev_side', '_dev_next', '_dev_done')             # (run-ahead state of fit(device_sampler=True): streams, events, device batches)


def _make_logger(name, handler):
    logger = logging.getLogger(name)
    logger.propagate = False
    logger.setLevel(logging.INFO)
    logger.handlers.clear()
    handler.setLevel(logging.INFO)
    handler.setFormatter(logging.Formatter(_LOG_FORMAT))
    logger.addHandler(handler)
    return logger


class _FitMonitor:
    """Progress text, epoch callbacks and early stopping of one fit() run (recommender_abc.py:170-256): host logic."""

    def __init__(self, model, epochs, kwds):
        self.model, self.epochs = model, epochs
        self.callback = kwds.get('epoch_callback_fn')
        self.callback_freq = kwds.get('epoch_callback_freq', 5)
        self.rule = kwds.get('early_stopping_rule')
        self.rule_freq = kwds.get('early_stopping_freq', 5)
        self.best_epoch = None
        self.last_metrics = None
        self.tracker = model._loss_tracker
        self.needs_loss = model.verbose or self.rule is not None

    def _run_callback(self, epoch):
        res = self.callback(self.model)
        assert isinstance(res, dict), f'The return type of the epoch_callback_fn should be dict, but found {type(res)}'
        for name, value in res.items():
            self.tracker.add_epoch_callback_result(name, value, epoch)
        self.last_metrics = res

    def before_training(self):
        if self.model.verbose and self.callback is not None:
            self._run_callback(0)

    def callback_due(self, epoch):
        return self.callback is not None and epoch % self.callback_freq == 0

    def _evaluate_rule(self):
        try:
            self.best_epoch = self.rule.compute(self.tracker.epoch_losses, self.tracker.epoch_callback_results,
                                                self.tracker.called_epochs)
            return True
        except InvalidEpochValidationResultsException as err:
            self.model._warn(f'Failed to compute early stopping rule {type(self.rule).__name__}: {err}')
            return False

    def after_step(self, epoch, loss):
        """Returns (stop?, progress text)."""
        text = ''
        if self.needs_loss:
            self.tracker.add_epoch_loss(loss)
            if self.callback_due(epoch):
                self._run_callback(epoch)
            text = f'Fitting model... Epoch {epoch} Loss: {loss:.4f}'
            for name, value in (self.last_metrics or {}).items():
                text += f' | {name}: {value}'
        stop = False
        if self.rule is not None and epoch % self.rule_freq == 0 and self._evaluate_rule():
            stop = bool(self.rule.stop_training(epoch, self.best_epoch, self.epochs))
        if self.best_epoch is not None:
            text += f' | {type(self.rule).__name__} best epoch: {self.best_epoch}'
        return stop, text

    def after_training(self, last_epoch):
        if self.rule is not None and last_epoch % self.rule_freq != 0:
            self._evaluate_rule()
        return self.best_epoch


class RecommenderABC(ABC):
    def __init__(self, **kwds):
        self.verbose = kwds.get('verbose', True)
        self.seed = kwds.get('seed', None)
        self.interaction_threshold = kwds.get('interaction_threshold', 1e-3)
        self.min_interaction = self.max_interaction = None
        self.n_users = self.n_items = self.n_rows = 0
        self.interaction_dataset = None
        self.fitted = False
        self.optimizer = None
        self.trainable_vars = []                      # recommender_abc.py:66-69
        self.trainable_weights, self.trainable_layers, self.trainable_models = [], [], []
        self.epoch_weights = {}                       # epoch -> device snapshot (only callback epochs)
        self._loss_tracker = None
        self._rng = random.Random(self.seed)          # recommender_abc.py:74
        self._device_lock = threading.RLock()         # rank()/predict() may be called from evaluator threads
        cls = type(self).__name__
        self._logger = _make_logger(f'{cls}_CLOGGER', logging.StreamHandler())
        self._file_logger = None
        if kwds.get('log_file', False):
            stamp = datetime.now().strftime('%Y-%m-%d_%H-%M-%S')
            self._file_logger = _make_logger(f'{cls}_FLOGGER', logging.FileHandler(f'drecpy_{stamp}_{cls}.log'))

    # ---- fit (recommender_abc.py:97-264) --------------------------------------------------------------
    def _bind_dataset(self, interaction_dataset, copy_dataset):
        """Id assignment and the dataset statistics the models read (recommender_abc.py:133-146)."""
        ds = interaction_dataset
        if copy_dataset:
            self._info('Cloning new dataset instance...')
            ds = interaction_dataset.__copy__()
        ds.assign_internal_ids()
        self.interaction_dataset = ds
        lo = ds.min('interaction')
        self.min_interaction = 0 if lo == 1 else lo          # quirk kept: a minimum of 1 is treated as 0
        self.max_interaction = ds.max('interaction')
        self.n_users, self.n_items, self.n_rows = ds.count_unique('uid'), ds.count_unique('iid'), len(ds)

    def _reset_fit_state(self):
        """A second fit() of the same object starts clean: per-dataset caches of the models (user -> items tables, sequences,
        top-k helpers), snapshots of the previous run and the registration lists (the reference appends to them again on every
        fit(), recommender_abc.py:66-69 are only initialised in __init__ — duplicates there, a fresh list here)."""
        for name in ('_user_items', '_ui_sorted', '_ui_ptr', '_seq_ptr', '_seq_items', '_topk_helper'):
            self.__dict__.pop(name, None)
        self.epoch_weights = {}
        self.trainable_vars = []
        self.trainable_weights, self.trainable_layers, self.trainable_models = [], [], []

    def fit(self, interaction_dataset, epochs=50, batch_size=32, learning_rate=0.001, neg_ratio=5, reg_rate=0.001,
            copy_dataset=False, **kwds):
        from .. import optimizers
        self._reset_fit_state()
        self._bind_dataset(interaction_dataset, copy_dataset)
        self._loss_tracker = LossTracker()
        self._log_initial_info()
        self._info('Creating auxiliary structures...')
        self.learning_rate, self.reg_rate, self.batch_size = learning_rate, reg_rate, batch_size
        self._register_optimizer(optimizers.Adam(learning_rate=learning_rate))   # default optimizer, recommender_abc.py:153
        self._pre_fit(learning_rate, neg_ratio, reg_rate, **kwds)
        self._optimizer_forced = kwds.get('optimizer', None) is not None
        if self._optimizer_forced:                                   # allow forcing a custom optimizer (:155-156): AFTER _pre_fit
            self._register_optimizer(optimizers.resolve(kwds.get('optimizer'), learning_rate))
        self._configure_optimizer()
        self.fitted = True  # should be able to make predictions after pre fit
        registered = self.trainable_weights + self.trainable_layers + self.trainable_models
        do_batch = getattr(self, '_do_batch', None)
        if do_batch is None:
            if len(registered) == 0:
                self._info('No trainable vars found: skipping further model training. If this is non-intentional, please '
                           'use self._register_trainable or self._register_trainables to register variables that are '
                           'subject to weight updates.')
                return
            do_batch = self._tape_do_batch          # a reference-style model: hooks only (recommender_abc.py:186-205)
        fused = self._fused_trainables()
        if fused is not None:
            stray = [t for t in registered if not any(t is f for f in fused)]
            if stray:
                raise NotImplementedError(
                    f'{type(self).__name__}: {stray} registered through _register_trainable(s) but the fused step of this model does '
                    f'not update them (no autodiff tape here). Bind them to the engine (e.g. DmfEngine.bind_prediction_scale) or '
                    f'update them in your own _do_batch with self._update_weights(gradients, variables).')
        self._info(f'Number of registered trainable variables: {len(registered)}')

        monitor = _FitMonitor(self, epochs, kwds)
        monitor.before_training()
        bar = self._progress_bar(epochs)
        epoch, ahead = 0, None
        if getattr(self, '_host_pool', None) is not None:      # (left behind by a fit() that raised: its worker may hold an unclaimed job)
            self._host_pool.close()
            self._host_pool = None
        if not monitor.needs_loss and bar is None and self._file_logger is None:
            # nothing observes single steps (no loss, callback, rule or log line per epoch): a model may run the loop natively
            epoch = int(self._run_steps(0, epochs, batch_size, **kwds) or 0)
        for epoch in (bar if bar is not None else range(epoch + 1, epochs + 1)):
            # The host work of the NEXT batch overlaps this batch's device step — never beyond the last epoch, so the sampler
            # streams end exactly where the reference's do.  Models whose _sample_batch is independent of _do_batch opt in with
            # `_host_prefetch` (the draw runs on a worker thread; ctypes calls release the GIL); CDAE handles `more_to_come`
            # itself (its corruption stream belongs to the draw).
            # From `_prefetch_from` samples per batch on: below it the draw is a few tens of microseconds, and two threads that both run
            # Python hand the interpreter lock back and forth through futex wake-ups — DMF at B = 256: 0.19 - 0.25 ms per step with the
            # worker, 0.125 inline (r06, profiles/r06_host_handover.log).
            if getattr(self, '_host_prefetch', False) and batch_size >= getattr(self, '_prefetch_from', 0) and self._prefetch_mode != 'inline':
                batch = ahead.result() if ahead is not None else self._sample_batch(batch_size, **kwds)
                ahead = self._prefetch_pool().submit(self._sample_batch, batch_size, **kwds) if epoch < epochs else None
            else:
                batch = self._sample_batch(batch_size, more_to_come=epoch < epochs, batches_after=epochs - epoch, **kwds)
            with self._device_lock:
                loss = do_batch(batch, step=epoch - 1, want_loss=monitor.needs_loss, **kwds)
            if monitor.rule is not None and monitor.callback_due(epoch):
                self._store_epoch_weights(epoch)
            stop, text = monitor.after_step(epoch, float(loss) if monitor.needs_loss else None)
            if bar is not None:
                bar.set_description(text)
            self._info(text, log_console=False)
            if stop:
                if ahead is not None:
                    ahead.result()                 # drawn but unused: an early stop leaves the sampler one batch further
                break
        if getattr(self, '_host_pool', None) is not None:      # (the prefetch worker of this fit(): the next fit() starts its own)
            self._host_pool.close()
            self._host_pool = None
        best = monitor.after_training(epoch)
        if best is not None and best != epochs:
            self._info(f'Reverting network weights to epoch {best} due to the evaluation of the early stopping rule '
                       f'{type(monitor.rule).__name__}.')
            self._revert_weights(best)
        if self.verbose and kwds.get('plot', False):
            self._loss_tracker.display_graph(model_name=type(self).__name__, stopping_epoch=None if best == epochs else best)
        self._info('Model fitted.')

    def _progress_bar(self, epochs):
        if not self.verbose:
            return None
        try:
            from tqdm import tqdm
        except ImportError:
            return None
        return tqdm(range(1, epochs + 1), total=epochs, desc='Fitting model...', position=0, leave=True)

    # ---- registration (recommender_abc.py:266-285) ----------------------------------------------------
    def _register_trainable(self, variable):
        from .trainables import TrainableLayer, TrainableModel, Variable
        if variable is None:
            raise Exception('Cannot register None as a trainable variable.')
        if isinstance(variable, TrainableModel):
            self.trainable_models.append(variable)
        elif isinstance(variable, TrainableLayer):
            self.trainable_layers.append(variable)
        elif isinstance(variable, Variable):
            self.trainable_weights.append(variable)
        else:
            raise Exception(f'Invalid trainable variable {variable}. The supported types are: drecpy_amd.Recommender.TrainableModel, '
                            f'TrainableLayer and Variable (the device-side stand-ins for tf.keras.models.Model, '
                            f'tf.keras.layers.Layer and tf.Variable; TensorFlow objects cannot be differentiated by this engine).')

    def _register_trainables(self, variables):
        for variable in variables:
            self._register_trainable(variable)

    def _register_optimizer(self, optimizer):
        self.optimizer = optimizer
        if hasattr(optimizer, 'reset'):       # an optimizer object reused for another fit() starts over: call counter and moments
            optimizer.reset()

    def _configure_optimizer(self):
        """Called once per fit() after _pre_fit and the `optimizer=` override: models push the registered optimizer's kind and
        hyper-parameters into their engine (and reject kinds their fused step does not implement)."""

    def _run_steps(self, first_step, n_steps, batch_size, **kwds):
        """Optional: run steps first_step .. first_step + n_steps - 1 of fit() without returning to this loop in between (called
        only when nothing observes single steps).  Returns how many were done; the per-step loop does the rest."""
        return 0

    def _fused_trainables(self):
        """The handles a model's fused _do_batch updates, or None when the model manages its variables itself."""
        return None

    def _apply_order(self):
        """Registered items in the order of the reference's per-step apply_gradients calls (recommender_abc.py:194-196)."""
        return self.trainable_weights + self.trainable_layers + self.trainable_models

    def _apply_position(self, handle):
        for j, t in enumerate(self._apply_order()):
            if t is handle:
                return j
        raise Exception(f'{handle} is not registered (self._register_trainable)')

    def _tape_do_batch(self, batch_samples, step=0, want_loss=False, **kwds):
        """The reference's training step (recommender_abc.py:186-205) for models that define only its hooks: predictions and loss
        under a tape, the gradient of loss + regularisation w.r.t. every registered item, one apply_gradients per item.  The tape is
        torch.autograd on the device: `_predict_batch` / `_compute_batch_loss` / `_compute_reg_loss` must build the loss with torch
        operations on the registered arrays (`Variable.tensor`, or the tensors a layer / model handle returns — the SAME tensor
        objects on every read).  The built-in models do not come here."""
        import torch
        items = self._apply_order()
        groups = []
        for it in items:
            ts = list(it.trainable_weights) if hasattr(it, 'trainable_weights') else [getattr(it, 'tensor', it)]
            if hasattr(it, 'trainable_weights') and any(a is not b for a, b in zip(ts, it.trainable_weights)):
                raise NotImplementedError(f'{it}: the generic tape step needs a handle whose trainable_weights are the same tensor objects '
                                          f'on every read (views made anew per call cannot be tracked); implement _do_batch instead.')
            for t in ts:
                t.requires_grad_(True)
            groups.append(ts)
        try:
            predictions, desired_values = self._predict_batch(batch_samples, **kwds)
            loss = self._compute_batch_loss(predictions, desired_values, **kwds)
            # (recommender_abc.py:200: the reference's five positional arguments and no keyword ones — a subclass written against it
            # may define _compute_reg_loss without **kwds)
            loss = loss + self._compute_reg_loss(self.reg_rate, len(batch_samples), self.trainable_models, self.trainable_layers,
                                                 self.trainable_weights)
            if not (torch.is_tensor(loss) and loss.requires_grad):
                raise NotImplementedError(
                    f'{type(self).__name__} defines no _do_batch and the loss its hooks return ({type(loss).__name__}) cannot be differentiated '
                    f'w.r.t. the {len(items)} registered trainable(s): build it with torch operations on the registered arrays in '
                    f'_predict_batch / _compute_batch_loss (the generic tape step), or implement the fused step '
                    f'_do_batch(batch_samples, step, want_loss, **kwds) (see INTEGRATION.md), or derive from CDAE / DMF / Caser.')
            flat = [t for ts in groups for t in ts]
            grads = torch.autograd.grad(loss.reshape(()), flat, allow_unused=True)
        finally:
            for ts in groups:
                for t in ts:
                    t.requires_grad_(False)
        grads = [g if g is not None else torch.zeros_like(t) for g, t in zip(grads, flat)]
        per_item, at = [], 0
        for ts in groups:
            per_item.append(list(grads[at:at + len(ts)]))
            at += len(ts)
        self._update_weights(per_item, items)
        return float(loss.detach().item()) if want_loss else None

    def _update_weights(self, gradients, trainable_weights):
        """One optimizer.apply_gradients per registered item (recommender_abc.py:328-334), on the device: `gradients[j]` is a
        device array (or a list of them) matching `trainable_weights[j]` — a Variable, or the trainable_weights list of a
        layer / model handle.  The fused steps of CDAE / DMF / Caser do this inside their kernels; this is for models that
        compute gradients themselves in `_do_batch`."""
        import torch
        for gradient, trainable_var in zip(gradients, trainable_weights):
            if torch.is_tensor(gradient):
                gradient = [gradient]
            if hasattr(trainable_var, 'trainable_weights'):             # a layer / model handle: its arrays
                trainable_var = trainable_var.trainable_weights
            elif not isinstance(trainable_var, (list, tuple)):           # a Variable (or a bare device array)
                trainable_var = [trainable_var]
            self.optimizer.apply_gradients(zip(gradient, trainable_var))

    # ---- hooks ------------------------------------------------------------------------------------
    @abstractmethod
    def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
        """Build the model state for `self.interaction_dataset` (tables on the device, sampler)."""

    _prefetch_mode = 'spin'       # who draws the next batch of a `_host_prefetch` model: 'spin' (a worker with polled hand-overs),
                                  # 'pool' (concurrent.futures: the r02 - r05 form) or 'inline' — an attribute, not an environment switch

    def _prefetch_pool(self):
        if getattr(self, '_host_pool', None) is None:
            if self._prefetch_mode == 'pool':          # (A/B: concurrent.futures, r02 - r05)
                from concurrent.futures import ThreadPoolExecutor

                class _Pool(ThreadPoolExecutor):
                    def close(self):
                        self.shutdown(wait=True)
                self._host_pool = _Pool(max_workers=1)
            else:
                from .._spinpool import SpinWorker        # (one worker; hand-overs polled without the interpreter lock)
                self._host_pool = SpinWorker()
        return self._host_pool

    @abstractmethod
    def _sample_batch(self, batch_size, **kwds):
        """Sample the data points of one step."""

    @abstractmethod
    def _predict_batch(self, batch_samples, **kwds):
        """(predictions, desired values) of a sampled batch."""

    @abstractmethod
    def _compute_batch_loss(self, predictions, desired_values, **kwds):
        """Prediction loss of a batch."""

    def _compute_reg_loss(self, reg_rate, batch_size, trainable_models, trainable_layers, trainable_weights, **kwds):
        return 0

    @abstractmethod
    def _predict(self, uid, iid, **kwds):
        """Predicted interaction of internal ids (uid, iid)."""

    # ---- snapshots (recommender_abc.py:336-352) -----------------------------------------------------
    def _store_epoch_weights(self, epoch):
        if hasattr(self, '_engine'):
            self.epoch_weights[epoch] = self._engine.snapshot()

    def _revert_weights(self, epoch):
        snap = self.epoch_weights.get(epoch)
        if snap is None:
            self._warn(f'No stored weights for epoch {epoch}; keeping the current ones.')
            return
        self._engine.restore(snap)
        self._info(f'Network weights reverted to epoch {epoch}.')

    # ---- front-ends (recommender_abc.py:354-461): raw ids in, raw ids out ------------------------------------
    def _require_user(self, user_id):
        assert self.fitted is True, 'The model requires to be fitted before being able to make predictions.'
        uid = self.interaction_dataset.user_to_uid(user_id)
        assert uid is not None, f'User {user_id} was not found.'
        return uid

    def predict(self, user_id, item_id, skip_errors=False, **kwds):
        assert self.fitted is True, 'The model requires to be fitted before being able to make predictions.'
        ds = self.interaction_dataset
        try:
            uid, iid = ds.user_to_uid(user_id), ds.item_to_iid(item_id)
        except Exception:
            if skip_errors:
                return None
            raise
        if not skip_errors:
            assert uid is not None, f'User {user_id} was not found.'
            assert iid is not None, f'Item {item_id} was not found.'
        try:
            value = self._predict(uid, iid, **kwds)
            if value is None:
                raise Exception(f'Failed to predict(user_id={user_id}, item_id={item_id}): None was returned.')
            return value
        except Exception:
            if skip_errors:
                return None
            raise

    def recommend(self, user_id, n=None, novelty=True, interaction_threshold=None, **kwds):
        uid = self._require_user(user_id)
        ranked = self._recommend(uid, self.n_items if n is None else n, novelty, interaction_threshold)
        to_item = self.interaction_dataset.iid_to_item
        return [(score, to_item(iid)) for score, iid in ranked]

    def _recommend(self, uid, n, novelty, threshold):
        ranked = self._rank(uid, range(self.n_items), n, novelty)
        return ranked if threshold is None else [pair for pair in ranked if pair[0] >= threshold]

    def rank(self, user_id, item_ids, novelty=True, skip_invalid_items=True, **kwds):
        uid = self._require_user(user_id)
        ds = self.interaction_dataset
        iids = []
        for item_id in item_ids:
            iid = ds.item_to_iid(item_id)
            if iid is None and not skip_invalid_items:
                raise Exception(f'Item {item_id} was not found.')
            if iid is not None:
                iids.append(iid)
        n = kwds.get('n', len(iids))
        assert n <= len(iids), \
            f'The number of best items to return must be <= len(item_ids) (current value is {n} > {len(iids)})'
        return [(score, ds.iid_to_item(iid)) for score, iid in self._rank(uid, iids, n, novelty)]

    def _rank(self, uid, iids, n, novelty):
        """Default: one _predict per candidate; nlargest over (prediction, iid) => ties broken by larger iid."""
        candidates = set(iids)
        if novelty:
            candidates -= set(self.interaction_dataset.select(f'uid == {uid}').values_list('iid', to_list=True))
        scored = ((self._predict(uid, iid), iid) for iid in candidates)
        return nlargest(n, (pair for pair in scored if pair[0] is not None))

    def _standardize_value(self, value):
        """[min_interaction, max_interaction] -> [0, 1] (recommender_abc.py:463-465)."""
        return (value - self.min_interaction) / (self.max_interaction - self.min_interaction)

    def _rescale_value(self, value):
        return self.min_interaction + value * (self.max_interaction - self.min_interaction)

    # ---- logging ------------------------------------------------------------------------------------
    def _log_initial_info(self):
        cells = self.n_users * self.n_items
        for line in (f'Max. interaction value: {self.max_interaction}', f'Min. interaction value: {self.min_interaction}',
                     f'Interaction threshold value: {self.interaction_threshold}', f'Number of unique users: {self.n_users}',
                     f'Number of unique items: {self.n_items}', f'Number of training points: {self.n_rows}',
                     f'Sparsity level: approx. {round(100 * (1 - self.n_rows / cells), 4)}%'):
            self._info(line)

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

    # ---- save / load (the reference dumps the whole object with joblib, recommender_abc.py:503-524) ----------------
    def save(self, save_path):
        state = {k: v for k, v in self.__dict__.items() if k not in _UNPICKLED}
        if hasattr(self, '_engine'):
            state['__params__'] = self._engine.get_params()
        with open(save_path, 'wb') as f:
            pickle.dump({'cls': type(self), 'state': state}, f)

    @staticmethod
    def load(load_path):
        with open(load_path, 'rb') as f:
            blob = pickle.load(f)
        state = blob['state']
        obj = blob['cls'].__new__(blob['cls'])
        RecommenderABC.__init__(obj, verbose=state.get('verbose', True), seed=state.get('seed'))
        params = state.pop('__params__', None)
        obj.__dict__.update(state)
        if params is not None:
            obj._restore_engine(params)
        return obj
