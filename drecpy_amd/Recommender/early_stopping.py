"""Early-stopping rules (API of DRecPy/Recommender/EarlyStopping/*): host control logic only; the engine's part is the
device snapshot/restore used when a rule picks an earlier epoch."""
from abc import ABC, abstractmethod


class InvalidRequiredValidationMetricsException(Exception):
    pass


class InvalidEpochValidationResultsException(Exception):
    pass


class EarlyStoppingRuleABC(ABC):
    def __init__(self, required_validation_metrics, **kwds):
        if required_validation_metrics is not None and not isinstance(required_validation_metrics, list):
            raise InvalidRequiredValidationMetricsException('required_validation_metrics must be a list or None')
        self.required_validation_metrics = required_validation_metrics

    def compute(self, epoch_losses, epoch_validation_results, called_epochs_validation_results, **kwds):
        if self.required_validation_metrics:
            n = None
            for m in self.required_validation_metrics:
                if m not in epoch_validation_results:
                    raise InvalidEpochValidationResultsException(f'Validation metric "{m}" was not found on the epoch '
                                                                 'callback results.')
                if n is not None and len(epoch_validation_results[m]) != n:
                    raise InvalidEpochValidationResultsException('Validation metrics have different lengths.')
                n = len(epoch_validation_results[m])
            if n == 0 or n != len(called_epochs_validation_results):
                raise InvalidEpochValidationResultsException('Validation results and called epochs differ in length.')
        return self._compute_best_epoch(epoch_losses, epoch_validation_results, called_epochs_validation_results, **kwds)

    @abstractmethod
    def _compute_best_epoch(self, epoch_losses, epoch_validation_results, called_epochs_validation_results, **kwds):
        pass

    @abstractmethod
    def stop_training(self, current_epoch, best_computed_epoch, target_epoch, **kwds):
        pass


class MaxValidationValueRule(EarlyStoppingRuleABC):
    """Never stops; reports the evaluated epoch with the largest value of `validation_metric`."""

    def __init__(self, validation_metric, **kwds):
        super().__init__(required_validation_metrics=[validation_metric])
        self.validation_metric = validation_metric

    def _compute_best_epoch(self, epoch_losses, epoch_validation_results, called_epochs_validation_results, **kwds):
        vals = epoch_validation_results[self.validation_metric]
        best = max(range(len(vals)), key=lambda i: (vals[i], -i))
        return called_epochs_validation_results[best]

    def stop_training(self, current_epoch, best_computed_epoch, target_epoch, **kwds):
        return False
