"""Per-epoch loss / callback bookkeeping with the attribute and method names fit() and the early-stopping rules use
(`epoch_losses`, `epoch_callback_results`, `called_epochs`; DRecPy/Evaluation/loss_tracker.py).  Plotting is optional."""


class LossTracker:
    def __init__(self):
        self.reset_epoch_losses()
        self.epoch_callback_results = {}     # metric name -> values, one per callback invocation
        self.called_epochs = []              # epochs at which the callback ran (ascending)

    # ---- losses ----
    def reset_epoch_losses(self):
        self.epoch_losses = []
        self._loss_sum = 0.0

    def add_epoch_loss(self, loss):
        self.epoch_losses.append(loss)
        self._loss_sum += loss

    def get_epoch_avg_loss(self):
        return self._loss_sum / len(self.epoch_losses) if self.epoch_losses else 0

    @property
    def curr_avg_epoch_loss(self):
        return self.get_epoch_avg_loss()

    # ---- callback metrics ----
    def add_epoch_callback_result(self, name, result, epoch):
        self.epoch_callback_results.setdefault(name, []).append(result)
        if not self.called_epochs or epoch > self.called_epochs[-1]:
            self.called_epochs.append(epoch)

    def display_graph(self, model_name=None, stopping_epoch=None, block=False):
        try:
            import matplotlib.pyplot as plt
        except ImportError:
            return
        _, ax = plt.subplots()
        ax.plot(range(1, len(self.epoch_losses) + 1), self.epoch_losses, label='loss')
        for name, values in self.epoch_callback_results.items():
            ax.plot(self.called_epochs[-len(values):], values, marker='o', label=name)
        if stopping_epoch is not None:
            ax.axvline(stopping_epoch, linestyle='--', color='grey')
        ax.set_xlabel('epoch (one mini-batch each)')
        ax.set_title(model_name or '')
        ax.legend()
        plt.show(block=block)
