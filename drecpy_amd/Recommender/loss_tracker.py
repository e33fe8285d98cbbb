"""Per-epoch loss / callback bookkeeping (API of DRecPy/Evaluation/loss_tracker.py; plotting is optional)."""


class LossTracker:
    def __init__(self):
        self.epoch_losses = []
        self.curr_avg_epoch_loss = 0
        self.epoch_callback_results = {}
        self.called_epochs = []

    def add_epoch_loss(self, loss):
        self.epoch_losses.append(loss)
        self.curr_avg_epoch_loss += (loss - self.curr_avg_epoch_loss) / len(self.epoch_losses)

    def get_epoch_avg_loss(self):
        return self.curr_avg_epoch_loss

    def reset_epoch_losses(self):
        self.epoch_losses = []
        self.curr_avg_epoch_loss = 0

    def add_epoch_callback_result(self, name, result, epoch):
        self.epoch_callback_results.setdefault(name, []).append(result)
        if len(self.called_epochs) == 0 or self.called_epochs[-1] < epoch:
            self.called_epochs.append(epoch)

    def display_graph(self, model_name=None, stopping_epoch=None, block=False):
        try:
            import matplotlib.pyplot as plt
        except ImportError:
            return
        fig, ax = plt.subplots()
        ax.plot(range(1, len(self.epoch_losses) + 1), self.epoch_losses, label='loss')
        for name, vals in self.epoch_callback_results.items():
            ax.plot(self.called_epochs[-len(vals):], vals, label=name)
        if stopping_epoch is not None:
            ax.axvline(stopping_epoch, linestyle='--')
        ax.set_title(model_name or '')
        ax.legend()
        plt.show(block=block)
