"""Host batch -> device in ONE asynchronous copy: numpy arrays are packed into a pinned staging buffer (a small ring, so the
host can run ahead of the device) and land in one device buffer; callers take typed views of it.  Several small pageable
copies per step each wait for the stream to drain — on this path that alone serialised host and device."""
import numpy as np
import torch


class StagedUpload:
    def __init__(self, device, slots=4):
        self.device, self.n = torch.device(device), slots
        self.host, self.ev, self.i = [None] * slots, [None] * slots, 0

    def __call__(self, arrays):
        """arrays: contiguous numpy arrays.  Returns (device uint8 tensor owning the bytes, [typed device views])."""
        offs, total = [], 0
        for a in arrays:
            offs.append(total)
            total += (a.nbytes + 15) & ~15
        total = max(total, 16)
        k = self.i % self.n
        self.i += 1
        if self.host[k] is None or self.host[k].numel() < total:
            self.host[k] = torch.empty(int(total * 1.5) + 4096, dtype=torch.uint8, pin_memory=True)
            self.ev[k] = None
        if self.ev[k] is not None:
            self.ev[k].synchronize()                 # the copy that last read this pinned slot has finished
        hv = self.host[k].numpy()
        for a, o in zip(arrays, offs):
            hv[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        dev.copy_(self.host[k][:total], non_blocking=True)
        self.ev[k] = torch.cuda.Event()
        self.ev[k].record(torch.cuda.current_stream(self.device))
        views = [dev[o:o + a.nbytes].view(_TORCH[a.dtype.type]).reshape(a.shape) if a.nbytes else
                 torch.empty(a.shape, dtype=_TORCH[a.dtype.type], device=self.device) for a, o in zip(arrays, offs)]
        return dev, views


_TORCH = {np.int32: torch.int32, np.int64: torch.int64, np.float32: torch.float32, np.float64: torch.float64, np.uint8: torch.uint8}
